"""HIP-vs-oracle parity of the batched pose-only BA (Optimizer::PoseOptimization), through the C ABI.
Floating point (FP64): poses within 1e-9 of the oracle (the only difference is the summation order of the
6x6 normal equations), outlier flags, round count and nBad identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["block", "wave"])
def pose_kernel_family(request, monkeypatch):
    """Both kernels on every test: four waves per frame (the default below 513 frames: shorter latency) and one wave per frame with
    the edges in registers (the throughput form)."""
    monkeypatch.setenv("ORBHIP_POSE_WAVE_MIN_FRAMES", "1" if request.param == "wave" else "1000000")
    return request.param


def _run(gpu_ctx, probs, max_edges):
    import torch
    import orbhip
    F = len(probs)
    Xw = np.zeros((F, max_edges, 3)); obs = np.full((F, max_edges, 3), -1.0); w = np.zeros((F, max_edges))
    n = np.array([len(p["Xw"]) for p in probs], np.int32)
    pose = np.stack([p["pose0"] for p in probs]).astype(np.float64)
    for f, p in enumerate(probs):
        Xw[f, :n[f]] = p["Xw"]; obs[f, :n[f]] = p["obs"]; w[f, :n[f]] = p["inv_sigma2"]
    t = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (Xw, obs, w, n, pose)]
    rig2 = probs[0].get("rig2")
    d_right = None
    if rig2 is not None:
        rt = np.zeros((F, max_edges), np.uint8)
        for f, p in enumerate(probs):
            rt[f, :n[f]] = p["right"]
        d_right = torch.from_numpy(rt).cuda()
    out = torch.full((F, max_edges), 9, dtype=torch.uint8, device="cuda")
    ninl = torch.full((F,), -9, dtype=torch.int32, device="cuda")
    stats = torch.full((F, 4), -9, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    orbhip.pose_optimization_device(gpu_ctx, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), F, max_edges,
                                    probs[0]["cam"], t[4].data_ptr(), out.data_ptr(), ninl.data_ptr(), stats.data_ptr(),
                                    kb8=probs[0].get("kb8"), rig2=rig2, d_right=None if d_right is None else d_right.data_ptr())
    gpu_ctx.synchronize()
    return t[4].cpu().numpy(), out.cpu().numpy(), ninl.cpu().numpy(), stats.cpu().numpy()


def _check(gpu_ctx, probs, max_edges):
    import oracle_ba_bind as ob
    pose, out, ninl, stats = _run(gpu_ctx, probs, max_edges)
    for f, p in enumerate(probs):
        r, pose_ref, out_ref, st = ob.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["cam"], p["pose0"], kb8=p.get("kb8"),
                                                        rig2=p.get("rig2"), right=p.get("right"))
        n = len(p["Xw"])
        assert ninl[f] == r, (f, ninl[f], r)
        np.testing.assert_array_equal(out[f, :n], out_ref)
        assert (out[f, n:] == 9).all()                                   # rows beyond n_edges are not touched
        if n >= 3:
            np.testing.assert_allclose(pose[f], pose_ref, rtol=0, atol=1e-9)
            # rounds and nBad are exact; once a round has converged the sign of rho = (chi - chi_trial) / scale of a
            # zero-progress trial depends on the summation order of chi2 (sequential in the reference, tree in the
            # kernel), so the count of such no-op trials / iterations may differ by a few
            assert stats[f][0] == st["rounds"] and stats[f][3] == st["n_bad"], (f, stats[f], st)
            # (a converged iteration retries with growing lambda until rho >= 0, up to 100 times): not compared
            assert 4 <= stats[f][1] <= 40 and stats[f][1] <= stats[f][2] <= 4000, (f, stats[f])
        else:
            assert np.array_equal(pose[f], p["pose0"])                   # untouched (Optimizer.cc:1040-1041)


def test_pose_optimization_parity_mixed_batch(gpu_ctx):
    import synth_ba
    probs = [synth_ba.make_pose_problem(100 + k, n=n, stereo_frac=sf, outlier_frac=of)
             for k, (n, sf, of) in enumerate([(800, 0.0, 0.1), (800, 1.0, 0.1), (500, 0.4, 0.3), (1000, 0.2, 0.0), (2, 0.0, 0.0),
                                              (9, 0.5, 0.0), (40, 0.0, 0.5), (2048, 0.3, 0.1), (300, 0.0, 0.6), (3, 1.0, 0.0)])]
    _check(gpu_ctx, probs, 2048)


def test_pose_optimization_golden(gpu_ctx):
    """Committed fixture, no oracle involved."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_golden.npz"))
    probs = [dict(Xw=g[f"Xw{k}"], obs=g[f"obs{k}"], inv_sigma2=g[f"w{k}"], cam=tuple(g[f"cam{k}"]), pose0=g[f"pose0_{k}"])
             for k in range(int(g["count"]))]
    for k, p in enumerate(probs):          # one camera per call
        pose, out, ninl, _ = _run(gpu_ctx, [p], 512)
        assert ninl[0] == int(g[f"r{k}"])
        np.testing.assert_array_equal(out[0, :len(p["Xw"])], g[f"out{k}"])
        np.testing.assert_allclose(pose[0], g[f"pose{k}"], rtol=0, atol=1e-9)


def test_pose_optimization_full_batch_properties(gpu_ctx):
    """1024 frames x 1000 edges: deterministic run to run, every pose close to the truth, gross outliers rejected."""
    import synth_ba
    base = [synth_ba.make_pose_problem(500 + k, n=1000, stereo_frac=0.25 * (k % 4), outlier_frac=0.1) for k in range(16)]
    probs = [base[k % 16] for k in range(1024)]
    a = _run(gpu_ctx, probs, 1000)
    b = _run(gpu_ctx, probs, 1000)
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()
    pose, out, ninl, stats = a
    for k in range(16):
        assert np.array_equal(pose[k], pose[k + 16 * 5])                 # same input, same answer anywhere in the batch
        p = base[k]
        err = min(np.linalg.norm(pose[k][:4] - p["pose_true"][:4]), np.linalg.norm(pose[k][:4] + p["pose_true"][:4])) + \
            np.linalg.norm(pose[k][4:] - p["pose_true"][4:])
        assert err < 0.02
        assert out[k][p["outlier_true"]].mean() > 0.98
        assert ninl[k] == 1000 - out[k].sum()


def test_pose_optimization_kannala_brandt_camera(gpu_ctx):
    """pFrame->mpCamera = KannalaBrandt8: monocular edges project through KannalaBrandt8.cpp:52-69 / :166-195.
    The projection runs through float atan2f / sqrtf; device and host libm differ in the last float ulp of theta
    (6e-8 rad), which moves the optimum by a few 1e-6 -- tolerance 1e-5, two orders below BASELINE's 1e-4."""
    import oracle_ba_bind as ob
    import synth_ba
    kb = (-0.0034, 0.0007, -0.0021, 0.0002)
    probs = [synth_ba.make_pose_problem(900 + k, n=n, stereo_frac=0.0, outlier_frac=of, kb8=kb) for k, (n, of) in enumerate([(600, 0.1), (900, 0.0), (40, 0.3)])]
    pose, out, ninl, stats = _run(gpu_ctx, probs, 1024)
    for f, p in enumerate(probs):
        r, pose_ref, out_ref, st = ob.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["cam"], p["pose0"], kb8=kb)
        n = len(p["Xw"])
        np.testing.assert_allclose(pose[f], pose_ref, rtol=0, atol=1e-5)
        assert int(np.sum(out[f, :n] != out_ref)) <= 2 and abs(int(ninl[f]) - r) <= 2


def test_pose_optimization_second_camera(gpu_ctx):
    """pFrame->mpCamera2 (Optimizer.cc:960-1037): left-camera edges through KannalaBrandt8 + EdgeSE3ProjectXYZOnlyPoseToBody
    for the observations made in the second camera."""
    import synth_ba
    kb = (-0.0034, 0.0007, -0.0021, 0.0002)
    rig = dict(Trl=(0.004, -0.012, 0.002, 0.99991, -0.101, 0.0007, 0.0012), cam=(458.0, 457.0, 322.0, 238.0), kb=(0.0031, 0.0007, -0.0019, 0.0003))
    probs = [synth_ba.make_pose_problem(950 + k, n=n, outlier_frac=of, kb8=kb, rig2=rig) for k, (n, of) in enumerate([(700, 0.1), (900, 0.0), (50, 0.2)])]
    assert all(p["right"].sum() > 10 for p in probs)
    _check(gpu_ctx, probs, 1024)
