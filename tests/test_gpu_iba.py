"""GPU parity of the inertial local BA (orbhip_inertial_ba_solve_batch, csrc/iba_kernels.hip) against the CPU oracle
(oracle/iba_oracle.c) on the same synthetic visual-inertial windows.  Tolerance: RMSE <= 1e-4 on keyframe states and points
(BASELINE.json north_star's BA tolerance), identical outlier flags and LM trial counts.  PARITY UNPINNED against the real
reference (see oracle/iba_oracle.h)."""
import ctypes as C
import numpy as np
import pytest
import oracle_iba_bind as ib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import orbhip
    ctx = orbhip.Context(0)
    yield orbhip, ctx
    ctx.close()


def _gpu_solve(hip, wins, large=False, params=None):
    orbhip, ctx = hip
    structs = [w.struct(orbhip.IbaWindow) for w in wins]
    p = params or orbhip.iba_default_params(large)
    return orbhip.inertial_ba_solve_batch(ctx, structs, [w.kf0 for w in wins], [w.pts0 for w in wins], p)


def _rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a) - np.asarray(b)) ** 2))) if np.size(a) else 0.0


def _compare(win, gpu, cpu, tol=1e-4):
    kf_g, pts_g, out_g, st_g = gpu
    kf_c, pts_c, out_c, st_c = cpu
    assert st_g["failed"] == st_c.failed
    assert st_g["iterations_run"] == st_c.iterations_run and st_g["lm_trials"] == st_c.lm_trials, (st_g, st_c.iterations_run, st_c.lm_trials)
    assert abs(st_g["err"] - st_c.err) <= 1e-9 * max(1.0, abs(st_c.err))
    assert abs(st_g["err_end"] - st_c.err_end) <= 1e-6 * max(1.0, abs(st_c.err_end))
    assert _rmse(kf_g, kf_c) <= tol and _rmse(pts_g, pts_c) <= tol, (_rmse(kf_g, kf_c), _rmse(pts_g, pts_c))
    assert np.array_equal(out_g, out_c)
    assert st_g["n_outliers"] == int(out_c.sum())


@pytest.mark.parametrize("seed,kw", [(21, dict(n_opt=8, n_fixed_vis=10, n_points=300)),
                                     (22, dict(n_opt=3, n_fixed_vis=2, n_points=60)),
                                     (23, dict(n_opt=10, n_fixed_vis=30, n_points=900, stereo_frac=0.0)),
                                     (24, dict(n_opt=10, n_fixed_vis=25, n_points=700, stereo_frac=1.0)),
                                     # two KannalaBrandt8 cameras per keyframe, EdgeMono(1) edges, twin edges on one (pose, point) pair
                                     (25, dict(n_opt=8, n_fixed_vis=10, n_points=400, fisheye_rig=True)),
                                     (26, dict(n_opt=10, n_fixed_vis=30, n_points=900, fisheye_rig=True))])
def test_inertial_ba_matches_oracle(hip, seed, kw):
    win = ib.make_window(seed, **kw)
    gpu = _gpu_solve(hip, [win])
    _compare(win, [x[0] for x in gpu], ib.solve(win))


def test_inertial_ba_large_variant(hip):
    win = ib.make_window(31, n_opt=25, n_fixed_vis=40, n_points=1200, large=True)
    gpu = _gpu_solve(hip, [win], large=True)
    _compare(win, [x[0] for x in gpu], ib.solve(win, ib.default_params(large=True)))


def test_inertial_ba_batch_of_windows_and_determinism(hip):
    wins = [ib.make_window(40 + i, n_opt=4 + i % 7, n_fixed_vis=3 + 2 * i, n_points=120 + 40 * i, fisheye_rig=(i % 3 == 2)) for i in range(9)]
    gpu = _gpu_solve(hip, wins)
    again = _gpu_solve(hip, wins)
    for i, w in enumerate(wins):
        _compare(w, [x[i] for x in gpu], ib.solve(w))
        assert np.array_equal(gpu[0][i], again[0][i]) and np.array_equal(gpu[1][i], again[1][i])      # fixed summation orders


def test_inertial_ba_resident_batch_matches_the_one_shot_call(hip):
    """orbhip_iba_batch_*: the constant part is packed and uploaded once, every solve starts from the states last set; results are the
    one-shot call's bit for bit (same kernel, same team size), re-solving is repeatable, set_states moves the starting point."""
    orbhip, ctx = hip
    wins = [ib.make_window(140 + i, n_opt=5 + i, n_fixed_vis=4 + 3 * i, n_points=150 + 60 * i, fisheye_rig=(i == 2)) for i in range(5)]
    structs = [w.struct(orbhip.IbaWindow) for w in wins]
    once = _gpu_solve(hip, wins)
    b = orbhip.IbaBatch(ctx, structs, [w.kf0 for w in wins], [w.pts0 for w in wins])
    try:
        del structs
        for _ in range(2):
            b.solve()
            got = b.download()
            for i, w in enumerate(wins):
                assert np.array_equal(got[0][i], once[0][i]) and np.array_equal(got[1][i], once[1][i]) and np.array_equal(got[2][i], once[2][i])
                assert got[3][i] == once[3][i]
                _compare(w, [x[i] for x in got], ib.solve(w))
        # second pass from the first pass's result, as a caller relinearising would: equals the one-shot call started there
        b.set_states(once[0], once[1])
        b.solve()
        got2 = b.download()
        for i, w in enumerate(wins):
            w.kf0, w.pts0 = once[0][i], once[1][i]
        twice = _gpu_solve(hip, wins)
        for i in range(len(wins)):
            if twice[3][i]["failed"]:
                assert got2[3][i]["failed"]
                continue
            assert np.array_equal(got2[0][i], twice[0][i]) and np.array_equal(got2[1][i], twice[1][i]) and got2[3][i] == twice[3][i]
    finally:
        b.close()


def test_inertial_ba_resident_batch_rejects_bad_arguments(hip):
    orbhip, ctx = hip
    h = C.c_void_p()
    assert orbhip.lib.orbhip_iba_batch_create(ctx.h, None, 1, None, None, C.byref(h)) == orbhip.E_BADARG
    assert orbhip.lib.orbhip_iba_batch_solve(None, None) == orbhip.E_BADARG
    assert orbhip.lib.orbhip_iba_batch_download(None, None, None, None, None) == orbhip.E_BADARG
    orbhip.lib.orbhip_iba_batch_destroy(None)


@pytest.mark.parametrize("team", [1, 2, 5, 16])
def test_inertial_ba_every_team_size(hip, team, monkeypatch):
    """The summation order of the team-wide sums depends on the team size G (team_sum2): every G the launch rule can pick -- 1 is
    what batches of more than ~128 windows and every solve that finds another team grid in flight run -- must reproduce the oracle's
    LM decisions and estimates.  ORBHIP_IBA_TEAM caps G; the diagnostic says what ran."""
    orbhip, _ = hip
    monkeypatch.setenv("ORBHIP_IBA_TEAM", str(team))
    wins = [ib.make_window(60 + i, n_opt=4 + 3 * i, n_fixed_vis=5 + 6 * i, n_points=150 + 250 * i, fisheye_rig=(i == 1)) for i in range(3)]
    gpu = _gpu_solve(hip, wins)
    assert orbhip.inertial_ba_last_team_size() == team
    for i, w in enumerate(wins):
        _compare(w, [x[i] for x in gpu], ib.solve(w))


def test_inertial_ba_many_windows_take_the_single_workgroup_path(hip):
    """More windows than the device holds teams for: G = 1 by the launch rule itself (no environment override)."""
    orbhip, _ = hip
    base = [ib.make_window(70 + i, n_opt=3 + i, n_fixed_vis=2 + i, n_points=60 + 30 * i) for i in range(4)]
    wins = [base[i % 4] for i in range(160)]
    gpu = _gpu_solve(hip, wins)
    assert orbhip.inertial_ba_last_team_size() == 1
    cpu = [ib.solve(w) for w in base]
    for i in range(160):
        _compare(wins[i], [x[i] for x in gpu], cpu[i % 4])


def test_inertial_ba_two_contexts_solve_concurrently(hip):
    """Two host threads, a context each, solving at the same time on one device: only one team grid may be in flight (the team
    barrier needs its whole grid resident), the other solve falls back to one workgroup per window.  Both must succeed and match
    the oracle; over the repetitions both kinds of launch occur."""
    import threading
    orbhip, _ = hip
    wins = [ib.make_window(80 + i, n_opt=6 + i, n_fixed_vis=8, n_points=400 + 100 * i) for i in range(2)]
    cpu = [ib.solve(w) for w in wins]
    ctxs = [orbhip.Context(0) for _ in range(2)]
    errs, teams = [], [[], []]
    start = threading.Barrier(2)

    def run(r):
        try:
            for rep in range(6):
                start.wait(timeout=120)
                gpu = _gpu_solve((orbhip, ctxs[r]), [wins[r]] * 3)
                teams[r].append(orbhip.inertial_ba_last_team_size())
                for k in range(3):
                    _compare(wins[r], [x[k] for x in gpu], cpu[r])
        except BaseException as e:                       # noqa: BLE001 (reported below)
            errs.append((r, repr(e)))
            start.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    for c in ctxs:
        c.close()
    assert not errs, errs
    assert all(len(t) == 6 for t in teams)
    assert any(g > 1 for t in teams for g in t), teams   # teams are still used when the device is free


def test_inertial_ba_fail_check_leaves_inputs(hip):
    win = ib.make_window(13, n_opt=4, n_fixed_vis=3, n_points=60)
    win.arrays["in_preint"][:, 13:16] += 40.0
    orbhip, _ = hip
    p = orbhip.iba_default_params(False)
    p.iterations, p.max_trials = 1, 1
    po = ib.default_params()
    po.iterations, po.max_trials = 1, 1
    gpu = _gpu_solve(hip, [win], params=p)
    cpu = ib.solve(win, po)
    assert gpu[3][0]["failed"] == cpu[3].failed
    if cpu[3].failed:
        assert np.array_equal(gpu[0][0], win.kf0) and np.array_equal(gpu[1][0], win.pts0)


def test_inertial_ba_rejects_malformed_windows(hip):
    orbhip, ctx = hip
    win = ib.make_window(50, n_opt=3, n_fixed_vis=1, n_points=30)
    win.arrays["edge_point"][:] = win.arrays["edge_point"][::-1].copy()           # not grouped by point
    with pytest.raises(orbhip.OrbHipError):
        _gpu_solve(hip, [win])
    win2 = ib.make_window(51, n_opt=3, n_fixed_vis=1, n_points=30)
    win2.arrays["in_kf1"][0] = 99
    with pytest.raises(orbhip.OrbHipError):
        _gpu_solve(hip, [win2])


def test_inertial_ba_golden_fixture_without_oracle(hip):
    """The HIP solver against the committed vectors of tests/golden/iba_golden.npz (a pinhole window with stereo + mono edges and a
    two-fisheye rig window, made by the oracle through tools/gen_golden.py) -- no live oracle involved."""
    from synth_iba import load_golden_windows
    cases = load_golden_windows()
    gpu = _gpu_solve(hip, [w for w, _ in cases])
    for i, (win, exp) in enumerate(cases):
        st = gpu[3][i]
        assert [st["iterations_run"], st["lm_trials"], st["n_outliers"], st["failed"]] == exp["stats"].tolist()
        assert _rmse(gpu[0][i], exp["kf"]) <= 1e-4 and _rmse(gpu[1][i], exp["pts"]) <= 1e-4
        assert _rmse(gpu[0][i], exp["kf"]) <= 1e-7, "closer than the tolerance in practice: a drift worth looking at"
        assert np.array_equal(gpu[2][i], exp["outlier"])
