"""HIP-vs-oracle parity of the Hamming matching kernels (rows M1, M2), through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bf(gpu_ctx, descs_a, descs_b, max_n, ratio=0.7):
    """descs_*: list of [n,32] uint8 arrays (ragged).  Returns per-pair (idx2, dist2, accept)."""
    import torch
    import orbhip
    P = len(descs_a)
    A = np.zeros((P, max_n, 32), np.uint8)
    B = np.zeros((P, max_n, 32), np.uint8)
    nA = np.array([len(d) for d in descs_a], np.int32)
    nB = np.array([len(d) for d in descs_b], np.int32)
    for p in range(P):
        A[p, :nA[p]] = descs_a[p]
        B[p, :nB[p]] = descs_b[p]
    dA, dB, dnA, dnB = _dev(A), _dev(B), _dev(nA), _dev(nB)
    idx = torch.full((P, max_n, 2), -7, dtype=torch.int32, device="cuda")
    dist = torch.full((P, max_n, 2), -7, dtype=torch.int32, device="cuda")
    acc = torch.full((P, max_n), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    orbhip.match_bf2nn_device(gpu_ctx, dA.data_ptr(), dnA.data_ptr(), max_n * 32, dB.data_ptr(), dnB.data_ptr(),
                              max_n * 32, P, max_n, ratio, idx.data_ptr(), dist.data_ptr(), acc.data_ptr())
    gpu_ctx.synchronize()
    idx, dist, acc = idx.cpu().numpy(), dist.cpu().numpy(), acc.cpu().numpy()
    return [(idx[p, :nA[p]], dist[p, :nA[p]], acc[p, :nA[p]]) for p in range(P)]


def test_descriptor_distance_host():
    import orbhip
    import oracle_match_bind as om
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        d = orbhip.descriptor_distance(a, b)
        assert d == om.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())
    z = np.zeros(32, np.uint8)
    assert orbhip.descriptor_distance(z, z) == 0
    assert orbhip.descriptor_distance(z, ~z) == 256


def test_bf2nn_random_ragged(gpu_ctx):
    import oracle_match_bind as om
    rng = np.random.default_rng(1)
    sizes = [(0, 5), (5, 0), (1, 1), (3, 1), (257, 300), (1000, 1037), (256, 256), (700, 2)]
    da = [rng.integers(0, 256, (a, 32), dtype=np.uint8) for a, _ in sizes]
    db = [rng.integers(0, 256, (b, 32), dtype=np.uint8) for _, b in sizes]
    got = _bf(gpu_ctx, da, db, 1100)
    for p, (a, b) in enumerate(sizes):
        oi, od, oa = om.bf2nn(da[p], db[p], 0.7)
        np.testing.assert_array_equal(got[p][0], oi, err_msg="idx pair %d" % p)
        np.testing.assert_array_equal(got[p][1], od, err_msg="dist pair %d" % p)
        np.testing.assert_array_equal(got[p][2], oa, err_msg="accept pair %d" % p)


def test_bf2nn_ties_and_duplicates(gpu_ctx):
    """Collisions: duplicated train rows -> lowest index must win both slots' ordering."""
    import oracle_match_bind as om
    rng = np.random.default_rng(2)
    base = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    B = np.concatenate([base, base, base[:10]])          # every row appears 2-3 times
    A = base.copy()
    A[::3, 0] ^= 1
    got = _bf(gpu_ctx, [A], [B], 128)[0]
    oi, od, oa = om.bf2nn(A, B, 0.7)
    np.testing.assert_array_equal(got[0], oi)
    np.testing.assert_array_equal(got[1], od)
    np.testing.assert_array_equal(got[2], oa)
    assert (got[0][:, 0] < 40).all()


def test_bf2nn_on_extracted_frames(gpu_ctx):
    import orbhip
    import oracle_match_bind as om
    ext = orbhip.Extractor(gpu_ctx, 1000, 1.2, 8, 20, 7)
    imgs = orbhip.synth_frames(640, 480, 4, seed=31)
    res = ext.extract_host(imgs)
    da = [res[i][1] for i in range(3)]
    db = [res[i + 1][1] for i in range(3)]
    got = _bf(gpu_ctx, da, db, ext.max_keypoints)
    n_acc = 0
    for p in range(3):
        oi, od, oa = om.bf2nn(da[p], db[p], 0.7)
        np.testing.assert_array_equal(got[p][0], oi)
        np.testing.assert_array_equal(got[p][1], od)
        np.testing.assert_array_equal(got[p][2], oa)
        n_acc += int(oa.sum())
    assert n_acc > 100          # consecutive synthetic frames really do match
    ext.close()
