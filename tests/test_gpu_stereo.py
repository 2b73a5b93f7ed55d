"""HIP-vs-oracle parity of Frame::ComputeStereoMatches, through the C ABI: bit-exact mvuRight / mvDepth."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_pairs(gpu_ctx, pairs, nfeat, mb, mbf, two_contexts=False):
    import torch
    import orbhip
    import oracle_bind as ob
    h, w = pairs[0][0].shape
    ctxR = orbhip.Context(0) if two_contexts else gpu_ctx
    extL = orbhip.Extractor(gpu_ctx, nfeat, 1.2, 8, 20, 7); extR = orbhip.Extractor(ctxR, nfeat, 1.2, 8, 20, 7)
    lefts = np.stack([p[0] for p in pairs]); rights = np.stack([p[1] for p in pairs])
    resL = extL.extract_host(lefts, lap=(0, 0)); resR = extR.extract_host(rights, lap=(0, 0))
    B, M = len(pairs), extL.max_keypoints
    ur = torch.full((B, M), 7.0, dtype=torch.float32, device="cuda"); dp = torch.full((B, M), 7.0, dtype=torch.float32, device="cuda")
    nk = torch.full((B,), -9, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    orbhip.compute_stereo_matches_device(extL, extR, mb, mbf, ur.data_ptr(), dp.data_ptr(), nk.data_ptr())
    gpu_ctx.synchronize()
    ur, dp, nk = ur.cpu().numpy(), dp.cpu().numpy(), nk.cpu().numpy()
    total = 0
    for f, (left, right) in enumerate(pairs):
        eL = ob.OracleExtractor(nfeat, 1.2, 8, 20, 7); eR = ob.OracleExtractor(nfeat, 1.2, 8, 20, 7)
        kpL, dL, _ = eL.extract(left, (0, 0)); kpR, dR, _ = eR.extract(right, (0, 0))
        assert kpL.tobytes() == resL[f][0].tobytes() and dR.tobytes() == resR[f][1].tobytes()      # same inputs on both sides
        kept, ur_ref, dp_ref, _ = ob.compute_stereo_matches(eL, eR, kpL, dL, kpR, dR, mb, mbf)
        n = len(kpL)
        assert nk[f] == kept, (f, nk[f], kept)
        assert ur[f, :n].tobytes() == ur_ref.tobytes()
        assert dp[f, :n].tobytes() == dp_ref.tobytes()
        assert (ur[f, n:] == 7.0).all()
        total += kept
    extL.close(); extR.close()
    if two_contexts:
        ctxR.close()
    return total


def test_stereo_matches_parity_vga(gpu_ctx):
    from test_oracle_orb import make_stereo_pair
    pairs = [make_stereo_pair(640, 480, d, seed=60 + d) for d in (3, 12, 31, 0)]
    blank = np.full((480, 640), 128, np.uint8)
    pairs.append((pairs[0][0], blank))                      # right view without features: no candidates at all
    mbf = 40.0
    assert _run_pairs(gpu_ctx, pairs, 1000, mbf / 458.0, mbf) > 1000


def test_stereo_matches_parity_close_range_and_two_contexts(gpu_ctx):
    """Small maxD (mb large): most candidates fall outside the disparity gate; extractors on two contexts/streams."""
    from test_oracle_orb import make_stereo_pair
    pairs = [make_stereo_pair(752, 480, d, seed=80 + d) for d in (8, 20)]
    mbf = 47.9
    assert _run_pairs(gpu_ctx, pairs, 1500, mbf / 10.0, mbf, two_contexts=True) > 100     # maxD = 10 px


def test_stereo_matches_rejects_mismatched_extractors(gpu_ctx):
    import torch
    import orbhip
    a = orbhip.Extractor(gpu_ctx, 500, 1.2, 8, 20, 7); b = orbhip.Extractor(gpu_ctx, 600, 1.2, 8, 20, 7)
    img = orbhip.synth_frames(320, 240, 1, seed=1)
    a.extract_host(img, lap=(0, 0)); b.extract_host(img, lap=(0, 0))
    t = torch.zeros(4096, dtype=torch.float32, device="cuda")
    with pytest.raises(orbhip.OrbHipError):
        orbhip.compute_stereo_matches_device(a, b, 0.1, 40.0, t.data_ptr(), t.data_ptr())
    a.close(); b.close()


def test_stereo_two_consecutive_steps_on_two_contexts(gpu_ctx):
    """Two steps back to back, nothing waited for in between, different frames in each, the right extractor on its own context /
    stream: the right stream must wait for the previous step's stereo kernels (on the left stream) before its next extraction
    overwrites the pyramid, keypoints and descriptors they read.  Both steps bit-exact vs the oracle."""
    import torch
    import orbhip
    import oracle_bind as ob
    from test_oracle_orb import make_stereo_pair
    W, H, S, nfeat, mbf = 640, 480, 12, 1000, 40.0
    mb = mbf / 458.0
    steps = [[make_stereo_pair(W, H, 3 + 5 * ((k + 7 * s) % 7), seed=500 + 20 * s + k) for k in range(S)] for s in range(2)]
    ctxR = orbhip.Context(0)
    extL = orbhip.Extractor(gpu_ctx, nfeat, 1.2, 8, 20, 7); extR = orbhip.Extractor(ctxR, nfeat, 1.2, 8, 20, 7)
    extL.reserve(W, H, S); extR.reserve(W, H, S)
    M = extL.max_keypoints
    dL = [torch.from_numpy(np.stack([p[0] for p in st])).cuda() for st in steps]
    dR = [torch.from_numpy(np.stack([p[1] for p in st])).cuda() for st in steps]
    ur = [torch.full((S, M), 7.0, dtype=torch.float32, device="cuda") for _ in range(2)]
    dp = [torch.full((S, M), 7.0, dtype=torch.float32, device="cuda") for _ in range(2)]
    nk = [torch.full((S,), -9, dtype=torch.int32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for s in range(2):                                              # asynchronous: step 1's extractions are queued behind step 0's stereo
        extL.extract_device(dL[s].data_ptr(), W, H, W, W * H, S, (0, 0))
        extR.extract_device(dR[s].data_ptr(), W, H, W, W * H, S, (0, 0))
        orbhip.compute_stereo_matches_device(extL, extR, mb, mbf, ur[s].data_ptr(), dp[s].data_ptr(), nk[s].data_ptr())
    gpu_ctx.synchronize(); ctxR.synchronize()
    total = 0
    for s in range(2):
        u, d, n = ur[s].cpu().numpy(), dp[s].cpu().numpy(), nk[s].cpu().numpy()
        for f in (0, 5, S - 1):
            eL = ob.OracleExtractor(nfeat, 1.2, 8, 20, 7); eR = ob.OracleExtractor(nfeat, 1.2, 8, 20, 7)
            kpL, deL, _ = eL.extract(steps[s][f][0], (0, 0)); kpR, deR, _ = eR.extract(steps[s][f][1], (0, 0))
            kept, ur_ref, dp_ref, _ = ob.compute_stereo_matches(eL, eR, kpL, deL, kpR, deR, mb, mbf)
            assert n[f] == kept, (s, f, n[f], kept)
            assert u[f, :len(kpL)].tobytes() == ur_ref.tobytes() and d[f, :len(kpL)].tobytes() == dp_ref.tobytes(), (s, f)
            total += kept
    assert total > 1000
    extL.close(); extR.close(); ctxR.close()
