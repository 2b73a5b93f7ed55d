"""HIP-vs-oracle parity of the ORB front-end, stage by stage, through the C ABI.
Bar: bit-exact (SURVEY.md 8c "Definition of bit-exact": candidates, octree set, angles,
descriptors, output order incl. lapping split, return value)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(gpu_ctx, nfeat=1000, nlev=8, ini=20, mn=7, scale=1.2):
    import orbhip
    import oracle_bind as ob
    return orbhip.Extractor(gpu_ctx, nfeat, scale, nlev, ini, mn), ob.OracleExtractor(nfeat, scale, nlev, ini, mn)


def _compare_frame(ext, ora, imgs, f, lap, got):
    kp, desc, mono = ora.extract(imgs[f], lap)
    for l in range(ext.nlevels):
        np.testing.assert_array_equal(ext.pyramid_level(f, l), ora.pyramid_level(l), err_msg="pyramid L%d" % l)
    for l in range(ext.nlevels):
        gx, gy, gs = ext.fast_candidates(f, l)
        ox, oy, os_ = ora.fast_candidates(l)
        assert len(gx) == len(ox), "FAST candidate count L%d: %d vs %d" % (l, len(gx), len(ox))
        np.testing.assert_array_equal(gx, ox, err_msg="cand x L%d" % l)
        np.testing.assert_array_equal(gy, oy, err_msg="cand y L%d" % l)
        np.testing.assert_array_equal(gs, os_, err_msg="cand score L%d" % l)
    for l in range(ext.nlevels):
        gk = ext.level_keypoints(f, l)
        ok = ora.level_keypoints(l)
        assert len(gk) == len(ok), "octree count L%d: %d vs %d" % (l, len(gk), len(ok))
        np.testing.assert_array_equal(gk["x"], ok["x"], err_msg="octree x L%d" % l)
        np.testing.assert_array_equal(gk["y"], ok["y"], err_msg="octree y L%d" % l)
        assert gk["angle"].tobytes() == ok["angle"].tobytes(), "angle bits L%d" % l
        assert gk.tobytes() == ok.tobytes(), "level keypoints L%d" % l
    for l in range(ext.nlevels):
        ob_ = ora.blurred_level(l)
        if ob_ is not None:
            np.testing.assert_array_equal(ext.blurred_level(f, l), ob_, err_msg="blur L%d" % l)
    gk, gd, gm = got[f]
    assert gm == mono
    assert len(gk) == len(kp)
    assert gk.tobytes() == kp.tobytes(), "final keypoints"
    assert gd.tobytes() == desc.tobytes(), "final descriptors"
    return len(kp)


@pytest.mark.parametrize("w,h,nfeat,lap", [
    (640, 480, 1000, (0, 1000)),     # monocular ctor lapping: everything 'stereo', reversed
    (640, 480, 1000, (0, 0)),        # stereo ctor lapping: everything 'mono'
    (640, 480, 1000, (200, 420)),    # genuine split
    (752, 480, 1000, (0, 1000)),     # EuRoC native size
    (512, 512, 1500, (0, 511)),      # TUM-VI fisheye (Frame.cc:1056)
    (333, 277, 300, (100, 150)),     # ragged sizes, few features
])
@pytest.mark.parametrize("rows_min", ["1", "16"])         # row-streaming pyramid / blur kernels (batches >= 16 by default) and tile kernels
def test_extract_parity_stages(gpu_ctx, w, h, nfeat, lap, rows_min, monkeypatch):
    import orbhip
    monkeypatch.setenv("ORBHIP_ROWS_MIN_BATCH", rows_min)
    ext, ora = _mk(gpu_ctx, nfeat)
    imgs = orbhip.synth_frames(w, h, 3, seed=1000 + w + nfeat)
    got = ext.extract_host(imgs, lap)
    total = 0
    for f in range(3):
        total += _compare_frame(ext, ora, imgs, f, lap, got)
    assert total > 0.8 * nfeat * 3
    ext.close()


def test_extract_padded_pyramid(gpu_ctx):
    import orbhip
    ext, ora = _mk(gpu_ctx)
    imgs = orbhip.synth_frames(640, 480, 1, seed=5)
    ext.extract_host(imgs)
    ora.extract(imgs[0])
    for l in (0, 3, 7):
        np.testing.assert_array_equal(ext.pyramid_level(0, l, padded=True), ora.pyramid_level(l, padded=True))
    allp = ext.pyramid_padded(0)                 # the whole pyramid in one pass: what ORBextractor::operator() leaves in mvImagePyramid
    for l in range(8):
        np.testing.assert_array_equal(allp[l], ora.pyramid_level(l, padded=True), err_msg="padded level %d" % l)
    ext.close()


def test_extract_edge_inputs(gpu_ctx):
    """Flat image (no corners at either threshold), pure noise, a min-threshold-only image, and one that mixes both kinds of cells."""
    import orbhip
    ext, ora = _mk(gpu_ctx)
    rng = np.random.default_rng(3)
    flat = np.full((480, 640), 77, np.uint8)
    noise = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    low = (128 + 6 * (rng.integers(0, 2, (60, 80)).repeat(8, 0).repeat(8, 1)) + rng.integers(0, 3, (480, 640))).astype(np.uint8)
    # cells that find corners at iniThFAST next to cells that only do at minThFAST (the kernel's second pass runs per cell), in stripes
    # narrower and wider than a cell
    mixed = low.copy()
    tex = orbhip.synth_frames(640, 480, 1, seed=4242)[0]
    for x0, w in ((0, 97), (180, 20), (260, 140), (470, 33), (560, 80)):
        mixed[:, x0:x0 + w] = tex[:, x0:x0 + w]
    mixed[200:230, :] = low[200:230, :]
    imgs = np.stack([flat, noise, low, mixed])
    got = ext.extract_host(imgs)
    assert len(got[0][0]) == 0 and got[0][2] == 0
    for f in range(4):
        _compare_frame(ext, ora, imgs, f, (0, 1000), got)
    ext.close()


def test_blur_on_the_matrix_cores_is_bit_exact(gpu_ctx, monkeypatch):
    """k_blur_mfma (opt-in, ORBHIP_BLUR_MFMA=1 at extractor creation): the 7x7 blur as two banded int8 MFMA products per window, image
    borders folded into the operand tables.  Descriptors are the blur's consumers: frames of several sizes (widths that are and are
    not multiples of 16 / 32, a level count that leaves narrow top levels) must match the default path and the oracle bit for bit."""
    import orbhip
    monkeypatch.setenv("ORBHIP_ROWS_MIN_BATCH", "1")
    for (W, H, nfeat, nlev, scale, seed) in ((640, 480, 1000, 8, 1.2, 31), (752, 480, 1200, 8, 1.2, 32), (1241, 376, 2000, 8, 1.2, 33), (801, 603, 1500, 6, 1.3, 34)):
        imgs = orbhip.synth_frames(W, H, 3, seed=seed)
        monkeypatch.setenv("ORBHIP_BLUR_MFMA", "0")
        ext0, ora = _mk(gpu_ctx, nfeat=nfeat, nlev=nlev, scale=scale)
        ref = ext0.extract_host(imgs)
        ext0.close()
        monkeypatch.setenv("ORBHIP_BLUR_MFMA", "1")
        ext1, _ = _mk(gpu_ctx, nfeat=nfeat, nlev=nlev, scale=scale)
        got = ext1.extract_host(imgs)
        for f in range(3):
            assert got[f][0].tobytes() == ref[f][0].tobytes() and got[f][1].tobytes() == ref[f][1].tobytes(), (W, H, f)
            _compare_frame(ext1, ora, imgs, f, (0, 1000), got)          # every level's blurred image and the descriptors against the oracle
        ext1.close()


def test_default_blur_kernel_by_size_and_batch(gpu_ctx, monkeypatch):
    """The blur kernel a batch takes is a measured choice (orbhip_extractor_blur_kernel): LDS tiles below 16 frames, the row-streaming
    kernel from there, the matrix-core one for images of up to 320 K pixels in batches of 128 frames or more -- and every choice
    gives the same bytes: a 256-frame VGA batch (matrix cores by default) equals the same frames extracted 8 at a time (tiles) and with
    ORBHIP_BLUR_MFMA=0 (rows), frame by frame; the first frames also against the oracle."""
    import orbhip
    monkeypatch.delenv("ORBHIP_BLUR_MFMA", raising=False)
    imgs = orbhip.synth_frames(640, 480, 256, seed=77)
    ext, ora = _mk(gpu_ctx)
    ext.reserve(640, 480, 256)
    assert (ext.blur_kernel(1), ext.blur_kernel(16), ext.blur_kernel(127), ext.blur_kernel(128)) == ("k_blur", "k_blur_rows", "k_blur_rows", "k_blur_mfma")
    big = ext.extract_host(imgs)
    for f in range(2):
        _compare_frame(ext, ora, imgs, f, (0, 1000), big)          # (reads the levels of the extraction just made: before the small batches)
    for f0 in (0, 120, 248):
        small = ext.extract_host(imgs[f0:f0 + 8])
        for f in range(8):
            assert small[f][0].tobytes() == big[f0 + f][0].tobytes() and small[f][1].tobytes() == big[f0 + f][1].tobytes(), f0 + f
    ext.close()
    monkeypatch.setenv("ORBHIP_BLUR_MFMA", "0")
    ext0, _ = _mk(gpu_ctx)
    ext0.reserve(640, 480, 256)
    assert ext0.blur_kernel(256) == "k_blur_rows"
    rows = ext0.extract_host(imgs)
    for f in range(256):
        assert rows[f][0].tobytes() == big[f][0].tobytes() and rows[f][1].tobytes() == big[f][1].tobytes(), f
    ext0.close()
    monkeypatch.delenv("ORBHIP_BLUR_MFMA", raising=False)
    exth, _ = _mk(gpu_ctx, nfeat=2000)
    exth.reserve(1920, 1080, 4)
    assert exth.blur_kernel(512) == "k_blur_rows"           # above VGA the row-streaming kernel stays
    exth.close()


def test_extract_other_params(gpu_ctx):
    import orbhip
    ext, ora = _mk(gpu_ctx, nfeat=2000, nlev=5, ini=12, mn=5, scale=1.3)
    imgs = orbhip.synth_frames(800, 600, 2, seed=99)
    got = ext.extract_host(imgs, (0, 400))
    for f in range(2):
        _compare_frame(ext, ora, imgs, f, (0, 400), got)
    ext.close()


def test_extract_batch_consistency(gpu_ctx):
    """A frame's result must not depend on its batch neighbours (no cross-frame leakage)."""
    import orbhip
    ext, _ = _mk(gpu_ctx)
    imgs = orbhip.synth_frames(640, 480, 24, seed=11)
    a = ext.extract_host(imgs)
    b = ext.extract_host(imgs[5:6])
    assert a[5][0].tobytes() == b[0][0].tobytes() and a[5][1].tobytes() == b[0][1].tobytes()
    c = ext.extract_host(imgs[::-1].copy())
    for f in range(24):
        assert a[f][1].tobytes() == c[23 - f][1].tobytes()
    ext.close()


def test_empty_image_and_bad_args(gpu_ctx):
    import orbhip
    ext, _ = _mk(gpu_ctx)
    rc = orbhip.lib.orbhip_extract_batch_host(ext.h, None, 0, 0, 0, 0, 1, 0, 1000, None, None, 0, None, None)
    assert rc == orbhip.E_EMPTY          # ORBextractor.cc:1072-1073 returns -1 on empty
    with pytest.raises(orbhip.OrbHipError):
        ext.reserve(64, 48, 1)           # smaller than one FAST cell: rejected loudly
    ext.close()


def _dev_extract(gpu_ctx, ext, imgs, lap, row_stride=None):
    """Device entry point (orbhip_extract_batch_device): images resident in HBM, results fetched from the
    device result arrays."""
    import ctypes as C
    import torch
    import orbhip
    B, H, W = imgs.shape
    rs = row_stride or W
    buf = np.zeros((B, H, rs), np.uint8)
    buf[:, :, :W] = imgs
    d = torch.from_numpy(buf).cuda()
    torch.cuda.synchronize()
    ext.extract_device(d.data_ptr(), W, H, rs, rs * H, B, lap)
    gpu_ctx.synchronize()
    kp_p, desc_p, cnt_p, mono_p = ext.results_device()
    mk = ext.max_keypoints
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    kp = np.zeros((B, mk), orbhip.KP_DTYPE); desc = np.zeros((B, mk, 32), np.uint8)
    cnt = np.zeros(B, np.int32); mono = np.zeros(B, np.int32)
    for dst, src in ((kp, kp_p), (desc, desc_p), (cnt, cnt_p), (mono, mono_p)):
        assert hip.hipMemcpy(dst.ctypes.data, src, dst.nbytes, 2) == 0
    return [(kp[f, :cnt[f]], desc[f, :cnt[f]], int(mono[f])) for f in range(B)]


@pytest.mark.parametrize("w,h,stride", [(640, 480, None), (333, 277, 333), (333, 277, 336)])
def test_device_entry_point_matches_host_entry_point(gpu_ctx, w, h, stride):
    """orbhip_extract_batch_device (level 0 aliases the caller's buffer; odd strides are staged) == host path."""
    import orbhip
    ext, _ = _mk(gpu_ctx, 500)
    imgs = orbhip.synth_frames(w, h, 5, seed=4)
    a = ext.extract_host(imgs, (0, 300))
    b = _dev_extract(gpu_ctx, ext, imgs, (0, 300), stride)
    for f in range(5):
        assert a[f][2] == b[f][2]
        assert a[f][0].tobytes() == b[f][0].tobytes() and a[f][1].tobytes() == b[f][1].tobytes()
    ext.close()


@pytest.mark.parametrize("w,h,nfeat", [(1920, 1080, 2000), (3840, 2160, 2000)])
def test_extract_parity_hd_and_4k(gpu_ctx, w, h, nfeat):
    """BASELINE configs #3 sizes: HD / 4K frames, 2000 features, bit-exact vs the oracle (final outputs +
    pre-octree candidates of the largest level)."""
    import orbhip
    ext, ora = _mk(gpu_ctx, nfeat)
    imgs = orbhip.synth_frames(w, h, 1, seed=w)
    got = ext.extract_host(imgs, (0, 1000))
    kp, desc, mono = ora.extract(imgs[0], (0, 1000))
    gx, gy, gs = ext.fast_candidates(0, 0)
    ox, oy, os_ = ora.fast_candidates(0)
    np.testing.assert_array_equal(gx, ox); np.testing.assert_array_equal(gy, oy); np.testing.assert_array_equal(gs, os_)
    assert got[0][2] == mono and len(got[0][0]) == len(kp) > 0.9 * nfeat
    assert got[0][0].tobytes() == kp.tobytes()
    assert got[0][1].tobytes() == desc.tobytes()
    ext.close()


def test_full_batch_1024_properties(gpu_ctx):
    """BASELINE config #2 at full size (1024 VGA frames): sampled frames bit-exact vs the oracle, run-to-run
    determinism of the whole batch (checksum of checksums), and structural invariants of every frame."""
    import hashlib
    import orbhip
    ext, ora = _mk(gpu_ctx, 1000)
    imgs = orbhip.synth_frames(640, 480, 1024, seed=20241004)
    r1 = _dev_extract(gpu_ctx, ext, imgs, (0, 0))
    digest1 = hashlib.sha256(b"".join(hashlib.sha256(k.tobytes() + d.tobytes()).digest() for k, d, _ in r1)).hexdigest()
    for f in (0, 1, 511, 1023):
        kp, desc, mono = ora.extract(imgs[f], (0, 0))
        assert r1[f][2] == mono and r1[f][0].tobytes() == kp.tobytes() and r1[f][1].tobytes() == desc.tobytes(), f
    quota = ext.features_per_level()
    for k, d, m in r1:
        assert 800 < len(k) <= ext.max_keypoints and m == len(k)
        assert (np.diff(k["octave"]) >= 0).all()                               # level order when nothing is 'stereo'
        assert (np.bincount(k["octave"], minlength=8) <= quota + 2).all()      # octree never overshoots by more than 2
        assert (k["x"] >= 19).all() and (k["y"] >= 19).all()
    r2 = _dev_extract(gpu_ctx, ext, imgs, (0, 0))
    digest2 = hashlib.sha256(b"".join(hashlib.sha256(k.tobytes() + d.tobytes()).digest() for k, d, _ in r2)).hexdigest()
    assert digest1 == digest2
    ext.close()


def test_graph_mode_is_identical_and_tracks_new_input(gpu_ctx):
    """hipGraph replay of the 19 launches (small batches are launch bound): same bytes as the eager path, for new image
    content in the same call shape, after a shape change (re-capture) and through both entry points."""
    import orbhip
    imgs = orbhip.synth_frames(640, 480, 4, seed=314)
    eager = orbhip.Extractor(gpu_ctx, 1000, 1.2, 8, 20, 7)
    graph = orbhip.Extractor(gpu_ctx, 1000, 1.2, 8, 20, 7)
    graph.set_graph_mode(True)
    for k in (0, 1, 2, 0):                                   # host entry, batch 1, replayed with changing content
        a = eager.extract_host(imgs[k:k + 1], lap=(0, 1000))
        b = graph.extract_host(imgs[k:k + 1], lap=(0, 1000))
        assert a[0][0].tobytes() == b[0][0].tobytes() and a[0][1].tobytes() == b[0][1].tobytes() and a[0][2] == b[0][2]
    a = eager.extract_host(imgs, lap=(100, 300)); b = graph.extract_host(imgs, lap=(100, 300))      # new shape: re-capture
    for f in range(4):
        assert a[f][0].tobytes() == b[f][0].tobytes() and a[f][1].tobytes() == b[f][1].tobytes() and a[f][2] == b[f][2]
    for _ in range(2):                                       # device entry (staged into the extractor's buffer), replayed
        c = _dev_extract(gpu_ctx, graph, imgs, (100, 300))
        for f in range(4):
            assert a[f][0].tobytes() == c[f][0].tobytes() and a[f][1].tobytes() == c[f][1].tobytes() and a[f][2] == c[f][2]
    eager.close(); graph.close()


@pytest.mark.parametrize("rows_min", ["1", "16"])
def test_extract_random_geometries(gpu_ctx, rows_min, monkeypatch):
    """Seeded sweep over image sizes, level counts, scale factors (up to the 2.0 limit), thresholds and feature budgets:
    bit-exact final output for every combination (catches tile / apron / table-size assumptions)."""
    import orbhip
    import oracle_bind as ob
    monkeypatch.setenv("ORBHIP_ROWS_MIN_BATCH", rows_min)
    rng = np.random.default_rng(2024)
    done = 0
    while done < 10:
        nlev = int(rng.integers(1, 9)); scale = float(np.round(rng.uniform(1.1, 2.0), 2))
        if done == 0:
            nlev, scale = 3, 2.0
        smin = 75 * scale ** (nlev - 1)
        w = int(rng.integers(int(smin), int(smin) + 500)); h = int(rng.integers(int(smin * 0.8) + 1, int(smin * 0.8) + 400))
        wl, hl = w / scale ** (nlev - 1), h / scale ** (nlev - 1)
        if (wl - 33) < 0.6 * (hl - 31) or w > 1400 or h > 1100:      # nIni = round(W/H) must be >= 1 on every level (ORBextractor.cc:541)
            continue
        nfeat = int(rng.integers(50, 1500)); ini = int(rng.integers(8, 40)); mn = int(rng.integers(2, ini))
        lap = (int(rng.integers(0, w // 2)), int(rng.integers(w // 2, w + 50)))
        ext = orbhip.Extractor(gpu_ctx, nfeat, scale, nlev, ini, mn)
        ora = ob.OracleExtractor(nfeat, scale, nlev, ini, mn)
        imgs = orbhip.synth_frames(w, h, 2, seed=int(rng.integers(1, 1 << 30)))
        got = ext.extract_host(imgs, lap)
        for f in range(2):
            kp, desc, mono = ora.extract(imgs[f], lap)
            assert got[f][2] == mono and got[f][0].tobytes() == kp.tobytes() and got[f][1].tobytes() == desc.tobytes(), (w, h, nlev, scale, nfeat, ini, mn, lap)
        ext.close()
        done += 1


def test_two_extractors_in_two_threads(gpu_ctx):
    """"One ORBextractor instance per thread" (src/Frame.cc:109-110: left / right extraction threads): two contexts + extractors driven
    concurrently from two host threads give exactly what each gives alone."""
    import threading
    import orbhip
    imgs = [orbhip.synth_frames(640, 480, 12, seed=40 + k) for k in range(2)]
    ref = []
    for k in range(2):
        ext, _ = _mk(gpu_ctx)
        ref.append(ext.extract_host(imgs[k], (0, 0)))
        ext.close()
    ctxs = [orbhip.Context(0) for _ in range(2)]
    exts = [orbhip.Extractor(ctxs[k], 1000, 1.2, 8, 20, 7) for k in range(2)]
    out, errs = [None, None], []

    def run(k):
        try:
            for _ in range(6):                                   # several overlapping calls per thread
                out[k] = exts[k].extract_host(imgs[k], (0, 0))
        except BaseException as e:
            errs.append(e)
    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errs, errs
    for k in range(2):
        for f in range(12):
            assert out[k][f][0].tobytes() == ref[k][f][0].tobytes() and out[k][f][1].tobytes() == ref[k][f][1].tobytes() and out[k][f][2] == ref[k][f][2]
        exts[k].close(); ctxs[k].close()


def test_golden_fixture_without_oracle(gpu_ctx):
    """The HIP path against the committed vectors of tests/golden/orb_golden.npz (tools/gen_golden.py) -- no live oracle involved:
    keypoints (all 28 bytes each), descriptors, order and return value, through the host and the device entry point."""
    import os
    import orbhip
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "orb_golden.npz"))
    for name in ("a", "b"):
        img = g["img_" + name]
        lap = tuple(int(v) for v in g["lap_" + name])
        ext = orbhip.Extractor(gpu_ctx, int(g["nfeat_" + name]), 1.2, 8, 20, 7)
        for kp, desc, mono in (ext.extract_host(img, lap)[0], _dev_extract(gpu_ctx, ext, img[None], lap)[0]):
            assert mono == int(g["mono_" + name])
            assert kp.tobytes() == g["kp_" + name].tobytes(), "keypoints differ from the golden fixture " + name
            assert desc.tobytes() == g["desc_" + name].tobytes(), "descriptors differ from the golden fixture " + name
        ext.close()


def test_octree_5000_features_initializer_extractor(gpu_ctx):
    """The monocular initialiser's extractor, ORBextractor(5 * nFeatures, ...) (src/Tracking.cc:210-212): 5000 features per VGA
    frame.  The octree's LDS node arrays then exceed the 64 KB default (dynamic-LDS opt-in) -- bit-exact, every stage."""
    import orbhip
    ext, ora = _mk(gpu_ctx, 5000)
    imgs = orbhip.synth_frames(640, 480, 2, seed=5000)
    got = ext.extract_host(imgs, (0, 1000))
    n = sum(_compare_frame(ext, ora, imgs, f, (0, 1000), got) for f in range(2))
    assert n > 2 * 3500
    ext.close()


@pytest.mark.parametrize("w,h,nfeat,nlev", [(2000, 100, 20, 2), (1500, 110, 40, 1), (3000, 140, 12, 3)])
def test_wide_image_small_budget(gpu_ctx, w, h, nfeat, nlev):
    """nIni = round(width / height) roots (ORBextractor.cc:541) and a small budget: the first octree pass creates up to 4 * nIni
    nodes, more than quota + 16 -- the LDS node capacity must follow nIni, not only the quota."""
    import orbhip
    import oracle_bind as ob
    ext = orbhip.Extractor(gpu_ctx, nfeat, 1.2, nlev, 20, 7)
    ora = ob.OracleExtractor(nfeat, 1.2, nlev, 20, 7)
    imgs = orbhip.synth_frames(w, h, 2, seed=w + nfeat)
    got = ext.extract_host(imgs, (0, 1000))
    for f in range(2):
        assert _compare_frame(ext, ora, imgs, f, (0, 1000), got) >= nfeat // 2
    ext.close()


def test_hd_batch_properties(gpu_ctx):
    """BASELINE config #3's per-GPU shard at full size: 512 frames of 1920x1080, 2000 features (4096 over 8 GPUs): sampled frames
    bit-exact vs the oracle, run-to-run determinism (checksum of checksums), structural invariants of every frame."""
    import hashlib
    import orbhip
    import bench
    ext, ora = _mk(gpu_ctx, 2000)
    imgs = bench.synth_frames_parallel(orbhip, 1920, 1080, 512, 20241004, 0)
    r1 = _dev_extract(gpu_ctx, ext, imgs, (0, 0))
    digest1 = hashlib.sha256(b"".join(hashlib.sha256(k.tobytes() + d.tobytes()).digest() for k, d, _ in r1)).hexdigest()
    for f in (0, 31, 255, 511):
        kp, desc, mono = ora.extract(imgs[f], (0, 0))
        assert r1[f][2] == mono and r1[f][0].tobytes() == kp.tobytes() and r1[f][1].tobytes() == desc.tobytes(), f
    quota = ext.features_per_level()
    for k, d, m in r1:
        assert 1800 < len(k) <= ext.max_keypoints and m == len(k)
        assert (np.diff(k["octave"]) >= 0).all()
        assert (np.bincount(k["octave"], minlength=8) <= quota + 2).all()
        assert (k["x"] >= 19).all() and (k["y"] >= 19).all() and (k["x"] < 1920 - 19).all() and (k["y"] < 1080 - 19).all()
    r2 = _dev_extract(gpu_ctx, ext, imgs, (0, 0))
    digest2 = hashlib.sha256(b"".join(hashlib.sha256(k.tobytes() + d.tobytes()).digest() for k, d, _ in r2)).hexdigest()
    assert digest1 == digest2
    ext.close()


def test_4k_batch_properties(gpu_ctx):
    """north_star's third image size at batch scale (bench.py's `4k` leg): 64 frames of 3840x2160, 2000 features: sampled frames
    bit-exact vs the oracle, run-to-run determinism (checksum of checksums), structural invariants of every frame."""
    import hashlib
    import orbhip
    import bench
    ext, ora = _mk(gpu_ctx, 2000)
    imgs = bench.synth_frames_parallel(orbhip, 3840, 2160, 64, 20241005, 0)
    r1 = _dev_extract(gpu_ctx, ext, imgs, (0, 0))
    digest1 = hashlib.sha256(b"".join(hashlib.sha256(k.tobytes() + d.tobytes()).digest() for k, d, _ in r1)).hexdigest()
    for f in (0, 33, 63):
        kp, desc, mono = ora.extract(imgs[f], (0, 0))
        assert r1[f][2] == mono and r1[f][0].tobytes() == kp.tobytes() and r1[f][1].tobytes() == desc.tobytes(), f
    quota = ext.features_per_level()
    for k, d, m in r1:
        assert 1800 < len(k) <= ext.max_keypoints and m == len(k)
        assert (np.diff(k["octave"]) >= 0).all()
        assert (np.bincount(k["octave"], minlength=8) <= quota + 4).all()          # 16:9: nIni = 2 roots, up to 4 * nIni nodes after the first split
        assert (k["x"] >= 19).all() and (k["y"] >= 19).all() and (k["x"] < 3840 - 19).all() and (k["y"] < 2160 - 19).all()
    r2 = _dev_extract(gpu_ctx, ext, imgs, (0, 0))
    digest2 = hashlib.sha256(b"".join(hashlib.sha256(k.tobytes() + d.tobytes()).digest() for k, d, _ in r2)).hexdigest()
    assert digest1 == digest2
    ext.close()


@pytest.mark.parametrize("w,h", [(640, 480), (333, 277), (1241, 376)])
def test_row_streaming_and_tile_kernels_agree(gpu_ctx, w, h, monkeypatch):
    """The pyramid and the blur have two kernels each: the row-streaming ones (default) and the LDS-tile ones (scale factors whose
    4-column source span exceeds 8 bytes, levels below 28 rows).  Same bytes from both, on every level, against the oracle too.
    1241 x 376 (KITTI): width % 4 != 0, so level 1 takes the unaligned-load variant of the row kernel."""
    import orbhip
    imgs = orbhip.synth_frames(w, h, 2, seed=91)
    monkeypatch.setenv("ORBHIP_ROWS_MIN_BATCH", "1")              # (small batches take the tile kernels by default: shorter latency)
    ext, ora = _mk(gpu_ctx, 600)
    rows = ext.extract_host(imgs, (0, 0))
    lv_rows = [[ext.pyramid_level(f, l).copy() for l in range(ext.nlevels)] + [ext.blurred_level(f, l).copy() for l in range(ext.nlevels)] for f in range(2)]
    _compare_frame(ext, ora, imgs, 1, (0, 0), rows)
    ext.close()
    monkeypatch.setenv("ORBHIP_RESIZE_TILES", "1")
    monkeypatch.setenv("ORBHIP_BLUR_TILES", "1")
    ext2 = orbhip.Extractor(gpu_ctx, 600, 1.2, 8, 20, 7)
    tiles = ext2.extract_host(imgs, (0, 0))
    for f in range(2):
        lv = [ext2.pyramid_level(f, l) for l in range(ext2.nlevels)] + [ext2.blurred_level(f, l) for l in range(ext2.nlevels)]
        for a, b in zip(lv_rows[f], lv):
            np.testing.assert_array_equal(a, b)
        assert rows[f][0].tobytes() == tiles[f][0].tobytes() and rows[f][1].tobytes() == tiles[f][1].tobytes()
    ext2.close()
