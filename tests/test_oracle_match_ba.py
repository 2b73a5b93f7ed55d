"""CPU tests pinning the matching and BA oracles (no GPU)."""
import os
import numpy as np
import pytest

import oracle_bind as ob
import oracle_match_bind as om
import oracle_ba_bind as obb


# ------------------------------------------------------------------ matching
def test_swar_hamming_equals_popcount():
    rng = np.random.default_rng(0)
    for _ in range(500):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert om.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())
    assert om.descriptor_distance(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256


def test_bf2nn_against_numpy():
    rng = np.random.default_rng(1)
    A = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    B = rng.integers(0, 256, (75, 32), dtype=np.uint8)
    B[10] = B[3]                                              # tie: lower index must win
    idx, dist, acc = om.bf2nn(A, B, 0.7)
    D = np.unpackbits(A[:, None, :] ^ B[None, :, :], axis=2).sum(2)
    order = np.argsort(D, axis=1, kind="stable")
    np.testing.assert_array_equal(idx, order[:, :2])
    np.testing.assert_array_equal(dist, np.take_along_axis(D, order[:, :2], 1))
    np.testing.assert_array_equal(acc, (dist[:, 0].astype(np.float32) < dist[:, 1].astype(np.float32) * 0.7).astype(np.uint8))
    idx1, dist1, acc1 = om.bf2nn(A[:3], B[:1], 0.7)           # fewer than k train rows -> never accepted
    assert (idx1[:, 1] == -1).all() and (acc1 == 0).all()


def _kps(n, rng, w=640, h=480):
    kp = np.zeros(n, ob.KP_DTYPE)
    kp["x"] = rng.uniform(0, w, n).astype(np.float32)
    kp["y"] = rng.uniform(0, h, n).astype(np.float32)
    kp["octave"] = rng.integers(0, 8, n)
    kp["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    return kp


def test_features_in_area_against_bruteforce_and_grid_rounding():
    """GetFeaturesInArea: cell index uses round() (Frame.cc:718), the window uses floor/ceil and a
    strict |d| < r test; the result equals a brute-force scan re-ordered cell-major."""
    rng = np.random.default_rng(2)
    kp = _kps(800, rng)
    b = (0.0, 0.0, 640.0, 480.0)
    inv_w, inv_h = np.float32(64) / np.float32(640), np.float32(48) / np.float32(480)
    cx = np.floor(np.float32(kp["x"] * inv_w) + np.float32(0.5)).astype(int)      # round half away (x >= 0)
    cy = np.floor(np.float32(kp["y"] * inv_h) + np.float32(0.5)).astype(int)
    ingrid = (cx < 64) & (cy < 48)
    for (x, y, r, lo, hi) in [(320, 240, 100, 0, 0), (5, 5, 50, -1, -1), (630, 470, 40, 2, 5), (100, 400, 15, 0, 7),
                              (-200, 50, 100, 0, 0), (320, 240, 1000, -1, -1)]:
        got = om.features_in_area(kp, b, x, y, r, lo, hi)
        sel = ingrid & (np.abs(kp["x"] - np.float32(x)) < r) & (np.abs(kp["y"] - np.float32(y)) < r)
        if lo > 0 or hi >= 0:
            sel &= kp["octave"] >= lo
            if hi >= 0:
                sel &= kp["octave"] <= hi
        # window cell range
        c0 = max(0, int(np.floor((np.float32(x) - np.float32(r)) * inv_w))); c1 = min(63, int(np.ceil((np.float32(x) + np.float32(r)) * inv_w)))
        r0 = max(0, int(np.floor((np.float32(y) - np.float32(r)) * inv_h))); r1 = min(47, int(np.ceil((np.float32(y) + np.float32(r)) * inv_h)))
        sel &= (cx >= c0) & (cx <= c1) & (cy >= r0) & (cy <= r1)
        want = sorted(np.nonzero(sel)[0].tolist(), key=lambda i: (cx[i], cy[i], i))
        assert got.tolist() == want


def test_search_for_initialization_identity_and_properties():
    import orbhip
    img = orbhip.synth_frames(640, 480, 2, seed=9)
    e = ob.OracleExtractor()
    k0, d0, _ = e.extract(img[0], (0, 0))
    k1, d1, _ = e.extract(img[1], (0, 0))
    b = (0.0, 0.0, 640.0, 480.0)
    prev = np.stack([k0["x"], k0["y"]], 1)
    n, m12, pv = om.search_for_initialization(k0, d0, k0, d0, b, prev, 100, 0.9, True)
    lvl0 = np.nonzero(k0["octave"] == 0)[0]
    assert n == int((m12 >= 0).sum())
    assert (m12[k0["octave"] > 0] == -1).all()                    # only level-0 keypoints are matched
    assert (m12[lvl0][m12[lvl0] >= 0] == lvl0[m12[lvl0] >= 0]).all()   # a frame matches itself point for point
    n2, m2, pv2 = om.search_for_initialization(k0, d0, k1, d1, b, prev, 100, 0.9, True)
    good = np.nonzero(m2 >= 0)[0]
    assert n2 == len(good) and len(set(m2[good].tolist())) == len(good)       # one-to-one
    assert (k1["octave"][m2[good]] == 0).all()
    np.testing.assert_array_equal(pv2[good], np.stack([k1["x"][m2[good]], k1["y"][m2[good]]], 1))
    n3, _, _ = om.search_for_initialization(k0, d0, k1, d1, b, prev, 100, 0.9, False)
    assert n3 >= n2                                                  # rotation check only removes matches


# ------------------------------------------------------------------ BA
def test_se3_exp_matches_rodrigues_and_small_angle_branch():
    rng = np.random.default_rng(3)
    for scale in (1.0, 1e-3, 1e-7):
        u = rng.normal(0, scale, 6)
        q = np.zeros(4); t = np.zeros(3)
        ob.lib.orc_se3_exp(u.ctypes.data, q.ctypes.data, t.ctypes.data)
        w = u[:3]; th = np.linalg.norm(w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        if th < 1e-5:
            R = np.eye(3) + K + K @ K; V = R                        # se3quat.h:240-246 (sic)
        else:
            R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
            V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K
        x, y, z, w_ = q
        Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w_), 2 * (x * z + y * w_)],
                       [2 * (x * y + z * w_), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w_)],
                       [2 * (x * z - y * w_), 2 * (y * z + x * w_), 1 - 2 * (x * x + y * y)]])
        np.testing.assert_allclose(Rq, R / np.cbrt(np.linalg.det(R)) if th < 1e-5 else R, atol=1e-9)
        np.testing.assert_allclose(t, V @ u[3:], atol=1e-12)
        assert q[3] >= 0 and abs(np.linalg.norm(q) - 1) < 1e-12


@pytest.mark.parametrize("stereo", [0, 1])
def test_edge_jacobians_against_finite_differences(stereo):
    rng = np.random.default_rng(4 + stereo)
    q = rng.normal(0, 1, 4); q /= np.linalg.norm(q); q *= np.sign(q[3])
    pose = np.concatenate([q, rng.normal(0, 1, 3)])
    Xc = np.array([0.3, -0.2, 4.0])
    # choose X so that the camera-frame point is Xc
    e0 = np.zeros(3); Jx = np.zeros(9); Jt = np.zeros(18)
    R = np.zeros(9)
    x, y, z, w = q
    Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    X = Rm.T @ (Xc - pose[4:])
    obs = np.array([300.0, 200.0, 290.0])
    fx, fy, cx, cy, bf = 458.0, 457.0, 320.0, 240.0, 50.0

    def err(p, Xw):
        e = np.zeros(3)
        ob.lib.orc_ba_edge(np.ascontiguousarray(p).ctypes.data, np.ascontiguousarray(Xw).ctypes.data, obs.ctypes.data,
                           stereo, fx, fy, cx, cy, bf, e.ctypes.data, Jx.ctypes.data, Jt.ctypes.data)
        return e.copy()
    e0 = err(pose, X)
    D = 3 if stereo else 2
    JxA = Jx[:3 * D].reshape(D, 3).copy(); JtA = Jt[:6 * D].reshape(D, 6).copy()
    h = 1e-3 if stereo else 1e-6      # stereo residual quantises 1/z to float32: FD needs a coarse step
    Jx_num = np.zeros((D, 3)); Jt_num = np.zeros((D, 6))
    for k in range(3):
        d = np.zeros(3); d[k] = h
        Jx_num[:, k] = (err(pose, X + d)[:D] - err(pose, X - d)[:D]) / (2 * h)
    for k in range(6):
        d = np.zeros(6); d[k] = h
        pp = pose.copy(); pm = pose.copy()
        ob.lib.orc_se3_oplus(d.ctypes.data, pp.ctypes.data)
        ob.lib.orc_se3_oplus((-d).ctypes.data, pm.ctypes.data)
        Jt_num[:, k] = (err(pp, X)[:D] - err(pm, X)[:D]) / (2 * h)
    tol = 5e-2 if stereo else 1e-5        # stereo error uses a float32 1/z (types_six_dof_expmap.cpp:191)
    np.testing.assert_allclose(JxA, Jx_num, atol=tol, rtol=1e-4)
    np.testing.assert_allclose(JtA, Jt_num, atol=tol, rtol=1e-4)


def test_ba_recovers_noise_free_structure():
    import synth_ba
    g = synth_ba.make_graph(n_kf=8, n_pts=150, obs=5, seed=5, outlier_frac=0.0, pixel_noise=0.0)
    rc, poses, pts, out, st = obb.solve(g)
    assert rc == 0 and st["discarded"] == 0
    assert st["chi2_final"] < 1e-3 * st["chi2_initial"]
    # gauge is fixed by the two fixed keyframes: the optimum is the ground truth (up to float32 obs rounding)
    assert np.sqrt(np.mean((pts - g["points_gt"]) ** 2)) < 2e-3
    assert np.sqrt(np.mean((poses[:, 4:] - g["poses_gt"][:, 4:]) ** 2)) < 2e-3
    np.testing.assert_array_equal(poses[:2], g["poses0"][:2])      # fixed keyframes untouched
    assert out.sum() == 0


def test_ba_schedule_abort_and_discard():
    import synth_ba
    g = synth_ba.make_graph(n_kf=8, n_pts=120, obs=5, seed=6)
    rc, poses, pts, out, st = obb.solve(g)
    assert st["iterations_run"][0] <= 5 and st["iterations_run"][1] <= 10 and st["lm_trials"] >= sum(st["iterations_run"])
    assert st["chi2_final"] <= st["chi2_initial"]
    rc2, p2, x2, _, st2 = obb.solve(g, abort=np.ones(1, np.uint8))
    assert rc2 == -5                                               # Optimizer.cc:2041-2043
    np.testing.assert_array_equal(p2, g["poses0"])
    gbad = synth_ba.make_graph(n_kf=8, n_pts=120, obs=5, seed=7, outlier_frac=0.9)
    rc3, p3, x3, out3, st3 = obb.solve(gbad)
    assert st3["discarded"] == 1 and out3.sum() >= 0.5 * len(out3)   # Optimizer.cc:2177-2181
    np.testing.assert_array_equal(p3, gbad["poses0"])
    np.testing.assert_array_equal(x3, gbad["points0"])


def test_ba_golden_regression():
    import os
    import synth_ba
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = np.load(os.path.join(root, "tests", "golden", "ba_golden.npz"))
    g = {k[2:]: gold[k] for k in gold.files if k.startswith("g_")}
    for k in ("n_poses", "n_points", "n_edges"):
        g[k] = int(g[k])
    for k in ("fx", "fy", "cx", "cy", "bf"):
        g[k] = float(g[k])
    rc, poses, pts, out, st = obb.solve(g)
    np.testing.assert_allclose(poses, gold["poses"], atol=1e-9)
    np.testing.assert_allclose(pts, gold["points"], atol=1e-9)
    np.testing.assert_array_equal(out, gold["outlier"])
    assert st["iterations_run"] == gold["iterations_run"].tolist() and st["lm_trials"] == int(gold["lm_trials"])


def _sbp_python(q, dq, kp, d, u_right, bounds, tm, th_high, check_ori):
    """Independent restatement of ORBmatcher.cc:1965-2181 (Nleft == -1) on top of the GetFeaturesInArea oracle."""
    import oracle_match_bind as om
    tm = tm.copy(); tm[tm != -1] = -2
    nm = 0
    hist = [[] for _ in range(30)]
    for t in range(len(q)):
        cand = om.features_in_area(kp, bounds, float(q["u"][t]), float(q["v"][t]), float(q["radius"][t]),
                                   int(q["min_level"][t]), int(q["max_level"][t]))
        best, bi = 256, -1
        for i2 in cand:
            h = tm[i2]
            if h <= -2 or (h >= 0 and q["has_obs"][h]):
                continue
            if u_right is not None and u_right[i2] > 0 and abs(np.float32(q["ur"][t]) - np.float32(u_right[i2])) > q["radius"][t]:
                continue
            dist = int(np.unpackbits(dq[t] ^ d[i2]).sum())
            if dist < best:
                best, bi = dist, i2
        if best <= th_high:
            tm[bi] = t; nm += 1
            if check_ori:
                rot = np.float32(q["angle"][t]) - np.float32(kp["angle"][bi])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360.0))
                b = int(np.floor(np.float32(rot * np.float32(1.0 / 30)) + 0.5))    # roundf for non-negative values
                hist[0 if b == 30 else b].append(bi)
    if check_ori:
        sizes = [len(h) for h in hist]
        m1 = m2 = m3 = 0; i1 = i2_ = i3 = -1
        for i, s in enumerate(sizes):
            if s > m1: m3, m2, m1, i3, i2_, i1 = m2, m1, s, i2_, i1, i
            elif s > m2: m3, m2, i3, i2_ = m2, s, i2_, i
            elif s > m3: m3, i3 = s, i
        if m2 < np.float32(0.1) * np.float32(m1): i2_ = i3 = -1
        elif m3 < np.float32(0.1) * np.float32(m1): i3 = -1
        for i in range(30):
            if i not in (i1, i2_, i3):
                for bi in hist[i]:
                    tm[bi] = -1; nm -= 1
    return nm, tm


def make_sbp_case(rng, n, nq, with_stereo, dup=True):
    """Synthetic SearchByProjection case: keypoints, queries near them, duplicated descriptors (ties)."""
    import oracle_match_bind as om
    from oracle_bind import KP_DTYPE
    kp = np.zeros(n, KP_DTYPE)
    kp["x"] = rng.uniform(-10, 650, n).astype(np.float32); kp["y"] = rng.uniform(-10, 490, n).astype(np.float32)
    kp["angle"] = rng.uniform(0, 360, n).astype(np.float32); kp["octave"] = rng.integers(0, 8, n)
    base = rng.integers(0, 256, (max(n // 6, 1), 32), dtype=np.uint8)
    d = base[rng.integers(0, len(base), n)].copy() if dup else rng.integers(0, 256, (n, 32), dtype=np.uint8)
    d[:, 1] ^= rng.integers(0, 8, n).astype(np.uint8)
    q = np.zeros(nq, om.PROJ_QUERY_DTYPE)
    src = rng.integers(0, max(n, 1), nq) if n else np.zeros(nq, np.int64)
    if n:
        q["u"] = kp["x"][src] + rng.normal(0, 4, nq).astype(np.float32); q["v"] = kp["y"][src] + rng.normal(0, 4, nq).astype(np.float32)
        octv = kp["octave"][src]
        dq = d[src].copy(); dq[:, 2] ^= rng.integers(0, 4, nq).astype(np.uint8)
        q["angle"] = (kp["angle"][src] + rng.choice([0, 0, 0, 90, 200], nq) + rng.normal(0, 3, nq)).astype(np.float32) % np.float32(360)
    else:
        q["u"] = rng.uniform(0, 640, nq); q["v"] = rng.uniform(0, 480, nq); octv = rng.integers(0, 8, nq)
        dq = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    q["radius"] = (np.float32(15.0) * np.float32(1.2) ** octv.astype(np.float32)).astype(np.float32)
    mode = rng.integers(0, 3, nq)
    q["min_level"] = np.where(mode == 0, octv, np.where(mode == 1, 0, octv - 1))
    q["max_level"] = np.where(mode == 0, -1, np.where(mode == 1, octv, octv + 1))
    q["has_obs"] = rng.integers(0, 2, nq)
    q["ur"] = q["u"] - rng.uniform(0, 40, nq).astype(np.float32)
    ur = None
    if with_stereo:
        ur = np.where(rng.random(n) < 0.6, kp["x"] - rng.uniform(0, 40, n), -1).astype(np.float32)
    tm = np.where(rng.random(n) < 0.1, 7, -1).astype(np.int32)        # some keypoints already hold a map point
    return q, dq, kp, d, ur, tm


@pytest.mark.parametrize("with_stereo,check_ori", [(False, True), (True, True), (True, False)])
def test_search_by_projection_oracle_against_python(with_stereo, check_ori):
    import oracle_match_bind as om
    rng = np.random.default_rng(11 + with_stereo)
    bounds = (0.0, 0.0, 640.0, 480.0)
    for n, nq in ((0, 5), (40, 0), (300, 250), (500, 400)):
        q, dq, kp, d, ur, tm = make_sbp_case(rng, n, nq, with_stereo)
        n1, tm1 = om.search_by_projection(q, dq, kp, d, ur, bounds, tm, 100, check_ori)
        n2, tm2 = _sbp_python(q, dq, kp, d, ur, bounds, tm, 100, check_ori)
        assert n1 == n2
        np.testing.assert_array_equal(tm1, tm2)
        if n >= 300:
            assert (tm1 >= 0).sum() > 20


def _sbp_map_python(q, dq, kp, d, u_right, bounds, tm, th_high, ratio):
    """Independent restatement of ORBmatcher.cc:48-218 (Nleft == -1)."""
    import oracle_match_bind as om
    tm = tm.copy(); tm[tm != -1] = -2
    nm = 0
    for t in range(len(q)):
        cand = om.features_in_area(kp, bounds, float(q["u"][t]), float(q["v"][t]), float(q["radius"][t]),
                                   int(q["min_level"][t]), int(q["max_level"][t]))
        b1, l1, b2, l2, bi = 256, -1, 256, -1, -1
        for idx in cand:
            h = tm[idx]
            if h <= -2 or (h >= 0 and q["has_obs"][h]):
                continue
            if u_right is not None and u_right[idx] > 0 and abs(np.float32(q["ur"][t]) - np.float32(u_right[idx])) > q["radius"][t]:
                continue
            dist = int(np.unpackbits(dq[t] ^ d[idx]).sum())
            if dist < b1:
                b2, l2, b1, l1, bi = b1, l1, dist, int(kp["octave"][idx]), idx
            elif dist < b2:
                b2, l2 = dist, int(kp["octave"][idx])
        if b1 <= th_high:
            if l1 == l2 and np.float32(b1) > np.float32(ratio) * np.float32(b2):
                continue
            tm[bi] = t; nm += 1
    return nm, tm


@pytest.mark.parametrize("with_stereo", [False, True])
def test_search_by_projection_map_oracle_against_python(with_stereo):
    import oracle_match_bind as om
    rng = np.random.default_rng(31 + with_stereo)
    bounds = (0.0, 0.0, 640.0, 480.0)
    for n, nq in ((0, 5), (40, 0), (300, 250), (500, 400)):
        q, dq, kp, d, ur, tm = make_sbp_case(rng, n, nq, with_stereo)
        q["min_level"] = np.maximum(q["max_level"], 0) - 1; q["max_level"] = q["min_level"] + 1      # (level-1, level)
        for ratio in (0.8, 0.6):
            n1, tm1 = om.search_by_projection_map(q, dq, kp, d, ur, bounds, tm, 100, ratio)
            n2, tm2 = _sbp_map_python(q, dq, kp, d, ur, bounds, tm, 100, ratio)
            assert n1 == n2
            np.testing.assert_array_equal(tm1, tm2)
        if n >= 300:
            assert (tm1 >= 0).sum() > 20


# ------------------------------------------------------------------ PoseOptimization oracle (8f N1)
def _pose_err(a, b):
    qa, qb = a[:4], b[:4]
    return min(np.linalg.norm(qa - qb), np.linalg.norm(qa + qb)) + np.linalg.norm(a[4:] - b[4:])


@pytest.mark.parametrize("stereo_frac", [0.0, 0.5, 1.0])
def test_pose_optimization_recovers_noise_free_pose(stereo_frac):
    import oracle_ba_bind as ob
    import synth_ba
    p = synth_ba.make_pose_problem(5, n=300, stereo_frac=stereo_frac, outlier_frac=0.0, noise=False)
    r, pose, out, st = ob.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["cam"], p["pose0"])
    assert r == 300 and out.sum() == 0 and st["rounds"] == 4
    assert _pose_err(pose, p["pose_true"]) < 2e-5          # observations are float32-rounded pixels


def test_pose_optimization_flags_gross_outliers_and_small_cases():
    import oracle_ba_bind as ob
    import synth_ba
    p = synth_ba.make_pose_problem(7, n=600, stereo_frac=0.3, outlier_frac=0.15)
    r, pose, out, st = ob.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["cam"], p["pose0"])
    assert out[p["outlier_true"]].mean() > 0.98             # +-40 px never survives chi2 < 5.991
    assert out[~p["outlier_true"]].mean() < 0.15
    assert r == 600 - out.sum() == 600 - st["n_bad"]
    assert _pose_err(pose, p["pose_true"]) < 0.02
    # fewer than 3 correspondences: untouched, returns 0 (Optimizer.cc:1040-1041)
    r2, pose2, out2, st2 = ob.pose_optimization(p["Xw"][:2], p["obs"][:2], p["inv_sigma2"][:2], p["cam"], p["pose0"])
    assert r2 == 0 and st2["rounds"] == 0 and np.array_equal(pose2, p["pose0"])
    # fewer than 10 edges: one round only (Optimizer.cc:1147-1148)
    r3, pose3, out3, st3 = ob.pose_optimization(p["Xw"][:8], p["obs"][:8], p["inv_sigma2"][:8], p["cam"], p["pose0"])
    assert st3["rounds"] == 1 and 0 <= r3 <= 8


def test_pose_optimization_golden_regression():
    """Committed fixture (tests/golden/pose_golden.npz, made by tools/gen_golden.py): guards the oracle itself."""
    import os
    import oracle_ba_bind as ob
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_golden.npz"))
    for k in range(int(g["count"])):
        r, pose, out, st = ob.pose_optimization(g[f"Xw{k}"], g[f"obs{k}"], g[f"w{k}"], g[f"cam{k}"], g[f"pose0_{k}"])
        assert r == int(g[f"r{k}"])
        np.testing.assert_array_equal(out, g[f"out{k}"])
        np.testing.assert_allclose(pose, g[f"pose{k}"], rtol=0, atol=1e-9)


def test_ba_merge_variant_oracle():
    """Merge-LBA switches (Optimizer.cc:6255-6800): pass 2 minimises the plain chi2 of the kept edges only, nothing is
    discarded; the default schedule on the same graph keeps every edge and the Huber kernel."""
    import oracle_ba_bind as ob
    import synth_ba
    g = synth_ba.make_graph(n_kf=20, n_pts=300, obs=10, seed=41, outlier_frac=0.03)
    rc_d, poses_d, pts_d, out_d, st_d = ob.solve(g)
    rc_m, poses_m, pts_m, out_m, st_m = ob.solve(g, ob.merge_params())
    assert rc_d == 0 and rc_m == 0 and st_m["discarded"] == 0
    assert np.isfinite(st_m["chi2_final"]) and st_m["chi2_final"] < st_d["chi2_final"]
    assert out_m.sum() >= 0.03 * len(out_m)
    gbad = synth_ba.make_graph(n_kf=8, n_pts=120, obs=5, seed=43, outlier_frac=0.9)
    assert ob.solve(gbad)[4]["discarded"] == 1 and ob.solve(gbad, ob.merge_params())[4]["discarded"] == 0


def test_distinctive_descriptor_oracle_against_numpy():
    import oracle_match_bind as om
    rng = np.random.default_rng(9)
    for n in (1, 2, 3, 7, 20, 64, 130):
        base = rng.integers(0, 256, (1, 32), dtype=np.uint8)
        d = np.repeat(base, n, 0)
        flips = rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8)
        d ^= flips
        if n > 3:
            d[n // 2] = d[1]                                       # duplicates: ties between medians
        D = np.unpackbits(d[:, None, :] ^ d[None, :, :], axis=2).sum(2)
        med = np.sort(D, axis=1)[:, int(0.5 * (n - 1))]
        assert om.distinctive_descriptor(d) == int(np.argmin(med))
    assert om.distinctive_descriptor(np.zeros((0, 32), np.uint8)) == 0


def test_bow_transform_oracle_against_python():
    """DBoW2 tree descent restated: greedy nearest child per level, first minimum wins, nid at level L - levelsup."""
    import oracle_match_bind as om
    rng = np.random.default_rng(23)
    for k, L, ragged in ((10, 3, False), (4, 5, True), (3, 2, False)):
        voc = om.make_vocabulary(rng, k, L, ragged)
        for levelsup in (0, 1, L - 1, L, L + 2):
            for _ in range(40):
                leaf = int(rng.integers(1, len(voc["node_desc"])))
                f = voc["node_desc"][leaf] ^ (rng.integers(0, 256, 32, dtype=np.uint8) & rng.integers(0, 256, 32, dtype=np.uint8))
                node, level, nid = 0, 0, 0
                nid_level = L - levelsup
                while voc["child_start"][node + 1] > voc["child_start"][node]:
                    level += 1
                    ch = voc["child_ids"][voc["child_start"][node]:voc["child_start"][node + 1]]
                    dist = [int(np.unpackbits(f ^ voc["node_desc"][c]).sum()) for c in ch]
                    node = int(ch[int(np.argmin(dist))])
                    if level == nid_level:
                        nid = node
                got = om.bow_transform(f, voc, levelsup)
                assert got == (int(voc["node_word"][node]), float(voc["node_weight"][node]), nid)


# ------------------------------------------------------------------ KannalaBrandt8 camera (rows B2 / B3)
KB8 = (-0.0034, 0.0007, -0.0021, 0.0002)          # k1..k4 of the order of the TUM-VI calibrations (Examples/*/TUM_512.yaml)


def test_kb8_edge_jacobians_against_finite_differences():
    import ctypes as C
    import oracle_ba_bind as ob
    rng = np.random.default_rng(3)
    fx, fy, cx, cy = 190.9, 190.9, 254.9, 256.8
    k = np.array(KB8)
    for _ in range(20):
        q = rng.normal(0, 1, 4); q /= np.linalg.norm(q)
        if q[3] < 0: q = -q
        pose = np.concatenate([q, rng.normal(0, 1, 3)])
        Xc = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(1.5, 6)])
        # world point that maps to Xc under pose: X = R^T (Xc - t)
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        X = R.T @ (Xc - pose[4:])
        obs = np.array([250.0, 260.0, 0.0])
        e = np.zeros(3); Jx = np.zeros(9); Jt = np.zeros(18)
        ob.lib.orc_ba_edge_kb8(pose.ctypes.data, X.ctypes.data, obs.ctypes.data, fx, fy, cx, cy, k.ctypes.data, e.ctypes.data, Jx.ctypes.data, Jt.ctypes.data)
        h = 1e-4                                       # the projection runs through float atan2f: coarse steps, loose tolerance
        for a in range(3):
            Xp, Xm = X.copy(), X.copy(); Xp[a] += h; Xm[a] -= h
            ep, em = np.zeros(3), np.zeros(3)
            ob.lib.orc_ba_edge_kb8(pose.ctypes.data, Xp.ctypes.data, obs.ctypes.data, fx, fy, cx, cy, k.ctypes.data, ep.ctypes.data, Jx.copy().ctypes.data, Jt.copy().ctypes.data)
            ob.lib.orc_ba_edge_kb8(pose.ctypes.data, Xm.ctypes.data, obs.ctypes.data, fx, fy, cx, cy, k.ctypes.data, em.ctypes.data, Jx.copy().ctypes.data, Jt.copy().ctypes.data)
            np.testing.assert_allclose((ep[:2] - em[:2]) / (2 * h), Jx.reshape(3, 3)[:2, a], atol=0.15, rtol=5e-3)
        for a in range(6):
            d = np.zeros(6); d[a] = h
            pp, pm = pose.copy(), pose.copy()
            ob.lib.orc_se3_oplus(d.ctypes.data, pp.ctypes.data); ob.lib.orc_se3_oplus((-d).ctypes.data, pm.ctypes.data)
            ep, em = np.zeros(3), np.zeros(3)
            ob.lib.orc_ba_edge_kb8(pp.ctypes.data, X.ctypes.data, obs.ctypes.data, fx, fy, cx, cy, k.ctypes.data, ep.ctypes.data, Jx.copy().ctypes.data, Jt.copy().ctypes.data)
            ob.lib.orc_ba_edge_kb8(pm.ctypes.data, X.ctypes.data, obs.ctypes.data, fx, fy, cx, cy, k.ctypes.data, em.ctypes.data, Jx.copy().ctypes.data, Jt.copy().ctypes.data)
            np.testing.assert_allclose((ep[:2] - em[:2]) / (2 * h), Jt.reshape(3, 6)[:2, a], atol=0.15, rtol=5e-3)


def test_kb8_ba_and_pose_recover_noise_free_solution():
    import oracle_ba_bind as ob
    import synth_ba
    g = synth_ba.make_graph(n_kf=8, n_pts=150, obs=6, seed=61, outlier_frac=0.0, pixel_noise=0.0, kb8=KB8)
    rc, poses, pts, out, st = ob.solve(g)
    assert rc == 0 and st["discarded"] == 0 and out.sum() == 0
    assert np.abs(poses[:, 4:] - g["poses_gt"][:, 4:]).max() < 5e-4 and np.abs(pts - g["points_gt"]).max() < 2e-3
    p = synth_ba.make_pose_problem(62, n=300, outlier_frac=0.0, noise=False, kb8=KB8)
    r, pose, o, _ = ob.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["cam"], p["pose0"], kb8=KB8)
    assert r == 300 and _pose_err(pose, p["pose_true"]) < 5e-5


RIG2 = dict(Trl=(0.004, -0.012, 0.002, 0.99991, -0.101, 0.0007, 0.0012), cam=(190.4, 190.6, 252.7, 255.0), kb=(0.0031, 0.0007, -0.0019, 0.0003))


def _rig_graph_struct():
    import oracle_ba_bind as ob
    g = dict(n_poses=0, n_points=0, n_edges=0, pose_fixed=np.zeros(1, np.uint8), edge_pose=np.zeros(1, np.int32), edge_point=np.zeros(1, np.int32),
             edge_obs=np.zeros(3), edge_inv_sigma2=np.zeros(1), edge_stereo=np.zeros(1, np.uint8), fx=190.9, fy=190.9, cx=254.9, cy=256.8, bf=0.0,
             kb=KB8, rig2=RIG2)
    return ob.make_cgraph(g)


def test_tobody_edge_jacobians_against_finite_differences():
    """EdgeSE3ProjectXYZToBody (OptimizableTypes.h:112-141, .cpp:192-213): observation in the second camera of a rigid pair."""
    import ctypes as C
    import oracle_ba_bind as ob
    cg, keep = _rig_graph_struct()
    rng = np.random.default_rng(8)
    for _ in range(15):
        q = rng.normal(0, 1, 4); q /= np.linalg.norm(q)
        if q[3] < 0: q = -q
        pose = np.concatenate([q, rng.normal(0, 1, 3)])
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        Xc = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(1.5, 6)])
        X = R.T @ (Xc - pose[4:])
        obs = np.array([250.0, 260.0, 0.0])
        e = np.zeros(3); Jx = np.zeros(9); Jt = np.zeros(18)
        f = lambda P, XX, out: ob.lib.orc_ba_edge_tobody(C.byref(cg), P.ctypes.data, XX.ctypes.data, obs.ctypes.data, out.ctypes.data, np.zeros(9).ctypes.data, np.zeros(18).ctypes.data)
        ob.lib.orc_ba_edge_tobody(C.byref(cg), pose.ctypes.data, X.ctypes.data, obs.ctypes.data, e.ctypes.data, Jx.ctypes.data, Jt.ctypes.data)
        h = 1e-4
        for a in range(3):
            Xp, Xm = X.copy(), X.copy(); Xp[a] += h; Xm[a] -= h
            ep, em = np.zeros(3), np.zeros(3); f(pose, Xp, ep); f(pose, Xm, em)
            np.testing.assert_allclose((ep[:2] - em[:2]) / (2 * h), Jx.reshape(3, 3)[:2, a], atol=0.15, rtol=5e-3)
        for a in range(6):
            d = np.zeros(6); d[a] = h
            pp, pm = pose.copy(), pose.copy()
            ob.lib.orc_se3_oplus(d.ctypes.data, pp.ctypes.data); ob.lib.orc_se3_oplus((-d).ctypes.data, pm.ctypes.data)
            ep, em = np.zeros(3), np.zeros(3); f(pp, X, ep); f(pm, X, em)
            np.testing.assert_allclose((ep[:2] - em[:2]) / (2 * h), Jt.reshape(3, 6)[:2, a], atol=0.15, rtol=5e-3)


def test_rig_ba_recovers_noise_free_solution():
    import oracle_ba_bind as ob
    import synth_ba
    g = synth_ba.make_graph(n_kf=8, n_pts=150, obs=6, seed=81, outlier_frac=0.0, pixel_noise=0.0, kb8=KB8, rig2=RIG2)
    assert (g["edge_stereo"] == 2).sum() > 200
    rc, poses, pts, out, st = ob.solve(g)
    assert rc == 0 and st["discarded"] == 0 and out.sum() == 0
    assert np.abs(poses[:, 4:] - g["poses_gt"][:, 4:]).max() < 5e-4 and np.abs(pts - g["points_gt"]).max() < 2e-3


def test_pose_optimization_second_camera_oracle():
    import oracle_ba_bind as ob
    import synth_ba
    rig = dict(Trl=(0.004, -0.012, 0.002, 0.99991, -0.101, 0.0007, 0.0012), cam=(458.0, 457.0, 322.0, 238.0), kb=(0.0031, 0.0007, -0.0019, 0.0003))
    p = synth_ba.make_pose_problem(63, n=400, outlier_frac=0.0, noise=False, kb8=KB8, rig2=rig)
    assert 100 < p["right"].sum() < 300
    r, pose, o, _ = ob.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["cam"], p["pose0"], kb8=KB8, rig2=rig, right=p["right"])
    assert r == 400 and _pose_err(pose, p["pose_true"]) < 5e-5
    # the same observations WITHOUT the rig information (treated as left-camera) cannot be explained
    r2, pose2, o2, _ = ob.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["cam"], p["pose0"], kb8=KB8)
    assert r2 < 350


def test_fuse_search_oracle_against_python():
    """Search part of ORBmatcher::Fuse: window + octave gate + chi2 reprojection gate + nearest descriptor."""
    import oracle_match_bind as om
    rng = np.random.default_rng(77)
    bounds = (0.0, 0.0, 640.0, 480.0)
    sig = (np.float32(1.0) / (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2).astype(np.float32)
    for with_stereo in (False, True):
        for n, nq in ((0, 4), (50, 0), (250, 120)):
            q, dq, kp, d, ur, _ = make_sbp_case(rng, n, nq, with_stereo)
            q["min_level"] = np.maximum(q["max_level"], 0) - 1; q["max_level"] = q["min_level"] + 1
            q["radius"] = np.float32(3.0) * np.float32(1.2) ** q["max_level"].astype(np.float32)
            bi, bd = om.fuse_search(q, dq, kp, d, ur, sig, bounds)
            for t in range(nq):
                cand = om.features_in_area(kp, bounds, float(q["u"][t]), float(q["v"][t]), float(q["radius"][t]), -1, -1)
                best, besti = 256, -1
                for idx in cand:
                    lv = int(kp["octave"][idx])
                    if lv < q["min_level"][t] or lv > q["max_level"][t]:
                        continue
                    ex = np.float32(q["u"][t]) - np.float32(kp["x"][idx]); ey = np.float32(q["v"][t]) - np.float32(kp["y"][idx])
                    if ur is not None and ur[idx] >= 0:
                        er = np.float32(q["ur"][t]) - np.float32(ur[idx])
                        e2 = np.float32(np.float32(ex * ex + ey * ey) + er * er)
                        if float(np.float32(e2 * sig[lv])) > 7.8:
                            continue
                    else:
                        e2 = np.float32(ex * ex + ey * ey)
                        if float(np.float32(e2 * sig[lv])) > 5.99:
                            continue
                    dist = int(np.unpackbits(dq[t] ^ d[idx]).sum())
                    if dist < best:
                        best, besti = dist, idx
                assert (bi[t], bd[t]) == (besti, best), (t, bi[t], bd[t], besti, best)
            if n == 250:
                assert (bi >= 0).sum() > 8


def test_search_by_bow_oracle_against_python():
    """ORBmatcher::SearchByBoW(KeyFrame, Frame) restated; the Python model walks the two std::map-like dicts."""
    import oracle_match_bind as om
    rng = np.random.default_rng(5)
    for nk, nf in ((0, 20), (30, 0), (200, 260), (500, 450)):
        c = om.make_bow_case(rng, nk, nf)
        for ratio, ori in ((0.7, True), (0.9, False)):
            n1, m1 = om.search_by_bow(c, ratio, ori)
            fvk = {}; fvf = {}
            for i, x in enumerate(c["nid_k"]): fvk.setdefault(int(x), []).append(i)
            for i, x in enumerate(c["nid_f"]): fvf.setdefault(int(x), []).append(i)
            m = np.full(nf, -1, np.int64); nm = 0; hist = [[] for _ in range(30)]
            for node in sorted(set(fvk) & set(fvf)):
                for ri in fvk[node]:
                    if not c["valid"][ri]:
                        continue
                    b1, b2, bi = 256, 256, -1
                    for rj in fvf[node]:
                        if m[rj] >= 0:
                            continue
                        dist = int(np.unpackbits(c["d_k"][ri] ^ c["d_f"][rj]).sum())
                        if dist < b1: b2, b1, bi = b1, dist, rj
                        elif dist < b2: b2 = dist
                    if b1 <= 50 and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                        m[bi] = ri; nm += 1
                        if ori:
                            rot = np.float32(c["kp_k"]["angle"][ri]) - np.float32(c["kp_f"]["angle"][bi])
                            if rot < 0: rot = np.float32(rot + np.float32(360.0))
                            b = int(np.floor(np.float32(rot * np.float32(1.0 / 30)) + 0.5))
                            hist[0 if b == 30 else b].append(bi)
            if ori:
                sizes = [len(h) for h in hist]
                m1_, m2_, m3_ = 0, 0, 0; i1 = i2 = i3 = -1
                for i, sz in enumerate(sizes):
                    if sz > m1_: m3_, m2_, m1_, i3, i2, i1 = m2_, m1_, sz, i2, i1, i
                    elif sz > m2_: m3_, m2_, i3, i2 = m2_, sz, i2, i
                    elif sz > m3_: m3_, i3 = sz, i
                if m2_ < np.float32(0.1) * np.float32(m1_): i2 = i3 = -1
                elif m3_ < np.float32(0.1) * np.float32(m1_): i3 = -1
                for i in range(30):
                    if i not in (i1, i2, i3):
                        for bi in hist[i]:
                            m[bi] = -1; nm -= 1
            assert n1 == nm
            np.testing.assert_array_equal(m1, m)
        if nk >= 200:
            assert n1 > 20


def _py_search_for_triangulation(c, check_ori, mono):
    """Independent numpy/python restatement of ORBmatcher.cc:969-1210 + Pinhole.cpp:122-144 (node-major, like the reference)."""
    f32 = np.float32
    n1 = len(c["kp1"])
    F = c["F12"].reshape(3, 3)
    m = np.full(n1, -1, np.int32)
    hist = [[] for _ in range(30)]
    nodes2 = {}
    for j, nid in enumerate(c["nid2"]):
        nodes2.setdefault(int(nid), []).append(j)
    nodes1 = {}
    for i, nid in enumerate(c["nid1"]):
        nodes1.setdefault(int(nid), []).append(i)
    nm = 0
    for nid in sorted(nodes1):
        if nid not in nodes2:
            continue
        for i1 in nodes1[nid]:
            if c["mp1"][i1]:
                continue
            st1 = (not mono) and c["ur1"][i1] >= 0
            if c["only_stereo"] and not st1:
                continue
            k1 = c["kp1"][i1]
            best, bi = 50, -1
            for i2 in nodes2[nid]:
                if c["mp2"][i2]:
                    continue
                st2 = (not mono) and c["ur2"][i2] >= 0
                if c["only_stereo"] and not st2:
                    continue
                dist = int(np.unpackbits(c["d1"][i1] ^ c["d2"][i2]).sum())
                if dist > 50 or dist > best:
                    continue
                k2 = c["kp2"][i2]
                if not st1 and not st2:
                    ex = f32(c["ep"][0]) - k2["x"]; ey = f32(c["ep"][1]) - k2["y"]
                    if f32(f32(ex * ex) + f32(ey * ey)) < f32(f32(100) * c["scale"][k2["octave"]]):
                        continue
                a = f32(f32(f32(k1["x"] * F[0, 0]) + f32(k1["y"] * F[1, 0])) + F[2, 0])
                b = f32(f32(f32(k1["x"] * F[0, 1]) + f32(k1["y"] * F[1, 1])) + F[2, 1])
                cc = f32(f32(f32(k1["x"] * F[0, 2]) + f32(k1["y"] * F[1, 2])) + F[2, 2])
                num = f32(f32(f32(a * k2["x"]) + f32(b * k2["y"])) + cc)
                den = f32(f32(a * a) + f32(b * b))
                ok = den != 0 and float(f32(f32(num * num) / den)) < 3.84 * float(c["sigma2"][k2["octave"]])
                if ok or c["coarse"]:
                    bi, best = i2, dist
            if bi >= 0:
                m[i1] = bi; nm += 1
                if check_ori:
                    rot = f32(k1["angle"] - c["kp2"][bi]["angle"])
                    if rot < 0:
                        rot = f32(rot + f32(360))
                    v = float(f32(rot * f32(1.0 / 30)))
                    b_ = int(np.floor(abs(v) + 0.5) * np.sign(v))
                    if b_ == 30:
                        b_ = 0
                    hist[b_].append(i1)
    if check_ori:
        sz = [len(h) for h in hist]
        order = []
        m1 = m2 = m3 = 0; i1_ = i2_ = i3_ = -1
        for i, s_ in enumerate(sz):
            if s_ > m1:
                m3, m2, m1 = m2, m1, s_; i3_, i2_, i1_ = i2_, i1_, i
            elif s_ > m2:
                m3, m2 = m2, s_; i3_, i2_ = i2_, i
            elif s_ > m3:
                m3, i3_ = s_, i
        if m2 < f32(0.1) * f32(m1):
            i2_ = i3_ = -1
        elif m3 < f32(0.1) * f32(m1):
            i3_ = -1
        for i in range(30):
            if i not in (i1_, i2_, i3_):
                for j in hist[i]:
                    m[j] = -1; nm -= 1
    return nm, m


@pytest.mark.parametrize("stereo_frac,only_stereo,coarse,mono", [(0.0, False, False, True), (0.4, False, False, False), (0.5, True, False, False),
                                                                   (0.0, False, True, True)])
def test_search_for_triangulation_oracle_against_python(stereo_frac, only_stereo, coarse, mono):
    """CreateNewMapPoints matcher (ORBmatcher.cc:969-1210): C oracle vs an independent python restatement.  parity unpinned
    against the running reference (OpenCV / DBoW2 absent); F12 is an input, computed by the caller as Pinhole.cpp:124-127."""
    rng = np.random.default_rng(77)
    tot = 0
    for n1, n2 in ((0, 10), (25, 0), (150, 170), (400, 380)):
        c = om.make_tri_case(rng, n1, n2, 25, stereo_frac, only_stereo, coarse)
        for ori in (True, False):
            n_a, m_a = om.search_for_triangulation(c, ori, mono)
            n_b, m_b = _py_search_for_triangulation(c, ori, mono)
            assert n_a == n_b
            np.testing.assert_array_equal(m_a, m_b)
            tot += n_a
    assert tot > 40


def test_search_by_bow_kf_oracle_against_python():
    """ORBmatcher::SearchByBoW(KeyFrame, KeyFrame) (ORBmatcher.cc:827-967): C oracle vs a python walk of the two maps."""
    rng = np.random.default_rng(15)
    for nk, nf in ((0, 20), (30, 0), (200, 260), (500, 450)):
        c = om.make_bow_case(rng, nk, nf)
        c["valid2"] = (rng.random(nf) < 0.8).astype(np.uint8)
        for ratio, ori in ((0.75, True), (0.9, False)):
            n1, m1 = om.search_by_bow_kf(c, ratio, ori)
            fv1 = {}; fv2 = {}
            for i, x in enumerate(c["nid_k"]): fv1.setdefault(int(x), []).append(i)
            for i, x in enumerate(c["nid_f"]): fv2.setdefault(int(x), []).append(i)
            m = np.full(nk, -1, np.int64); taken = np.zeros(nf, bool); nm = 0; hist = [[] for _ in range(30)]
            for node in sorted(set(fv1) & set(fv2)):
                for ri in fv1[node]:
                    if not c["valid"][ri]:
                        continue
                    b1, b2, bi = 256, 256, -1
                    for rj in fv2[node]:
                        if taken[rj] or not c["valid2"][rj]:
                            continue
                        dist = int(np.unpackbits(c["d_k"][ri] ^ c["d_f"][rj]).sum())
                        if dist < b1: b2, b1, bi = b1, dist, rj
                        elif dist < b2: b2 = dist
                    if b1 < 50 and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                        m[ri] = bi; taken[bi] = True; nm += 1
                        if ori:
                            rot = np.float32(c["kp_k"]["angle"][ri]) - np.float32(c["kp_f"]["angle"][bi])
                            if rot < 0: rot = np.float32(rot + np.float32(360.0))
                            b = int(np.floor(np.float32(rot * np.float32(1.0 / 30)) + 0.5))
                            hist[0 if b == 30 else b].append(ri)
            if ori:
                sizes = [len(h) for h in hist]
                m1_, m2_, m3_ = 0, 0, 0; i1 = i2 = i3 = -1
                for i, sz in enumerate(sizes):
                    if sz > m1_: m3_, m2_, m1_, i3, i2, i1 = m2_, m1_, sz, i2, i1, i
                    elif sz > m2_: m3_, m2_, i3, i2 = m2_, sz, i2, i
                    elif sz > m3_: m3_, i3 = sz, i
                if m2_ < np.float32(0.1) * np.float32(m1_): i2 = i3 = -1
                elif m3_ < np.float32(0.1) * np.float32(m1_): i3 = -1
                for i in range(30):
                    if i not in (i1, i2, i3):
                        for ri in hist[i]:
                            m[ri] = -1; nm -= 1
            assert n1 == nm
            np.testing.assert_array_equal(m1, m)
        if nk >= 200:
            assert n1 > 20


def _sim3_case(rng, n, nq):
    q, dq, kp, d, _, tm = make_sbp_case(rng, n, nq, False)
    q["min_level"] = np.maximum(q["max_level"], 0) - 1; q["max_level"] = q["min_level"] + 1        # (pred-1, pred)
    q["radius"] = np.float32(8.0) * np.float32(1.2) ** q["max_level"].astype(np.float32)
    q["has_obs"] = 1; q["ur"] = -1
    return q, dq, kp, d, tm


def test_sim3_searches_map_onto_the_windowed_oracles():
    """The Sim3 SearchByProjection overloads (ORBmatcher.cc:477-708) are the last-frame claim-rule search with every query owning
    observations, no uRight gate, no rotation histogram and th_high = floor(TH_LOW*ratioHamming); the per-point searches of
    SearchBySim3 (:1813-1851) and Fuse(KF, Scw) (:1687-1720) are the Fuse search with the chi2 gates open (zero inverse sigma
    table).  Checked here oracle against oracle; tests/test_gpu_match.py runs the same mapping through the kernels."""
    rng = np.random.default_rng(21)
    bounds = (0.0, 0.0, 640.0, 480.0)
    tot = 0
    for n, nq in ((0, 5), (40, 0), (300, 250), (600, 500)):
        q, dq, kp, d, tm = _sim3_case(rng, n, nq)
        for ratio in (1.0, 0.9, 0.5):
            n_a, m_a = om.search_by_projection_sim3(q, dq, kp, d, bounds, tm, ratio)
            n_b, m_b = om.search_by_projection(q, dq, kp, d, None, bounds, tm, int(np.floor(np.float32(50) * np.float32(ratio))), False)
            assert n_a == n_b
            np.testing.assert_array_equal(np.where(tm != -1, -2, m_a), m_b)
            tot += n_a
        bi_a, bd_a = om.window_best(q, dq, kp, d, bounds)
        bi_b, bd_b = om.fuse_search(q, dq, kp, d, None, np.zeros(8, np.float32), bounds)
        np.testing.assert_array_equal(bi_a, bi_b)
        np.testing.assert_array_equal(np.where(bi_a < 0, 256, bd_a), bd_b)
    assert tot > 150


EUROC_K = (458.654, 457.296, 367.215, 248.375)
EUROC_DIST = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)      # Examples/Monocular/EuRoC.yaml:14-17


def test_undistort_keypoints_oracle():
    """Frame::UndistortKeyPoints (Frame.cc:738-771): oracle vs a numpy float64 restatement of OpenCV's 5-iteration inverse Brown
    model, the forward distortion of the result lands back on the input (5 iterations: < 0.5 px at the corners), zero k1 copies.
    parity unpinned against OpenCV itself (absent here)."""
    from oracle_bind import KP_DTYPE
    rng = np.random.default_rng(3)
    n = 500
    kp = np.zeros(n, KP_DTYPE)
    kp["x"] = rng.uniform(0, 752, n).astype(np.float32); kp["y"] = rng.uniform(0, 480, n).astype(np.float32)
    kp["octave"] = rng.integers(0, 8, n); kp["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    for dist in (EUROC_DIST, EUROC_DIST + (0.01,)):
        out = om.undistort_keypoints(kp, EUROC_K, dist)
        fx, fy, cx, cy = (float(np.float32(v)) for v in EUROC_K)
        k = [float(np.float32(v)) for v in dist] + [0.0] * (5 - len(dist))
        x0 = (kp["x"].astype(np.float64) - cx) * (1.0 / fx); y0 = (kp["y"].astype(np.float64) - cy) * (1.0 / fy)
        x, y = x0.copy(), y0.copy()
        for _ in range(5):
            r2 = x * x + y * y
            ic = 1.0 / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
            dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x); dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
            x = (x0 - dx) * ic; y = (y0 - dy) * ic
        np.testing.assert_array_equal(out["x"], (fx * x + cx).astype(np.float32))
        np.testing.assert_array_equal(out["y"], (fy * y + cy).astype(np.float32))
        for f in ("size", "angle", "response", "octave", "class_id"):
            np.testing.assert_array_equal(out[f], kp[f])
        # forward model on the undistorted points
        xu = (out["x"].astype(np.float64) - cx) / fx; yu = (out["y"].astype(np.float64) - cy) / fy
        r2 = xu * xu + yu * yu
        cd = 1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2
        xd = xu * cd + 2 * k[2] * xu * yu + k[3] * (r2 + 2 * xu * xu); yd = yu * cd + k[2] * (r2 + 2 * yu * yu) + 2 * k[3] * xu * yu
        err = np.hypot(fx * xd + cx - kp["x"], fy * yd + cy - kp["y"])
        assert err.max() < 0.5 and np.median(err) < 0.01, (err.max(), np.median(err))   # 5 iterations: corners converge to ~0.3 px
    same = om.undistort_keypoints(kp, EUROC_K, (0.0, 0.1, 0.0, 0.0))
    assert same.tobytes() == kp.tobytes()


def test_assign_features_to_grid_oracle():
    """Frame::AssignFeaturesToGrid (Frame.cc:377-408) as a CSR: every in-grid keypoint sits in the cell PosInGrid rounds it to,
    cells keep index order, and GetFeaturesInArea over the whole image returns exactly the CSR's items."""
    rng = np.random.default_rng(4)
    bounds = (-12.5, -9.0, 760.0, 490.0)
    for n in (0, 1, 300, 2500):
        _, _, kp, _, _, _ = make_sbp_case(rng, n, 0, False)
        cs, it = om.assign_features_to_grid(kp, bounds)
        iw = np.float32(64) / (np.float32(bounds[2]) - np.float32(bounds[0])); ih = np.float32(48) / (np.float32(bounds[3]) - np.float32(bounds[1]))
        cells = {}
        for i in range(n):
            vx = float(np.float32(np.float32(kp["x"][i] - np.float32(bounds[0])) * iw)); vy = float(np.float32(np.float32(kp["y"][i] - np.float32(bounds[1])) * ih))
            px = int(np.floor(abs(vx) + 0.5) * np.sign(vx)); py = int(np.floor(abs(vy) + 0.5) * np.sign(vy))
            if 0 <= px < 64 and 0 <= py < 48:
                cells.setdefault(px * 48 + py, []).append(i)
        assert cs[0] == 0 and cs[-1] == sum(len(v) for v in cells.values()) == len(it)
        for c in range(64 * 48):
            assert list(it[cs[c]:cs[c + 1]]) == cells.get(c, [])


def _bow_inputs(rng, n, n_words=300, n_nodes=40, stop_frac=0.1):
    wid = rng.integers(0, n_words, n).astype(np.int32) * 7 + 3
    nid = (rng.integers(0, n_nodes, n) * 5 + 11).astype(np.int32)
    w = rng.uniform(0.1, 9.0, n)
    w[rng.random(n) < stop_frac] = 0.0                           # stopped words
    return wid, w, nid


def test_bow_vectors_oracle_against_python():
    """TemplatedVocabulary::transform's BowVector / FeatureVector (TemplatedVocabulary.h:1139-1208): the oracle's sorted arrays vs
    python dicts replaying addWeight / addFeature in feature order and the sequential L1 norm."""
    rng = np.random.default_rng(8)
    for n in (0, 1, 50, 700):
        wid, w, nid = _bow_inputs(rng, n)
        ni, ns, ft, bw, bv = om.bow_vectors(wid, w, nid)
        v = {}; fv = {}
        for i in range(n):
            if w[i] > 0:
                v[int(wid[i])] = v.get(int(wid[i]), 0.0) + float(w[i])
                fv.setdefault(int(nid[i]), []).append(i)
        norm = 0.0
        for k in sorted(v):
            norm += abs(v[k])
        assert list(bw) == sorted(v)
        np.testing.assert_array_equal(bv, np.array([v[k] / norm for k in sorted(v)], np.float64) if v else np.zeros(0))
        assert list(ni) == sorted(fv)
        for j, k in enumerate(sorted(fv)):
            assert list(ft[ns[j]:ns[j + 1]]) == fv[k]
        if n >= 50:
            assert abs(bv.sum() - 1.0) < 1e-12


def _load_match_golden():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "match_golden.npz"))
    return g, tuple(float(v) for v in g["bounds"])


def test_matcher_golden_regression():
    """Committed fixture (tests/golden/match_golden.npz, made by tools/gen_golden.py): the matcher oracles must not drift."""
    g, bounds = _load_match_golden()
    n0, m0 = om.search_by_projection(g["sbp_q"], g["sbp_dq"], g["sbp_kp"], g["sbp_d"], g["sbp_ur"], bounds, g["sbp_tm"], 100, True)
    assert n0 == int(g["sbp_n"]); np.testing.assert_array_equal(m0, g["sbp_m"])
    n1, m1 = om.search_by_projection_map(g["sbp_q"], g["sbp_dq"], g["sbp_kp"], g["sbp_d"], g["sbp_ur"], bounds, g["sbp_tm"], 100, 0.8)
    assert n1 == int(g["map_n"]); np.testing.assert_array_equal(m1, g["map_m"])
    bi, bd = om.fuse_search(g["fuse_q"], g["sbp_dq"], g["sbp_kp"], g["sbp_d"], g["sbp_ur"], g["fuse_sig"], bounds)
    np.testing.assert_array_equal(bi, g["fuse_bi"]); np.testing.assert_array_equal(bd, g["fuse_bd"])
    c = {k[4:]: g[k] for k in g.files if k.startswith("bow_") and k not in ("bow_n", "bow_m")}
    nb, mb = om.search_by_bow(c, 0.7, True); nk, mk = om.search_by_bow_kf(c, 0.75, True)
    assert nb == int(g["bow_n"]) and nk == int(g["bowkf_n"])
    np.testing.assert_array_equal(mb, g["bow_m"]); np.testing.assert_array_equal(mk, g["bowkf_m"])
    t = {k[4:]: g[k] for k in g.files if k.startswith("tri_") and k not in ("tri_n", "tri_m")}
    t["ep"] = tuple(float(v) for v in t["ep"]); t["only_stereo"] = bool(t["only_stereo"]); t["coarse"] = bool(t["coarse"])
    nt, mt = om.search_for_triangulation(t, True, False)
    assert nt == int(g["tri_n"]); np.testing.assert_array_equal(mt, g["tri_m"])
    un = om.undistort_keypoints(g["sbp_kp"], EUROC_K, EUROC_DIST)
    assert un.tobytes() == g["un_kp"].tobytes()
    cs, it = om.assign_features_to_grid(un, bounds)
    np.testing.assert_array_equal(cs, g["grid_cs"]); np.testing.assert_array_equal(it, g["grid_it"])
    ni, ns, ft, bw, bv = om.bow_vectors(g["bv_wid"], g["bv_w"], g["bv_nid"])
    np.testing.assert_array_equal(ni, g["bv_ni"]); np.testing.assert_array_equal(ft, g["bv_ft"]); np.testing.assert_array_equal(bw, g["bv_bw"])
    assert bv.tobytes() == g["bv_bv"].tobytes()


# ------------------------------------------------------------------ two-camera rig frames (Nleft != -1): independent models
def make_rig_case(rng, nleft, nright, npts, mode):
    """A rig frame (left | right keypoints, cross links between the cameras) and the query list the reference's loop would issue:
    per point a left-camera query and / or a right-camera query (has_obs bit 1), in order."""
    import oracle_match_bind as om
    from oracle_bind import KP_DTYPE
    n = nleft + nright
    kp = np.zeros(n, KP_DTYPE)
    kp["x"] = rng.uniform(-5, 520, n).astype(np.float32); kp["y"] = rng.uniform(-5, 520, n).astype(np.float32)
    kp["angle"] = rng.uniform(0, 360, n).astype(np.float32); kp["octave"] = rng.integers(0, 8, n)
    base = rng.integers(0, 256, (max(n // 6, 1), 32), dtype=np.uint8)
    d = base[rng.integers(0, len(base), n)].copy(); d[:, 1] ^= rng.integers(0, 8, n).astype(np.uint8)
    mirror = np.full(n, -1, np.int32)                       # stereo matches between the cameras (mvLeftToRightMatch / mvRightToLeftMatch)
    k = min(nleft, nright) // 3
    if k:
        li = rng.choice(nleft, k, replace=False); ri = rng.choice(nright, k, replace=False) + nleft
        mirror[li] = ri; mirror[ri] = li
    tm = rng.choice([-1, -1, -1, 5], n).astype(np.int32)   # a few keypoints already hold a map point with observations
    q, dq = [], []
    for _ in range(npts):
        obs = int(rng.integers(0, 2))
        for cam in (0, 1):
            if rng.random() < (0.8 if cam == 0 else 0.6):
                lo, hi = (0, nleft) if cam == 0 else (nleft, n)
                if hi <= lo:
                    continue
                s = int(rng.integers(lo, hi))
                octv = int(kp["octave"][s])
                m = int(rng.integers(0, 3))
                lv = (octv, -1) if (mode == 0 and m == 0) else (0, octv) if (mode == 0 and m == 1) else (octv - 1, octv + (1 if mode == 0 else 0))
                dd = d[s].copy(); dd[2] ^= int(rng.integers(0, 4))
                q.append((kp["x"][s] + rng.normal(0, 4), kp["y"][s] + rng.normal(0, 4), np.float32(15.0) * np.float32(1.2) ** np.float32(octv), -1.0,
                          (kp["angle"][s] + rng.choice([0, 0, 0, 90, 200]) + rng.normal(0, 3)) % 360, lv[0], lv[1], obs | (cam << 1)))
                dq.append(dd)
    return np.array(q, om.PROJ_QUERY_DTYPE), np.array(dq, np.uint8).reshape(-1, 32), kp, d, mirror, tm


def _rig_sbp_python(mode, q, dq, kp, d, nleft, mirror, bounds, tm, th_high, ratio, check_ori):
    """ORBmatcher.cc:48-218 / 1965-2181 with Nleft != -1, written against GetFeaturesInArea(..., bRight) of each camera's own grid."""
    import oracle_match_bind as om
    tm = tm.copy(); tm[tm != -1] = -2
    nm = 0
    hist = [[] for _ in range(30)]
    halves = (kp[:nleft], kp[nleft:])
    for t in range(len(q)):
        cam = (int(q["has_obs"][t]) >> 1) & 1
        off = nleft if cam else 0
        cand = om.features_in_area(halves[cam], bounds, float(q["u"][t]), float(q["v"][t]), float(q["radius"][t]), int(q["min_level"][t]), int(q["max_level"][t]))
        best = best2 = 256; lv = lv2 = -1; bi = -1
        for i2 in cand:
            g = int(i2) + off
            h = tm[g]
            if h <= -2 or (h >= 0 and (q["has_obs"][h] & 1)):
                continue
            dist = int(np.unpackbits(dq[t] ^ d[g]).sum())
            if dist < best:
                best2, best, lv2, lv, bi = best, dist, lv, int(kp["octave"][g]), g
            elif dist < best2:
                lv2, best2 = int(kp["octave"][g]), dist
        if best > th_high:
            continue
        if mode == 1:
            if lv == lv2 and best > np.float32(ratio) * np.float32(best2):
                continue
            tm[bi] = t; nm += 1
            if mirror is not None and mirror[bi] >= 0:
                tm[mirror[bi]] = t; nm += 1
        else:
            tm[bi] = t; nm += 1
            if check_ori:
                rot = np.float32(q["angle"][t]) - np.float32(kp["angle"][bi])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360.0))
                b = int(np.round(np.float32(rot * np.float32(1.0 / 30))))
                hist[0 if b == 30 else b].append(bi)
    if mode == 0 and check_ori:
        sizes = [len(h) for h in hist]
        order = sorted(range(30), key=lambda i: -sizes[i])
        m1, m2, m3 = order[0], order[1], order[2]
        keep = {m1}
        if sizes[m2] >= 0.1 * sizes[m1]:
            keep.add(m2)
            if sizes[m3] >= 0.1 * sizes[m1]:
                keep.add(m3)
        # ComputeThreeMaxima picks the first of equal sizes; sorted() with a stable key does the same
        for i in range(30):
            if i not in keep:
                for j in hist[i]:
                    tm[j] = -1; nm -= 1
    return nm, tm


@pytest.mark.parametrize("mode", [0, 1])
def test_rig_search_by_projection_oracle_against_python(mode):
    import oracle_match_bind as om
    rng = np.random.default_rng(500 + mode)
    bounds = (0.0, 0.0, 512.0, 512.0)
    tot = 0
    for nleft, nright, npts in ((120, 100, 90), (300, 340, 260), (40, 0, 30)):
        q, dq, kp, d, mirror, tm = make_rig_case(rng, nleft, nright, npts, mode)
        n_ref, tm_ref = om.search_by_projection_rig(mode, q, dq, kp, d, nleft, mirror if mode == 1 else None, bounds, tm, 100, 0.8, True)
        n_py, tm_py = _rig_sbp_python(mode, q, dq, kp, d, nleft, mirror if mode == 1 else None, bounds, tm, 100, 0.8, True)
        assert n_ref == n_py
        np.testing.assert_array_equal(tm_ref, tm_py)
        tot += n_ref
        if nright:
            assert (tm_ref[nleft:] >= 0).sum() > 5           # the right camera's keypoints do get matched
    assert tot > 150


def test_rig_search_by_bow_oracle_against_python():
    """SearchByBoW on a rig frame: per keyframe feature a best / second best per camera, the right camera's best rides on the left
    test's TH_LOW branch without a ratio test (ORBmatcher.cc:338-359, 393-425), restated with plain Python containers."""
    import oracle_match_bind as om
    rng = np.random.default_rng(77)
    for nk, nf, nn in ((150, 200, 25), (400, 520, 60)):
        c = om.make_bow_case(rng, nk, nf, nn)
        nleft = nf * 3 // 5
        n_ref, m_ref = om.search_by_bow_rig(c, nleft, 0.7, True)
        # python model
        fk, ff = {}, {}
        for i, v in enumerate(c["nid_k"]):
            fk.setdefault(int(v), []).append(i)
        for j, v in enumerate(c["nid_f"]):
            ff.setdefault(int(v), []).append(j)
        m = np.full(nf, -1, np.int64); nm = 0; hist = [[] for _ in range(30)]
        for node in sorted(set(fk) & set(ff)):
            for ri in fk[node]:
                if not c["valid"][ri]:
                    continue
                b = [256, 256]; b2 = [256, 256]; bi = [-1, -1]
                for rj in ff[node]:
                    if m[rj] >= 0:
                        continue
                    dist = int(np.unpackbits(c["d_k"][ri] ^ c["d_f"][rj]).sum())
                    s = 0 if rj < nleft else 1
                    if dist < b[s]:
                        b2[s], b[s], bi[s] = b[s], dist, rj
                    elif dist < b2[s]:
                        b2[s] = dist
                if b[0] <= 50:
                    picks = []
                    if np.float32(b[0]) < np.float32(0.7) * np.float32(b2[0]):
                        picks.append(bi[0])
                    if b[1] <= 50:
                        picks.append(bi[1])
                    for j in picks:
                        m[j] = ri; nm += 1
                        rot = np.float32(c["kp_k"]["angle"][ri]) - np.float32(c["kp_f"]["angle"][j])
                        if rot < 0:
                            rot = np.float32(rot + np.float32(360.0))
                        bb = int(np.round(np.float32(rot * np.float32(1.0 / 30))))
                        hist[0 if bb == 30 else bb].append(j)
        sizes = [len(h) for h in hist]
        order = sorted(range(30), key=lambda i: -sizes[i])
        keep = {order[0]}
        if sizes[order[1]] >= 0.1 * sizes[order[0]]:
            keep.add(order[1])
            if sizes[order[2]] >= 0.1 * sizes[order[0]]:
                keep.add(order[2])
        for i in range(30):
            if i not in keep:
                for j in hist[i]:
                    m[j] = -1; nm -= 1
        assert n_ref == nm and n_ref > 20
        np.testing.assert_array_equal(m_ref, m)
        assert (m_ref[nleft:] >= 0).sum() > 3


# ------------------------------------------------------------------ KannalaBrandt8 epipolar constraint (SearchForTriangulation, rig / fisheye)
def _kb8_unproject_np(p, u, v):
    """KannalaBrandt8::unproject in double (independent model): Newton on theta_d = theta (1 + k1 theta^2 + ...)"""
    x, y = (u - p[2]) / p[0], (v - p[3]) / p[1]
    td = min(max(-np.pi / 2, np.hypot(x, y)), np.pi / 2)
    s = 1.0
    if td > 1e-8:
        th = td
        for _ in range(50):
            f = th * (1 + p[4] * th ** 2 + p[5] * th ** 4 + p[6] * th ** 6 + p[7] * th ** 8) - td
            fp = 1 + 3 * p[4] * th ** 2 + 5 * p[5] * th ** 4 + 7 * p[6] * th ** 6 + 9 * p[7] * th ** 8
            th -= f / fp
        s = np.tan(th) / td
    return np.array([x * s, y * s, 1.0])


def _triangulate_matches_np(cam1, cam2, p1, p2, R12, t12, s1, s2):
    """KannalaBrandt8::TriangulateMatches (KannalaBrandt8.cpp:334-401) in double with numpy's SVD; returns (z1 or -1, x3D, margin) where
    margin is the smallest relative distance of any of its tests from its threshold."""
    import oracle_match_bind as om
    r1, r2 = _kb8_unproject_np(cam1, *p1), _kb8_unproject_np(cam2, *p2)
    r21 = R12 @ r2
    cosp = r1 @ r21 / (np.linalg.norm(r1) * np.linalg.norm(r21))
    margins = [abs(cosp - 0.9998) / 0.9998]
    if cosp > 0.9998:
        return -1.0, None, min(margins)
    R21 = R12.T; t21 = -R21 @ t12
    T1 = np.hstack([np.eye(3), np.zeros((3, 1))]); T2 = np.hstack([R21, t21[:, None]])
    A = np.stack([r1[0] * T1[2] - T1[0], r1[1] * T1[2] - T1[1], r2[0] * T2[2] - T2[0], r2[1] * T2[2] - T2[1]])
    v = np.linalg.svd(A)[2][3]
    X = v[:3] / v[3]
    z1 = X[2]; z2 = R21[2] @ X + t21[2]
    margins += [abs(z1), abs(z2)]
    if z1 <= 0 or z2 <= 0:
        return -1.0, X, min(margins)
    e1 = om.kb8_project_np((1, cam1), X) - np.asarray(p1); e2 = om.kb8_project_np((1, cam2), R21 @ X + t21) - np.asarray(p2)
    margins += [abs(e1 @ e1 - 5.991 * s1) / (5.991 * s1), abs(e2 @ e2 - 5.991 * s2) / (5.991 * s2)]
    if e1 @ e1 > 5.991 * s1 or e2 @ e2 > 5.991 * s2:
        return -1.0, X, min(margins)
    return z1, X, min(margins)


def test_kb8_triangulate_matches_oracle_against_numpy_model():
    """The oracle's float restatement (one-sided Jacobi SVD as OpenCV's, fixed sincos / atan2 sequences) against an independent double
    model with numpy's SVD: same accept / reject decision wherever no test sits within 1e-3 of its threshold, triangulated point
    within 1e-3 relative.  Covers inliers, gross outliers, low parallax, points behind a camera."""
    import oracle_match_bind as om
    rng = np.random.default_rng(2718)
    cam1 = np.array([190.9, 190.8, 254.9, 256.8, 0.0034, 0.0007, -0.0020, 0.0002]); cam2 = np.array([190.4, 190.6, 252.7, 255.0, 0.0031, 0.0009, -0.0019, 0.0003])
    decided = accepted = 0
    for it in range(600):
        R12 = om._rot(rng.normal(size=3), rng.uniform(0, 0.3)); t12 = rng.normal(size=3) * rng.choice([0.02, 0.3, 1.0])
        X2 = np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(0.5, 12)])
        X1 = R12 @ X2 + t12
        if X1[2] < 0.2:
            continue
        p1 = om.kb8_project_np((1, cam1), X1) + rng.normal(0, 0.3, 2)
        p2 = om.kb8_project_np((1, cam2), X2) + rng.normal(0, 0.3, 2) * rng.choice([1, 1, 30])
        s1, s2 = float(np.float32(1.2) ** (2 * rng.integers(0, 8))), float(np.float32(1.2) ** (2 * rng.integers(0, 8)))
        R32, t32 = R12.astype(np.float32), t12.astype(np.float32)
        p1f, p2f = p1.astype(np.float32), p2.astype(np.float32)
        z, x = om.kb8_triangulate_matches(1, cam1, 1, cam2, p1f, p2f, R32, t32, s1, s2)
        zr, xr, margin = _triangulate_matches_np(cam1.astype(np.float32).astype(np.float64), cam2.astype(np.float32).astype(np.float64),
                                                 p1f.astype(np.float64), p2f.astype(np.float64), R32.astype(np.float64), t32.astype(np.float64), s1, s2)
        if margin < 1e-3:
            continue
        decided += 1
        assert (z > 0) == (zr > 0), (it, z, zr, margin)
        if z > 0:
            accepted += 1
            assert np.allclose(x, xr, rtol=2e-3, atol=2e-3), (x, xr)
    assert decided > 300 and 50 < accepted < decided - 50, (decided, accepted)


def test_camera_project_unproject_roundtrip_oracle():
    """GeometricCamera::unproject then project returns the pixel (both models), and the float KB8 projection agrees with a double model."""
    import oracle_match_bind as om
    rng = np.random.default_rng(99)
    kb = np.array([190.9, 190.8, 254.9, 256.8, 0.0034, 0.0007, -0.0020, 0.0002], np.float32)
    pin = np.array([458.0, 457.0, 367.0, 248.0, 0, 0, 0, 0], np.float32)
    for _ in range(200):
        u, v = rng.uniform(80, 430), rng.uniform(80, 430)          # inside 85 degrees of the axis (unproject clamps theta at 90, as the reference does)
        for t, cam in ((1, kb), (0, pin)):
            ray = om.camera_unproject_f(t, cam, u, v)
            uv = om.camera_project_f(t, cam, ray * np.float32(rng.uniform(0.5, 8)))
            assert abs(uv[0] - u) < 2e-2 and abs(uv[1] - v) < 2e-2, (t, u, v, uv)
        P = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(0.3, 9)], np.float32)
        assert np.allclose(om.camera_project_f(1, kb, P), om.kb8_project_np((1, kb.astype(np.float64)), P.astype(np.float64)), atol=2e-3)


def test_search_for_triangulation_general_oracle_reduces_to_the_pinhole_form():
    """The general restatement on single Pinhole cameras == the fast-path restatement (same inputs, F12 of combination 0), and on rigs every
    match joins keypoints whose cameras were handed the right relative pose (a match's pair passes TriangulateMatches when re-evaluated)."""
    import oracle_match_bind as om
    rng = np.random.default_rng(5)
    for n1, n2 in ((0, 10), (300, 0), (500, 650)):
        c = om.make_tri_general_case(rng, n1, n2, "pinhole")
        n, m = om.search_for_triangulation_general(c, True)
        c2 = dict(c); c2["F12"] = c["geom"]["F12"][0]; c2["ep"] = (c["geom"]["ep_x"], c["geom"]["ep_y"]); c2["only_stereo"] = False; c2["coarse"] = False
        n2_, m2 = om.search_for_triangulation(c2, True, False)
        assert n == n2_ and np.array_equal(m, m2)
    c = om.make_tri_general_case(rng, 700, 800, "rig")
    n, m = om.search_for_triangulation_general(c, False)
    g = c["geom"]
    assert n > 100
    for i1 in np.flatnonzero(m >= 0):
        i2 = m[i1]
        b1, b2 = int(i1 >= g["nleft1"]), int(i2 >= g["nleft2"])
        z, _ = om.kb8_triangulate_matches(1, g["cam1"][b1], 1, g["cam2"][b2], (c["kp1"]["x"][i1], c["kp1"]["y"][i1]), (c["kp2"]["x"][i2], c["kp2"]["y"][i2]),
                                          g["R12"][2 * b1 + b2], g["t12"][2 * b1 + b2], c["sigma2_1"][c["kp1"]["octave"][i1]], c["sigma2"][c["kp2"]["octave"][i2]])
        assert z > 0.0001
    # rig keyframes have no stereo keypoints: bOnlyStereo finds nothing (ORBmatcher.cc:1044-1048)
    c = om.make_tri_general_case(rng, 300, 300, "rig", only_stereo=True)
    assert om.search_for_triangulation_general(c, True)[0] == 0


def test_search_for_triangulation_points_oracle_against_the_scene_and_the_first_overload():
    """The overload that returns the points (ORBmatcher.cc:1212-1402), restated: (1) with Tcw1 = [I | 0] and Tcw2 = the inverse of the
    first overload's (R12, t12), KannalaBrandt8::matchAndtriangulate is TriangulateMatches without the z1 > 1e-4 test -- same accept /
    reject and the same x3D bit for bit; (2) on a scene with absolute poses the kept points are the scene's world points (a few cm off
    at most: the keypoints carry sub-pixel jitter); (3) a Pinhole first camera matches nothing; bOnlyStereo / the epipole are not read."""
    import oracle_match_bind as om
    rng = np.random.default_rng(77)
    c = om.make_tri_general_case(rng, 900, 1000, "kb8")
    g = c["geom"]
    R12 = g["R12"][0].reshape(3, 3); t12 = g["t12"][0]
    R21 = np.ascontiguousarray(R12.T)
    t21 = np.array([np.float32(-1.0 * (float(R21[i, 0]) * float(t12[0]) + float(R21[i, 1]) * float(t12[1]) + float(R21[i, 2]) * float(t12[2]))) for i in range(3)], np.float32)
    T1 = np.concatenate([np.eye(3, dtype=np.float32), np.zeros((3, 1), np.float32)], 1).reshape(12)
    T2 = np.concatenate([R21, t21[:, None]], 1).astype(np.float32).reshape(12)
    cam = np.ascontiguousarray(g["cam1"][0]); x = np.zeros(3, np.float32)
    acc = 0
    for i2 in range(0, 1000):
        i1 = int(rng.integers(0, 900))
        p1 = (float(c["kp1"]["x"][i1]), float(c["kp1"]["y"][i1])); p2 = (float(c["kp2"]["x"][i2]), float(c["kp2"]["y"][i2]))
        s1 = float(c["sigma2_1"][c["kp1"]["octave"][i1]]); s2 = float(c["sigma2"][c["kp2"]["octave"][i2]])
        ok = om.lib.orc_kb8_match_and_triangulate(cam.ctypes.data, 1, cam.ctypes.data, p1[0], p1[1], p2[0], p2[1], T1.ctypes.data, T2.ctypes.data, s1, s2, x.ctypes.data)
        z, x_ref = om.kb8_triangulate_matches(1, cam, 1, cam, p1, p2, R12, t12, s1, s2)
        assert bool(ok) == (z > 0), (i1, i2, ok, z)
        if ok:
            assert np.array_equal(x.view(np.uint32), x_ref.view(np.uint32)); acc += 1
    # matched reprojections among random pairs are rare: walk true pairs as well
    n, m = om.search_for_triangulation_general(c, False)
    for i1 in np.flatnonzero(m >= 0)[:200]:
        i2 = int(m[i1])
        p1 = (float(c["kp1"]["x"][i1]), float(c["kp1"]["y"][i1])); p2 = (float(c["kp2"]["x"][i2]), float(c["kp2"]["y"][i2]))
        s1 = float(c["sigma2_1"][c["kp1"]["octave"][i1]]); s2 = float(c["sigma2"][c["kp2"]["octave"][i2]])
        ok = om.lib.orc_kb8_match_and_triangulate(cam.ctypes.data, 1, cam.ctypes.data, p1[0], p1[1], p2[0], p2[1], T1.ctypes.data, T2.ctypes.data, s1, s2, x.ctypes.data)
        z, x_ref = om.kb8_triangulate_matches(1, cam, 1, cam, p1, p2, R12, t12, s1, s2)
        assert ok and z > 0 and np.array_equal(x.view(np.uint32), x_ref.view(np.uint32)); acc += 1
    assert acc > 100
    # (2) absolute poses: the kept points re-project onto the KF1 keypoint within the gate and lie in front of both cameras
    for mode in ("kb8", "rig"):
        c = om.make_tri_general_case(rng, 900, 1000, mode, only_stereo=True)
        P = om.tri_case_poses(c, 3)
        for ori in (True, False):
            n, m, pts = om.search_for_triangulation_points(c, P, ori)
            assert n == (m >= 0).sum() and n > 150
            gm = c["geom"]
            for i1 in np.flatnonzero(m >= 0):
                b1 = int(gm["nleft1"] != -1 and i1 >= gm["nleft1"]); b2 = int(gm["nleft2"] != -1 and m[i1] >= gm["nleft2"])
                for T, camp, kp, idx, sg in ((P["Tcw1"][b1], gm["cam1"][b1], c["kp1"], i1, c["sigma2_1"]), (P["Tcw2"][b2], gm["cam2"][b2], c["kp2"], m[i1], c["sigma2"])):
                    T = T.reshape(3, 4).astype(np.float64)
                    Xc = T[:, :3] @ pts[i1].astype(np.float64) + T[:, 3]
                    assert Xc[2] > 0
                    uv = om.kb8_project_np((1, camp.astype(np.float64)), Xc)
                    assert ((uv - (kp["x"][idx], kp["y"][idx])) ** 2).sum() <= 5.991 * sg[kp["octave"][idx]] * 1.001 + 1e-3
            assert (pts[m < 0] == 0).all() or ori            # only rotation-histogram rejects keep a point (vMatchesPoints12 is not cleared)
        # the same walk with only_stereo cleared and the epipole moved: nothing changes
        c2 = dict(c); g2 = c["geom"].copy(); g2["only_stereo"] = 0; g2["ep_x"] += 300; c2["geom"] = g2
        assert np.array_equal(om.search_for_triangulation_points(c2, P, True)[1], om.search_for_triangulation_points(c, P, True)[1])
    # (3)
    c = om.make_tri_general_case(rng, 400, 400, "pinhole")
    n, m, pts = om.search_for_triangulation_points(c, om.tri_case_poses(c, 4), True)
    assert n == 0 and (m == -1).all() and (pts == 0).all()


def _dense_numpy_ba_model(g, p, schedule):
    """Optimizer::LocalBundleAdjustment's optimisation (Optimizer.cc:2046-2122: optimize(5), then optimize(10) on the unchanged graph) on
    the vendored Levenberg-Marquardt (optimization_algorithm_levenberg.cpp:61-194), sharing no code with the oracle: numpy only, NUMERIC
    Jacobians of the mono residual (central differences of the left-multiplied SE3 perturbation [omega, upsilon], g2o's SE3Quat::exp
    order, and of the point), the DENSE (6 nf + 3 L) system solved by numpy.linalg.solve (no Schur complement), Huber weights with the
    reference's float constants, lambda = 1e-50 max diag per optimize() call.  Returns (R, t, X, chi2 first / last, trials, per-edge
    chi2 of the last evaluated estimate, per-edge depth at the final one)."""
    def quat_to_R(q):
        x, y, z, w = q / np.linalg.norm(q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def skew(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])

    def se3_exp(d):                                   # d = [omega, upsilon] -> (R, t), closed form with series for small angles
        om, up = d[:3], d[3:]
        th = np.linalg.norm(om)
        Om = skew(om)
        if th < 1e-8:
            return np.eye(3) + Om + 0.5 * Om @ Om, (np.eye(3) + 0.5 * Om + Om @ Om / 6) @ up
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * Om @ Om
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * Om @ Om
        return R, V @ up

    fx, fy, cx, cy = g["fx"], g["fy"], g["cx"], g["cy"]
    R = [quat_to_R(q[:4]) for q in g["poses0"]]; t = [q[4:].copy() for q in g["poses0"]]
    X = g["points0"].astype(np.float64).copy()
    free = np.flatnonzero(np.asarray(g["pose_fixed"]) == 0)
    hidx = {int(k): i for i, k in enumerate(free)}
    nf, L, E = len(free), g["n_points"], g["n_edges"]
    assert np.all(np.asarray(g["edge_stereo"]) == 0)
    # thHuber = (float)sqrt(5.991) and delta^2 kept as a float (Optimizer.cc:1910-1911, robust_kernel_impl.h:84)
    delta = float(np.float32(np.sqrt(p.huber_mono2))); delta2 = float(np.float32(delta * delta))
    ek = [int(v) for v in g["edge_pose"]]; el = [int(v) for v in g["edge_point"]]
    ob2 = g["edge_obs"][:, :2].astype(np.float64); is2 = g["edge_inv_sigma2"].astype(np.float64)

    def residual(R, t, X, e):
        Pc = R[ek[e]] @ X[el[e]] + t[ek[e]]
        return ob2[e] - np.array([fx * Pc[0] / Pc[2] + cx, fy * Pc[1] / Pc[2] + cy])

    def chis(R, t, X):
        return np.array([float(r @ r) for r in (residual(R, t, X, e) for e in range(E))]) * is2

    def robust(c):
        return float(np.sum(np.where(c <= delta2, c, 2 * np.sqrt(c) * delta - delta2)))

    n = 6 * nf + 3 * L
    h = 1e-6
    trials = 0
    chi_first = chi_last = None
    c_eval = None
    for iters in schedule:
        lam = ni = 0.0; nbad = 0
        for it in range(iters):
            c_eval = chis(R, t, X)
            cur = robust(c_eval); ini = cur
            if chi_first is None:
                chi_first = cur
            H = np.zeros((n, n)); b = np.zeros(n)
            for e in range(E):
                k, l = ek[e], el[e]
                r = residual(R, t, X, e)
                w = is2[e] * (1.0 if c_eval[e] <= delta2 else delta / np.sqrt(c_eval[e]))
                cols, J = [], []
                if k in hidx:
                    Jp = np.zeros((2, 6))
                    for a in range(6):
                        d = np.zeros(6); d[a] = h
                        Rp, tp = se3_exp(d); Rm, tm = se3_exp(-d)
                        Ra = list(R); ta = list(t); Ra[k] = Rp @ R[k]; ta[k] = Rp @ t[k] + tp
                        Rb = list(R); tb = list(t); Rb[k] = Rm @ R[k]; tb[k] = Rm @ t[k] + tm
                        Jp[:, a] = (residual(Ra, ta, X, e) - residual(Rb, tb, X, e)) / (2 * h)
                    cols += list(range(6 * hidx[k], 6 * hidx[k] + 6)); J.append(Jp)
                Jx = np.zeros((2, 3))
                for a in range(3):
                    Xa = X.copy(); Xa[l, a] += h; Xb = X.copy(); Xb[l, a] -= h
                    Jx[:, a] = (residual(R, t, Xa, e) - residual(R, t, Xb, e)) / (2 * h)
                cols += list(range(6 * nf + 3 * l, 6 * nf + 3 * l + 3)); J.append(Jx)
                Je = np.hstack(J)
                H[np.ix_(cols, cols)] += w * Je.T @ Je
                b[cols] += -w * Je.T @ r
            if it == 0:
                lam = 1e-50 * np.max(np.abs(np.diag(H))); ni = 2.0; nbad = 0
            q = 0
            while True:
                d = np.linalg.solve(H + lam * np.eye(n), b)
                Rt = list(R); tt = list(t)
                for k, i in hidx.items():
                    Rd, td = se3_exp(d[6 * i:6 * i + 6])
                    Rt[k] = Rd @ R[k]; tt[k] = Rd @ t[k] + td
                Xt = X + d[6 * nf:].reshape(L, 3)
                c_eval = chis(Rt, tt, Xt)
                tmp = robust(c_eval)
                rho = (cur - tmp) / (float(d @ (lam * d + b)) + 1e-3)
                if rho > 0 and np.isfinite(tmp):
                    lam *= max(1 / 3, min(1 - (2 * rho - 1) ** 3, 2 / 3)); ni = 2.0; cur = tmp; R, t, X = Rt, tt, Xt
                else:
                    lam *= ni; ni *= 2
                q += 1; trials += 1
                if not (rho < 0 and q < p.max_trials):
                    break
            chi_last = cur
            if q == p.max_trials or rho == 0:
                break
            nbad = nbad + 1 if (ini - cur) * 1e3 < ini else 0
            if nbad >= 3:
                break
    depth = np.array([(R[ek[e]] @ X[el[e]] + t[ek[e]])[2] for e in range(E)])
    return R, t, X, chi_first, chi_last, trials, c_eval, depth, quat_to_R


def test_ba_first_lm_iteration_against_an_independent_dense_numpy_model():
    """One optimize(1): pins the oracle's normal equations, Schur elimination, LDL^T, update rule and lambda initialisation
    (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:47,171-185) against the dense model above: estimates after the step
    and the robust chi2 agree to finite-difference accuracy."""
    import synth_ba
    import oracle_ba_bind as ob
    g = synth_ba.make_graph(n_kf=6, n_pts=40, obs=4, seed=321, n_fixed=2, outlier_frac=0.1)
    p = ob.default_params()
    p.iters1, p.iters2 = 1, 0
    rc, o_poses, o_pts, _, o_st, _, _ = ob.solve_with_gate_values(g, p)
    assert o_st["iterations_run"][0] == 1 and o_st["lm_trials"] == 1
    R1, t1, X1, chi0, chi1, trials, _, _, quat_to_R = _dense_numpy_ba_model(g, p, [1])
    assert trials == 1
    assert abs(chi0 - o_st["chi2_initial"]) <= 1e-9 * chi0
    assert chi1 < chi0                                                    # the step is accepted on both sides (one trial)
    assert abs(chi1 - o_st["chi2_final"]) <= 1e-5 * chi1, (chi1, o_st["chi2_final"])
    for k in range(g["n_poses"]):
        assert np.max(np.abs(quat_to_R(o_poses[k, :4]) - R1[k])) <= 1e-6 and np.max(np.abs(o_poses[k, 4:] - t1[k])) <= 1e-6, k
    assert np.max(np.abs(o_pts - X1)) <= 1e-5


@pytest.mark.parametrize("seed", [322, 323])
def test_ba_whole_schedule_against_an_independent_dense_numpy_model(seed):
    """The whole LocalBundleAdjustment optimisation -- optimize(5) + optimize(10), lambda re-initialised by the second call, the accept
    rule, the lambda schedule and the three-bad-iterations stop of the vendored LM -- on the dense model: same number of LM trials,
    estimates to 1e-5, the same outlier set (chi2 of the last evaluated estimate > 5.991 or depth <= 0, Optimizer.cc:2126-2173) for
    every edge whose chi2 is not within 1e-4 of the gate."""
    import synth_ba
    import oracle_ba_bind as ob
    g = synth_ba.make_graph(n_kf=6, n_pts=40, obs=4, seed=seed, n_fixed=2, outlier_frac=0.1)
    p = ob.default_params()
    rc, o_poses, o_pts, o_out, o_st, o_chi2, o_depth = ob.solve_with_gate_values(g, p)
    R1, t1, X1, chi0, chi1, trials, c_eval, depth, quat_to_R = _dense_numpy_ba_model(g, p, [p.iters1, p.iters2])
    assert trials == o_st["lm_trials"], (trials, o_st)
    assert abs(chi1 - o_st["chi2_final"]) <= 1e-5 * chi1, (chi1, o_st["chi2_final"])
    for k in range(g["n_poses"]):
        assert np.max(np.abs(quat_to_R(o_poses[k, :4]) - R1[k])) <= 1e-5 and np.max(np.abs(o_poses[k, 4:] - t1[k])) <= 1e-5, k
    assert np.max(np.abs(o_pts - X1)) <= 1e-4
    clear = np.abs(c_eval - p.huber_mono2) > 1e-4
    out_m = (c_eval > p.huber_mono2) | ~(depth > 0)
    np.testing.assert_array_equal(o_out.astype(bool)[clear], out_m[clear])
    assert clear.mean() > 0.95 and 3 <= int(out_m.sum()) < g["n_edges"] // 2


@pytest.mark.parametrize("check_ori,window,ratio", [(True, 100, 0.9), (False, 60, 0.8), (True, 300, 0.95)])
def test_search_for_initialization_oracle_against_python(check_ori, window, ratio):
    """ORBmatcher::SearchForInitialization (ORBmatcher.cc:709-824) restated in plain Python on top of the GetFeaturesInArea oracle and
    numpy popcounts: level-0 F1 points in order, vMatchedDistance skips, best / second best with strict <, TH_LOW and the ratio test in
    float, displaced matches, the rotation histogram that keeps displaced entries (round(), bin 30 -> 0), ComputeThreeMaxima
    (ORBmatcher.cc:2307-2348), vbPrevMatched update.  Clustered points with few distinct descriptors make displacement frequent."""
    rng = np.random.default_rng(5 + window)

    def mk(n, ndesc):
        kp = np.zeros(n, ob.KP_DTYPE)
        kp["x"] = np.clip(rng.normal(320, 120, n), -5, 645).astype(np.float32)
        kp["y"] = np.clip(rng.normal(240, 90, n), -5, 485).astype(np.float32)
        kp["angle"] = rng.uniform(0, 360, n).astype(np.float32)
        kp["octave"] = (rng.uniform(0, 1, n) < 0.3).astype(np.int32) * rng.integers(1, 4, n)
        base = rng.integers(0, 256, (ndesc, 32), dtype=np.uint8)
        d = base[rng.integers(0, ndesc, n)].copy()
        d[np.arange(n), rng.integers(0, 32, n)] ^= (1 << rng.integers(0, 8, n)).astype(np.uint8)
        d[np.arange(n), rng.integers(0, 32, n)] ^= (1 << rng.integers(0, 8, n)).astype(np.uint8)
        return kp, d
    b = (0.0, 0.0, 640.0, 480.0)
    for trial in range(3):
        (k1, d1), (k2, d2) = mk(300, 8), mk(320, 8)
        d2[:] = d1[rng.integers(0, len(d1), len(d2))]
        d2[np.arange(len(d2)), rng.integers(0, 32, len(d2))] ^= (1 << rng.integers(0, 8, len(d2))).astype(np.uint8)
        prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32) + rng.normal(0, 10, (len(k1), 2)).astype(np.float32)
        n, m12, pv = om.search_for_initialization(k1, d1, k2, d2, b, prev, window, ratio, check_ori)
        # ---- the Python model
        pop = np.array([bin(i).count("1") for i in range(256)], np.int32)
        INT_MAX = 2 ** 31 - 1
        vm12 = [-1] * len(k1); vm21 = [-1] * len(k2); vmd = [INT_MAX] * len(k2)
        hist = [[] for _ in range(30)]
        nm = 0
        for i1 in range(len(k1)):
            if k1["octave"][i1] > 0:
                continue
            idx = om.features_in_area(k2, b, float(prev[i1, 0]), float(prev[i1, 1]), float(window), 0, 0)
            if len(idx) == 0:
                continue
            best = best2 = INT_MAX; bi = -1
            for i2 in idx.tolist():
                dist = int(pop[d1[i1] ^ d2[i2]].sum())
                if vmd[i2] <= dist:
                    continue
                if dist < best:
                    best2, best, bi = best, dist, i2
                elif dist < best2:
                    best2 = dist
            if best <= 50 and np.float32(best) < np.float32(best2) * np.float32(ratio):
                if vm21[bi] >= 0:
                    vm12[vm21[bi]] = -1; nm -= 1
                vm12[i1] = bi; vm21[bi] = i1; vmd[bi] = best; nm += 1
                if check_ori:
                    rot = np.float32(k1["angle"][i1]) - np.float32(k2["angle"][bi])
                    if rot < 0:
                        rot = np.float32(rot + np.float32(360.0))
                    v = float(np.float32(rot * np.float32(1.0 / 30)))
                    bn = int(np.floor(abs(v) + 0.5) * (1 if v >= 0 else -1))       # round(): half away from zero
                    hist[0 if bn == 30 else bn].append(i1)
        if check_ori:
            m1 = m2 = m3 = 0; i1_ = i2_ = i3_ = -1
            for i in range(30):
                s = len(hist[i])
                if s > m1: m3, m2, m1, i3_, i2_, i1_ = m2, m1, s, i2_, i1_, i
                elif s > m2: m3, m2, i3_, i2_ = m2, s, i2_, i
                elif s > m3: m3, i3_ = s, i
            if m2 < np.float32(0.1) * np.float32(m1): i2_ = i3_ = -1
            elif m3 < np.float32(0.1) * np.float32(m1): i3_ = -1
            for i in range(30):
                if i in (i1_, i2_, i3_):
                    continue
                for j in hist[i]:
                    if vm12[j] >= 0:
                        vm12[j] = -1; nm -= 1
        pv_m = prev.copy()
        for i1 in range(len(k1)):
            if vm12[i1] >= 0:
                pv_m[i1] = (k2["x"][vm12[i1]], k2["y"][vm12[i1]])
        assert n == nm, (trial, n, nm)
        np.testing.assert_array_equal(m12, np.array(vm12, np.int32))
        assert pv.tobytes() == pv_m.tobytes()
        assert nm > 20


@pytest.mark.parametrize("stereo_frac,seed", [(0.0, 3), (0.5, 4), (1.0, 5)])
def test_pose_optimization_oracle_against_an_independent_numpy_model(stereo_frac, seed):
    """Optimizer::PoseOptimization (Optimizer.cc:854-1168) and the vendored Levenberg-Marquardt (optimization_algorithm_levenberg.cpp:61-194)
    restated in numpy without any oracle code: NUMERIC Jacobians of the two unary edges (the stereo one with its float 1 / z,
    types_six_dof_expmap.cpp:339-346), dense 6 x 6 solve by numpy, lambda = 1e-50 max diag per optimize() call, the rho / scale accept
    rule with lambda *= max(1/3, min(1 - (2 rho - 1)^3, 2/3)) or *= ni, ni *= 2, Raul's three-bad-iterations stop, four rounds from the
    SAME initial pose, inliers classified on the errors of the last EVALUATED estimate (a rejected trial's when the round ends on one),
    outliers re-evaluated at the final one, float chi2 against float gates, no robust kernel after the third round.  Same inlier set,
    pose to 1e-6."""
    import synth_ba
    prob = synth_ba.make_pose_problem(seed, n=160, stereo_frac=stereo_frac, outlier_frac=0.12)
    Xw, obs, is2, cam, pose0 = prob["Xw"], prob["obs"], prob["inv_sigma2"], prob["cam"], prob["pose0"]
    n_in, o_pose, o_out, o_st = obb.pose_optimization(Xw, obs, is2, cam, pose0)
    fx, fy, cx, cy, bf = [float(c) for c in cam]
    f32 = np.float32

    def quat_to_R(q):
        x, y, z, w = q / np.linalg.norm(q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def skew(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])

    def se3_exp(d):
        om, up = d[:3], d[3:]
        th = np.linalg.norm(om); Om = skew(om)
        if th < 1e-8:
            return np.eye(3) + Om + 0.5 * Om @ Om, (np.eye(3) + 0.5 * Om + Om @ Om / 6) @ up
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * Om @ Om
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * Om @ Om
        return R, V @ up

    stereo = obs[:, 2] >= 0
    n = len(Xw)

    def err(R, t, e, smooth=False):                            # smooth: for the difference quotients (the float rounding of 1 / z is not differentiable)
        Xc = R @ Xw[e] + t
        if not stereo[e]:
            return np.array([obs[e, 0] - (fx * Xc[0] / Xc[2] + cx), obs[e, 1] - (fy * Xc[1] / Xc[2] + cy), 0.0])
        invz = 1.0 / Xc[2] if smooth else float(f32(1.0 / Xc[2]))
        u = Xc[0] * invz * fx + cx
        return np.array([obs[e, 0] - u, obs[e, 1] - (Xc[1] * invz * fy + cy), obs[e, 2] - (u - bf * invz)])

    dm, ds = float(f32(np.sqrt(5.991))), float(f32(np.sqrt(7.815)))
    dm2, ds2 = float(f32(dm * dm)), float(f32(ds * ds))

    def rho(c, e, robust):
        if not robust:
            return c, 1.0
        d, d2 = (ds, ds2) if stereo[e] else (dm, dm2)
        return (c, 1.0) if c <= d2 else (2 * np.sqrt(c) * d - d2, d / np.sqrt(c))

    R_init, t_init = quat_to_R(np.asarray(pose0[:4], np.float64)), np.asarray(pose0[4:], np.float64).copy()
    level = np.zeros(n, bool)                                   # True: outlier, not optimised
    R, t = R_init, t_init
    robust = True
    for it in range(4):
        R, t = R_init, t_init
        act = np.flatnonzero(~level)
        Re, te = R, t                                           # estimate the stored errors belong to
        lam = ni = 0.0; nbad = 0
        for k in range(10):
            Re, te = R, t
            E = {e: err(R, t, e) for e in act}
            cur = sum(rho(float(E[e] @ E[e]) * is2[e], e, robust)[0] for e in act)
            ini = cur
            H = np.zeros((6, 6)); b = np.zeros(6)
            for e in act:
                J = np.zeros((3, 6)); h = 1e-6
                for a in range(6):
                    d = np.zeros(6); d[a] = h
                    Rp, tp = se3_exp(d); Rm, tm = se3_exp(-d)
                    J[:, a] = (err(Rp @ R, Rp @ t + tp, e, True) - err(Rm @ R, Rm @ t + tm, e, True)) / (2 * h)
                w = rho(float(E[e] @ E[e]) * is2[e], e, robust)[1] * is2[e]
                H += w * J.T @ J; b += -w * J.T @ E[e]
            if k == 0:
                lam = 1e-50 * np.max(np.abs(np.diag(H))); ni = 2.0; nbad = 0
            q = 0
            while True:
                x = np.linalg.solve(H + lam * np.eye(6), b)
                Rd, td = se3_exp(x)
                Rt, tt = Rd @ R, Rd @ t + td
                Re, te = Rt, tt
                tmp = sum(rho(float(v @ v) * is2[e], e, robust)[0] for e, v in ((e, err(Rt, tt, e)) for e in act))
                r_ = (cur - tmp) / (float(x @ (lam * x + b)) + 1e-3)
                if r_ > 0 and np.isfinite(tmp):
                    lam *= max(1 / 3, min(1 - (2 * r_ - 1) ** 3, 2 / 3)); ni = 2.0; cur = tmp; R, t = Rt, tt
                else:
                    lam *= ni; ni *= 2
                q += 1
                if not (r_ < 0 and q < 100):
                    break
            if q == 100 or r_ == 0:
                break
            nbad = nbad + 1 if (ini - cur) * 1e3 < ini else 0
            if nbad >= 3:
                break
        nb = 0
        for e in range(n):
            v = err(R, t, e) if level[e] else err(Re, te, e)
            c = f32(float(v @ v) * is2[e])
            if c > (f32(7.815) if stereo[e] else f32(5.991)):
                level[e] = True; nb += 1
            else:
                level[e] = False
        if it == 2:
            robust = False
        if n < 10:
            break
    assert n_in == n - nb
    np.testing.assert_array_equal(o_out.astype(bool), level)
    assert np.max(np.abs(quat_to_R(o_pose[:4]) - R)) <= 1e-6 and np.max(np.abs(o_pose[4:] - t)) <= 1e-6


def test_ba_oracle_per_keyframe_calibration():
    """The oracle with a camera table (round 4): (1) a table whose entries all equal the graph's single calibration reproduces the
    single-calibration solve bit for bit; (2) a noise-free window whose keyframes alternate between two Pinhole calibrations is
    recovered exactly -- and is NOT when every keyframe is given camera 0 (the table is really what the edges project through)."""
    import oracle_ba_bind as ob
    import synth_ba
    g = synth_ba.make_graph(n_kf=8, n_pts=120, obs=5, seed=3, stereo_frac=0.3)
    rc, p0, x0, o0, s0 = ob.solve(g)
    gt = dict(g)
    gt["cameras"] = [dict(fx=g["fx"], fy=g["fy"], cx=g["cx"], cy=g["cy"], bf=g["bf"])] * 3
    gt["pose_camera"] = np.array([i % 3 for i in range(8)], np.int32)
    rc, p1, x1, o1, s1 = ob.solve(gt)
    assert np.array_equal(p0, p1) and np.array_equal(x0, x1) and np.array_equal(o0, o1) and s0["lm_trials"] == s1["lm_trials"]
    cams = [dict(fx=458.0, fy=458.0, cx=320.0, cy=240.0, bf=50.0, stereo_frac=0.3), dict(fx=380.0, fy=395.0, cx=300.0, cy=255.0, bf=27.0, stereo_frac=0.3)]
    g2 = synth_ba.make_graph(n_kf=10, n_pts=200, obs=6, seed=4, cameras=cams, pose_camera=[i % 2 for i in range(10)], pixel_noise=0.0, outlier_frac=0.0)
    rc, p2, x2, o2, s2 = ob.solve(g2)
    assert s2["chi2_final"] < 1e-3 and not o2.any()
    assert np.sqrt(np.mean((x2 - g2["points_gt"]) ** 2)) < 2e-3
    wrong = dict(g2); wrong.pop("cameras"); wrong.pop("pose_camera")
    rc, p3, x3, o3, s3 = ob.solve(wrong)
    assert s3["chi2_final"] > 1e3 * max(s2["chi2_final"], 1e-6)
