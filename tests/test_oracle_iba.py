"""CPU tests of the inertial local-BA oracle (oracle/iba_oracle.c): the restated Jacobians (G2oTypes.cc:349-482, :742-800) against
finite differences taken through the restated vertex updates (G2oTypes.cc:192-220), the SO3 helpers, and the LM loop on synthetic
visual-inertial windows.  PARITY UNPINNED (no reference fixtures exist for this path; see oracle/iba_oracle.h)."""
import numpy as np
import pytest
import oracle_iba_bind as ib


def test_so3_helpers_roundtrip():
    rng = np.random.default_rng(0)
    for s in (1e-7, 1e-3, 0.3, 2.0):
        w = rng.normal(0, 1, 3)
        w = w / np.linalg.norm(w) * s
        R = ib.exp_so3(w)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14)
        assert np.allclose(ib.log_so3(R), w, atol=1e-9 * max(1, s / 1e-3))


def _fd_inertial(s1, s2, pre, h=1e-6):
    J = np.zeros((9, 24))
    for c in range(24):
        d = np.zeros(24)
        d[c] = h
        ep, _ = ib.edge_inertial(ib.kf_update(s1, d[0:15]), ib.kf_update(s2, np.concatenate([d[15:24], np.zeros(6)])), pre, jac=False)
        d[c] = -h
        em, _ = ib.edge_inertial(ib.kf_update(s1, d[0:15]), ib.kf_update(s2, np.concatenate([d[15:24], np.zeros(6)])), pre, jac=False)
        J[:, c] = (ep - em) / (2 * h)
    return J


def test_inertial_edge_jacobian_matches_finite_differences():
    win = ib.make_window(3, n_opt=4, n_fixed_vis=2, n_points=30)
    a = win.arrays
    for m in range(win.n_inertial):
        s1, s2 = win.kf0[a["in_kf1"][m]], win.kf0[a["in_kf2"][m]]
        err, J = ib.edge_inertial(s1, s2, a["in_preint"][m])
        Jn = _fd_inertial(s1, s2, a["in_preint"][m])
        # The reference's analytic Jacobian drops second-order terms in the rotation residual (it is exact at er = 0); the
        # synthetic states are close to consistent, so the two agree to ~1e-2 relative on the rotation rows and tightly elsewhere
        assert np.allclose(J[3:], Jn[3:], atol=2e-5), np.abs(J[3:] - Jn[3:]).max()
        assert np.allclose(J[:3], Jn[:3], atol=2e-2), np.abs(J[:3] - Jn[:3]).max()
        assert np.linalg.norm(err) < 1.0


def test_inertial_edge_rotation_rows_exact_at_small_residual():
    rng = np.random.default_rng(5)
    win = ib.make_window(4, n_opt=3, n_fixed_vis=1, n_points=20, state_noise=0.0)
    a = win.arrays
    s1, s2 = win.d["kf_true"][a["in_kf1"][0]], win.d["kf_true"][a["in_kf2"][0]]
    err, J = ib.edge_inertial(s1, s2, a["in_preint"][0])
    assert np.linalg.norm(err[:3]) < 5e-3
    Jn = _fd_inertial(s1, s2, a["in_preint"][0])
    assert np.allclose(J, Jn, atol=5e-4), np.abs(J - Jn).max()
    del rng


def test_visual_edge_jacobians_match_finite_differences():
    win = ib.make_window(6, n_opt=3, n_fixed_vis=1, n_points=40)
    a = win.arrays
    h = 1e-6
    for e in range(0, win.n_edges, 7):
        s, X, obs, st = win.kf0[a["edge_kf"][e]], win.pts0[a["edge_point"][e]], a["edge_obs"][e], int(a["edge_stereo"][e])
        err, Jx, Jp = ib.edge_visual(win, s, X, obs, st)
        D = 3 if st else 2
        for c in range(3):
            d = np.zeros(3); d[c] = h
            ep = ib.edge_visual(win, s, X + d, obs, st)[0]; em = ib.edge_visual(win, s, X - d, obs, st)[0]
            assert np.allclose(((ep - em) / (2 * h))[:D], Jx[:D, c], rtol=1e-5, atol=1e-4)
        for c in range(6):
            d = np.zeros(15); d[c] = h
            ep = ib.edge_visual(win, ib.kf_update(s, d), X, obs, st)[0]; em = ib.edge_visual(win, ib.kf_update(s, -d), X, obs, st)[0]
            assert np.allclose(((ep - em) / (2 * h))[:D], Jp[:D, c], rtol=1e-5, atol=1e-4)


def test_solve_reduces_error_and_recovers_states():
    win = ib.make_window(11, n_opt=8, n_fixed_vis=10, n_points=300)
    kf, pts, out, st = ib.solve(win)
    assert st.failed == 0 and st.iterations_run >= 3
    assert st.err_end < 0.5 * st.err
    n_opt = 8
    e0 = np.linalg.norm(win.kf0[:n_opt, 9:12] - win.d["kf_true"][:n_opt, 9:12], axis=1).mean()
    e1 = np.linalg.norm(kf[:n_opt, 9:12] - win.d["kf_true"][:n_opt, 9:12], axis=1).mean()
    assert e1 < 0.5 * e0, (e0, e1)
    v0 = np.linalg.norm(win.kf0[:n_opt, 12:15] - win.d["kf_true"][:n_opt, 12:15], axis=1).mean()
    v1 = np.linalg.norm(kf[:n_opt, 12:15] - win.d["kf_true"][:n_opt, 12:15], axis=1).mean()
    assert v1 < 0.7 * v0, (v0, v1)
    # fixed keyframes untouched, rotations stay orthonormal
    assert np.array_equal(kf[n_opt:], win.kf0[n_opt:])
    for k in range(n_opt):
        R = kf[k, :9].reshape(3, 3)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
    # most injected outliers are flagged, few inliers are
    assert 0 < out.sum() < 0.15 * win.n_edges


def test_solve_large_variant_and_is_deterministic():
    win = ib.make_window(12, n_opt=14, n_fixed_vis=20, n_points=350, large=True)
    a = ib.solve(win, ib.default_params(large=True))
    b = ib.solve(win, ib.default_params(large=True))
    assert a[3].iterations_run <= 4 and a[3].failed == 0
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_fail_check_leaves_inputs_untouched():
    win = ib.make_window(13, n_opt=4, n_fixed_vis=3, n_points=60)
    # an inconsistent preintegration makes the first steps worse than the start: the reference's 2*err < err_end test (Optimizer.cc:5096)
    win.arrays["in_preint"][:, 13:16] += 40.0
    p = ib.default_params()
    p.iterations = 1
    p.max_trials = 1
    kf, pts, out, st = ib.solve(win, p)
    if st.failed:
        assert np.array_equal(kf, win.kf0) and np.array_equal(pts, win.pts0)
    else:
        assert st.err_end <= 2 * st.err * (1 + 1e-6)


def test_fisheye_rig_edges_and_solve():
    """KannalaBrandt8 cameras + EdgeMono(1) right-camera edges (Optimizer.cc:5000-5031, G2oTypes.cc:57-67): Jacobians against
    finite differences (step 1e-3: KannalaBrandt8::project rounds theta and psi to float, the error is piecewise constant at the
    1e-5 px level), then the whole optimisation on a rig window where keyframes hold left and right edges to the same point."""
    win = ib.make_window(71, n_opt=5, n_fixed_vis=4, n_points=150, fisheye_rig=True)
    a = win.arrays
    assert set(np.unique(a["edge_stereo"])) == {0, 2}
    h = 1e-3
    for typ in (0, 2):
        es = [e for e in range(win.n_edges) if a["edge_stereo"][e] == typ][:12]
        for e in es:
            s, X, obs = win.kf0[a["edge_kf"][e]], win.pts0[a["edge_point"][e]], a["edge_obs"][e]
            err, Jx, Jp = ib.edge_visual(win, s, X, obs, typ)
            for c in range(3):
                d = np.zeros(3); d[c] = h
                fd = (ib.edge_visual(win, s, X + d, obs, typ)[0] - ib.edge_visual(win, s, X - d, obs, typ)[0]) / (2 * h)
                assert np.allclose(fd[:2], Jx[:2, c], rtol=2e-3, atol=2e-2)
            for c in range(6):
                d = np.zeros(15); d[c] = h
                fd = (ib.edge_visual(win, ib.kf_update(s, d), X, obs, typ)[0] - ib.edge_visual(win, ib.kf_update(s, -d), X, obs, typ)[0]) / (2 * h)
                assert np.allclose(fd[:2], Jp[:2, c], rtol=2e-3, atol=5e-2)
    twins = sum(1 for e in range(1, win.n_edges) if a["edge_point"][e] == a["edge_point"][e - 1] and a["edge_kf"][e] == a["edge_kf"][e - 1])
    assert twins > 50                                   # both cameras of one keyframe see the point
    kf, pts, out, st = ib.solve(win)
    assert st.failed == 0 and st.err_end < 0.5 * st.err
    n = 5
    e0 = np.linalg.norm(win.kf0[:n, 9:12] - win.d["kf_true"][:n, 9:12], axis=1).mean()
    e1 = np.linalg.norm(kf[:n, 9:12] - win.d["kf_true"][:n, 9:12], axis=1).mean()
    assert e1 < 0.5 * e0


def test_iba_golden_regression():
    """Committed fixture (tests/golden/iba_golden.npz, made by tools/gen_golden.py): guards the inertial oracle itself against drift."""
    from synth_iba import load_golden_windows
    for win, exp in load_golden_windows():
        kf, pts, out, st = ib.solve(win)
        assert [st.iterations_run, st.lm_trials, st.n_outliers, st.failed] == exp["stats"].tolist()
        assert np.abs(kf - exp["kf"]).max() <= 1e-9 and np.abs(pts - exp["pts"]).max() <= 1e-9
        assert np.array_equal(out, exp["outlier"])
        assert abs(st.err - exp["err"][0]) <= 1e-9 * abs(exp["err"][0]) and abs(st.err_end - exp["err"][1]) <= 1e-6 * max(1.0, abs(exp["err"][1]))


@pytest.mark.parametrize("kw", [dict(n_opt=6, n_fixed_vis=5, n_points=120, stereo_frac=0.5),
                                dict(n_opt=4, n_fixed_vis=3, n_points=60, stereo_frac=0.0),
                                dict(n_opt=5, n_fixed_vis=4, n_points=90, fisheye_rig=True)])
def test_initial_robust_chi2_against_an_independent_numpy_restatement(kw):
    """optimizer.activeRobustChi2() before optimize() (Optimizer.cc:5045) from a numpy model written from the reference text alone --
    EdgeInertial::computeError (G2oTypes.cc:717-740) with the bias-corrected deltas of ImuTypes.cc:357-378 (ExpSO3 :48-60, the
    nearest rotation by numpy's SVD for NormalizeRotation :30-36), EdgeGyroRW / EdgeAccRW (G2oTypes.h:632-700), EdgeMono / EdgeStereo
    on ImuCamPose::Project / ProjectStereo (G2oTypes.cc:170-185; camera = Tcb . Twb^-1, second camera through Trl :57-67),
    Pinhole / KannalaBrandt8 project, the Huber kernels with their float dsqr (robust_kernel_impl.cpp:65-91; deltas
    Optimizer.cc:4834,4892-4894) -- must equal the oracle's `err`: pins every residual and information weight of the restated
    graph (the deltas in double here as in the oracle: see oracle/iba_oracle.h for that documented deviation)."""
    win = ib.make_window(77, **kw)
    _, _, _, st = ib.solve(win)
    a, d = win.arrays, win.d
    kf, pts = win.kf0.reshape(-1, 21), win.pts0.reshape(-1, 3)
    Rcb, tcb = win.Rcb.reshape(3, 3), win.tcb
    fx, fy, cx, cy, bf = [float(x) for x in win.cam]

    def huber(e, delta):
        dsqr = float(np.float32(delta * delta))
        return e if e <= dsqr else 2 * np.sqrt(e) * delta - dsqr

    def project(Xc, cam, model, kb):
        f0, f1, c0, c1 = cam
        if model == 0:
            return np.array([f0 * Xc[0] / Xc[2] + c0, f1 * Xc[1] / Xc[2] + c1])
        x2y2 = Xc[0] ** 2 + Xc[1] ** 2                                       # KannalaBrandt8.cpp:52-69: atan2f / sqrtf, i.e. float angles
        th = float(np.float32(np.arctan2(float(np.sqrt(np.float32(x2y2))), float(np.float32(Xc[2])))))
        psi = float(np.float32(np.arctan2(float(np.float32(Xc[1])), float(np.float32(Xc[0])))))
        r = th + kb[0] * th ** 3 + kb[1] * th ** 5 + kb[2] * th ** 7 + kb[3] * th ** 9
        return np.array([f0 * r * np.cos(psi) + c0, f1 * r * np.sin(psi) + c1])

    def exp_so3(v):                                                          # ImuTypes.cc:48-60 (eps 1e-4)
        th2 = float(v @ v); th = np.sqrt(th2)
        W = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        return np.eye(3) + W + 0.5 * W @ W if th < 1e-4 else np.eye(3) + W * np.sin(th) / th + W @ W * (1 - np.cos(th)) / th2

    def log_so3(R):                                                          # G2oTypes.cc:1010-1025
        tr = np.trace(R)
        w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
        c = (tr - 1) * 0.5
        if c > 1 or c < -1:
            return w
        th = np.arccos(c); s = np.sin(th)
        return w if abs(s) < 1e-5 else th * w / s

    th_mono, th_stereo = float(np.float32(np.sqrt(5.991))), float(np.float32(np.sqrt(7.815)))
    total = 0.0
    cam2 = model2 = kb2 = Trl = None
    if "Trl" in d:
        Trl = np.asarray(d["Trl"], np.float64).reshape(3, 4); cam2 = [float(x) for x in d["cam2"]]
        model2 = int(d.get("camera2_model", 0)); kb2 = d.get("kb2", (0, 0, 0, 0))
    model, kb = int(d.get("camera_model", 0)), d.get("kb", (0, 0, 0, 0))
    for e in range(win.n_edges):
        s = kf[a["edge_kf"][e]]
        Rwb, twb = s[:9].reshape(3, 3), s[9:12]
        Rcw = Rcb @ Rwb.T; tcw = Rcb @ (-Rwb.T @ twb) + tcb
        X = pts[a["edge_point"][e]]
        typ = int(a["edge_stereo"][e]); obs = a["edge_obs"][e]
        if typ == 2:                                                         # EdgeMono(1): the second camera
            Xc = Trl[:, :3] @ (Rcw @ X + tcw) + Trl[:, 3]
            r = obs[:2] - project(Xc, cam2, model2, kb2)
            total += huber(float(r @ r) * a["edge_inv_sigma2"][e], th_mono)
        elif typ == 0:
            r = obs[:2] - project(Rcw @ X + tcw, (fx, fy, cx, cy), model, kb)
            total += huber(float(r @ r) * a["edge_inv_sigma2"][e], th_mono)
        else:
            Xc = Rcw @ X + tcw
            uv = project(Xc, (fx, fy, cx, cy), model, kb)
            r = obs - np.array([uv[0], uv[1], uv[0] - bf / Xc[2]])
            total += huber(float(r @ r) * a["edge_inv_sigma2"][e], th_stereo)
    g = np.array([0.0, 0.0, -float(np.float32(9.81))])                       # IMU::GRAVITY_VALUE is a float (ImuTypes.h:40)
    for m in range(win.n_inertial):
        s1, s2 = kf[a["in_kf1"][m]], kf[a["in_kf2"][m]]
        p = a["in_preint"][m]
        dt, dR, dV, dP = p[0], p[1:10].reshape(3, 3), p[10:13], p[13:16]
        JRg, JVg, JVa, JPg, JPa = [p[16 + 9 * i:25 + 9 * i].reshape(3, 3) for i in range(5)]
        bg0, ba0 = p[61:64], p[64:67]
        R1, t1, v1, bg1, ba1 = s1[:9].reshape(3, 3), s1[9:12], s1[12:15], s1[15:18], s1[18:21]
        R2, t2, v2, bg2, ba2 = s2[:9].reshape(3, 3), s2[9:12], s2[12:15], s2[15:18], s2[18:21]
        dbg, dba = bg1 - bg0, ba1 - ba0
        U, _, Vt = np.linalg.svd(dR @ exp_so3(JRg @ dbg))
        dRc = U @ Vt
        er = log_so3(dRc.T @ R1.T @ R2)
        ev = R1.T @ (v2 - v1 - g * dt) - (dV + JVg @ dbg + JVa @ dba)
        ep = R1.T @ (t2 - t1 - v1 * dt - g * dt * dt / 2) - (dP + JPg @ dbg + JPa @ dba)
        e9 = np.concatenate([er, ev, ep])
        c9 = float(e9 @ a["in_info"][m].reshape(9, 9) @ e9)
        total += huber(c9, np.sqrt(16.92)) if a["in_robust"][m] else c9
        eg, ea = bg2 - bg1, ba2 - ba1
        total += float(eg @ a["in_info_g"][m].reshape(3, 3) @ eg) + float(ea @ a["in_info_a"][m].reshape(3, 3) @ ea)
    assert abs(total - st.err) <= 1e-9 * abs(st.err), (total, st.err)


def _numpy_inertial_lm_model(win, iterations, lambda_init, max_trials=100):
    """Optimizer::LocalInertialBA's optimisation on the vendored Levenberg-Marquardt, numpy only, no oracle code: residuals as in the test
    above, NUMERIC Jacobians through the vertex updates of the reference (ImuCamPose::Update: twb += Rwb ut, Rwb = Rwb ExpSO3(ur),
    G2oTypes.cc:192-222, ExpSO3 with its nearest-rotation clean-up :986-1008; velocity / biases / points additive), the DENSE system over
    every free vertex solved by numpy.linalg.solve (no Schur complement), robustInformation = rho' Omega (base_edge.h:96-102).
    Returns (kf states, points, err, err_end, trials, per-edge visual chi2 of the last evaluated estimate)."""
    a, d = win.arrays, win.d
    Rcb, tcb = win.Rcb.reshape(3, 3), win.tcb
    fx, fy, cx, cy, bf = [float(x) for x in win.cam]
    model, kb = int(d.get("camera_model", 0)), d.get("kb", (0, 0, 0, 0))
    Trl = cam2 = kb2 = None; model2 = 0
    if "Trl" in d:
        Trl = np.asarray(d["Trl"], np.float64).reshape(3, 4); cam2 = [float(x) for x in d["cam2"]]
        model2 = int(d.get("camera2_model", 0)); kb2 = d.get("kb2", (0, 0, 0, 0))

    smooth = [False]                                                         # difference quotients are taken on the un-rounded angles

    def project(Xc, cam, mdl, kk):
        f0, f1, c0, c1 = cam
        if mdl == 0:
            return np.array([f0 * Xc[0] / Xc[2] + c0, f1 * Xc[1] / Xc[2] + c1])
        if smooth[0]:
            th = np.arctan2(np.sqrt(Xc[0] ** 2 + Xc[1] ** 2), Xc[2]); psi = np.arctan2(Xc[1], Xc[0])
        else:                                                                # KannalaBrandt8.cpp:52-69: atan2f / sqrtf
            th = float(np.float32(np.arctan2(float(np.sqrt(np.float32(Xc[0] ** 2 + Xc[1] ** 2))), float(np.float32(Xc[2])))))
            psi = float(np.float32(np.arctan2(float(np.float32(Xc[1])), float(np.float32(Xc[0])))))
        r = th + kk[0] * th ** 3 + kk[1] * th ** 5 + kk[2] * th ** 7 + kk[3] * th ** 9
        return np.array([f0 * r * np.cos(psi) + c0, f1 * r * np.sin(psi) + c1])

    def skew(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])

    def nearest_rot(M):
        U, _, Vt = np.linalg.svd(M)
        return U @ Vt

    def exp_imu(v):                                                          # ImuTypes.cc:48-60
        th2 = float(v @ v); th = np.sqrt(th2); W = skew(v)
        return np.eye(3) + W + 0.5 * W @ W if th < 1e-4 else np.eye(3) + W * np.sin(th) / th + W @ W * (1 - np.cos(th)) / th2

    def exp_g2o(v):                                                          # G2oTypes.cc:991-1008
        th2 = float(v @ v); th = np.sqrt(th2); W = skew(v)
        return nearest_rot(np.eye(3) + W + 0.5 * W @ W if th < 1e-5 else np.eye(3) + W * np.sin(th) / th + W @ W * (1 - np.cos(th)) / th2)

    def log_so3(R):
        w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
        c = (np.trace(R) - 1) * 0.5
        if c > 1 or c < -1:
            return w
        th = np.arccos(c); s = np.sin(th)
        return w if abs(s) < 1e-5 else th * w / s

    nk, L = win.n_kf, win.n_points
    S = {"R": [win.kf0[k, :9].reshape(3, 3).copy() for k in range(nk)], "t": [win.kf0[k, 9:12].copy() for k in range(nk)],
         "v": [win.kf0[k, 12:15].copy() for k in range(nk)], "g": [win.kf0[k, 15:18].copy() for k in range(nk)],
         "a": [win.kf0[k, 18:21].copy() for k in range(nk)], "X": [win.pts0[l].copy() for l in range(L)]}
    # free vertices and their local dimensions
    verts = []
    for k in range(nk):
        if not a["kf_fixed"][k]:
            verts.append(("P", k, 6))
            if a["kf_imu"][k]:
                verts += [("v", k, 3), ("g", k, 3), ("a", k, 3)]
    verts += [("X", l, 3) for l in range(L)]
    off = {}; n = 0
    for (ty, i, dim) in verts:
        off[(ty, i)] = n; n += dim

    def apply(S0, ty, i, dx):
        S1 = {key: list(val) for key, val in S0.items()}
        if ty == "P":
            S1["t"][i] = S0["t"][i] + S0["R"][i] @ dx[3:]
            S1["R"][i] = S0["R"][i] @ exp_g2o(dx[:3])
        else:
            S1[ty][i] = S0[ty][i] + dx
        return S1

    g_ = np.array([0.0, 0.0, -float(np.float32(9.81))])
    th_m, th_s = float(np.float32(np.sqrt(5.991))), float(np.float32(np.sqrt(7.815)))
    edges = []                                                               # (residual fn, Omega, delta or None, vertex keys)
    for e in range(win.n_edges):
        k, l, typ, obs, w = int(a["edge_kf"][e]), int(a["edge_point"][e]), int(a["edge_stereo"][e]), a["edge_obs"][e], float(a["edge_inv_sigma2"][e])

        def rv(S_, k=k, l=l, typ=typ, obs=obs):
            Rwb, twb = S_["R"][k], S_["t"][k]
            Xc = Rcb @ (Rwb.T @ (S_["X"][l] - twb)) + tcb
            if typ == 2:
                return obs[:2] - project(Trl[:, :3] @ Xc + Trl[:, 3], cam2, model2, kb2)
            uv = project(Xc, (fx, fy, cx, cy), model, kb)
            return obs[:2] - uv if typ == 0 else obs - np.array([uv[0], uv[1], uv[0] - bf / Xc[2]])
        m = 3 if typ == 1 else 2
        edges.append((rv, w * np.eye(m), th_s if typ == 1 else th_m, [("P", k), ("X", l)]))
    n_vis = len(edges)
    for m in range(win.n_inertial):
        k1, k2, p = int(a["in_kf1"][m]), int(a["in_kf2"][m]), a["in_preint"][m]

        def ri(S_, k1=k1, k2=k2, p=p):
            dt, dR, dV, dP = p[0], p[1:10].reshape(3, 3), p[10:13], p[13:16]
            JRg, JVg, JVa, JPg, JPa = [p[16 + 9 * i:25 + 9 * i].reshape(3, 3) for i in range(5)]
            dbg, dba = S_["g"][k1] - p[61:64], S_["a"][k1] - p[64:67]
            R1 = S_["R"][k1]
            er = log_so3(nearest_rot(dR @ exp_imu(JRg @ dbg)).T @ R1.T @ S_["R"][k2])
            ev = R1.T @ (S_["v"][k2] - S_["v"][k1] - g_ * dt) - (dV + JVg @ dbg + JVa @ dba)
            ep = R1.T @ (S_["t"][k2] - S_["t"][k1] - S_["v"][k1] * dt - g_ * dt * dt / 2) - (dP + JPg @ dbg + JPa @ dba)
            return np.concatenate([er, ev, ep])
        edges.append((ri, a["in_info"][m].reshape(9, 9), np.sqrt(16.92) if a["in_robust"][m] else None,
                      [("P", k1), ("v", k1), ("g", k1), ("a", k1), ("P", k2), ("v", k2)]))
        edges.append((lambda S_, k1=k1, k2=k2: S_["g"][k2] - S_["g"][k1], a["in_info_g"][m].reshape(3, 3), None, [("g", k1), ("g", k2)]))
        edges.append((lambda S_, k1=k1, k2=k2: S_["a"][k2] - S_["a"][k1], a["in_info_a"][m].reshape(3, 3), None, [("a", k1), ("a", k2)]))

    def rho(c, delta):
        if delta is None:
            return c, 1.0
        d2 = float(np.float32(delta * delta))
        return (c, 1.0) if c <= d2 else (2 * np.sqrt(c) * delta - d2, delta / np.sqrt(c))

    def evaluate(S_):
        rs = [f(S_) for (f, _, _, _) in edges]
        cs = [float(r @ Om @ r) for r, (_, Om, _, _) in zip(rs, edges)]
        return rs, cs, sum(rho(c, dl)[0] for c, (_, _, dl, _) in zip(cs, edges))

    dims = {(ty, i): dim for (ty, i, dim) in verts}
    h = 1e-6
    rs, cs, cur = evaluate(S)
    err0 = cur
    lam = lambda_init; ni = 2.0; nbad = 0; trials = 0
    cs_eval = cs
    for it in range(iterations):
        rs, cs, cur = evaluate(S); cs_eval = cs
        ini = cur
        H = np.zeros((n, n)); b = np.zeros(n)
        for (f, Om, dl, keys), r, c in zip(edges, rs, cs):
            cols, J = [], []
            for key in keys:
                if key not in off:
                    continue
                dim = dims[key]; Jk = np.zeros((len(r), dim))
                smooth[0] = True
                for q in range(dim):
                    dx = np.zeros(dim); dx[q] = h
                    Jk[:, q] = (f(apply(S, key[0], key[1], dx)) - f(apply(S, key[0], key[1], -dx))) / (2 * h)
                smooth[0] = False
                cols += list(range(off[key], off[key] + dim)); J.append(Jk)
            if not cols:
                continue
            Je = np.hstack(J); W = rho(c, dl)[1] * Om
            H[np.ix_(cols, cols)] += Je.T @ W @ Je
            b[cols] += -Je.T @ W @ r
        q_ = 0
        while True:
            dxs = np.linalg.solve(H + lam * np.eye(n), b)
            St = S
            for (ty, i, dim) in verts:
                St = apply(St, ty, i, dxs[off[(ty, i)]:off[(ty, i)] + dim])
            rs_t, cs_t, tmp = evaluate(St); cs_eval = cs_t
            r_ = (cur - tmp) / (float(dxs @ (lam * dxs + b)) + 1e-3)
            if r_ > 0 and np.isfinite(tmp):
                lam *= max(1 / 3, min(1 - (2 * r_ - 1) ** 3, 2 / 3)); ni = 2.0; cur = tmp; S = St
            else:
                lam *= ni; ni *= 2
            q_ += 1; trials += 1
            if not (r_ < 0 and q_ < max_trials):
                break
        if q_ == max_trials or r_ == 0:
            break
        nbad = nbad + 1 if (ini - cur) * 1e3 < ini else 0
        if nbad >= 3:
            break
    err_end = sum(rho(c, dl)[0] for c, (_, _, dl, _) in zip(cs_eval, edges))
    kf = np.stack([np.concatenate([S["R"][k].reshape(-1), S["t"][k], S["v"][k], S["g"][k], S["a"][k]]) for k in range(nk)])
    return kf, np.stack(S["X"]), err0, err_end, trials, np.array(cs_eval[:n_vis]), it + 1


@pytest.mark.parametrize("kw", [dict(n_opt=3, n_fixed_vis=2, n_points=30, stereo_frac=0.5), dict(n_opt=3, n_fixed_vis=2, n_points=24, fisheye_rig=True)])
def test_whole_inertial_lm_against_an_independent_numpy_model(kw):
    """The whole optimize(10) of a small LocalInertialBA window against the dense numpy model: the same LM trials and iterations, keyframe
    states and points to 1e-5 (rotations, positions, velocities, biases), err / err_end, and the visual outlier flags
    (Optimizer.cc:5058-5090, chi2 of the last evaluated estimate) for every edge not within 1e-4 of its gate."""
    win = ib.make_window(91, **kw)
    kf_o, pts_o, out_o, st = ib.solve(win)
    kf_m, pts_m, err0, err_end, trials, cs_vis, its = _numpy_inertial_lm_model(win, 10, 1.0)
    assert st.failed == 0
    assert abs(err0 - st.err) <= 1e-9 * st.err
    assert (its, trials) == (st.iterations_run, st.lm_trials), (its, trials, st.iterations_run, st.lm_trials)
    assert abs(err_end - st.err_end) <= 1e-5 * st.err_end, (err_end, st.err_end)
    assert np.max(np.abs(kf_m - kf_o.reshape(-1, 21))) <= 1e-5 and np.max(np.abs(pts_m - pts_o.reshape(-1, 3))) <= 1e-4
    typ = win.arrays["edge_stereo"]; close = win.arrays["edge_close"].astype(bool)
    gate = np.where(typ == 1, 7.815, np.where(close, 1.5 * 5.991, 5.991))
    clear = np.abs(cs_vis - gate) > 1e-4
    np.testing.assert_array_equal(out_o.astype(bool)[clear & (typ == 1)], (cs_vis > gate)[clear & (typ == 1)])
    mono = clear & (typ != 1)
    assert (out_o.astype(bool)[mono] >= (cs_vis > gate)[mono]).all()            # (a mono edge is also erased when its depth is not positive)
