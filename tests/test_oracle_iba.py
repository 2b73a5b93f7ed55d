"""CPU tests of the inertial local-BA oracle (oracle/iba_oracle.c): the restated Jacobians (G2oTypes.cc:349-482, :742-800) against
finite differences taken through the restated vertex updates (G2oTypes.cc:192-220), the SO3 helpers, and the LM loop on synthetic
visual-inertial windows.  PARITY UNPINNED (no reference fixtures exist for this path; see oracle/iba_oracle.h)."""
import numpy as np
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
