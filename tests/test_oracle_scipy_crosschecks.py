"""Cross-checks of the optimisation oracles against libraries that were NOT written for this repository (CPU tests, no GPU).

The numpy models of tests/test_oracle_match_ba.py restate the vendored g2o algorithms step by step -- same author, same reading of the
text.  Here the END RESULTS are compared with general-purpose solvers instead:
  * Optimizer::PoseOptimization (oracle/ba_oracle.c) against scipy.optimize.least_squares on the same residuals: on outlier-free data
    the Huber kernels stay inactive (every chi2 far below 5.991 / 7.815) and four rounds of ten LM iterations from the same start are a
    converged least-squares fit, so both must reach the same pose;
  * the local bundle adjustment oracle (5 + 10 LM iterations, Schur complement on the points, fixed gauge keyframes) against
    least_squares over all free poses and points at once (dense numeric Jacobian, no Schur complement, scipy's own trust-region LM);
  * the restated one-sided Jacobi cv::SVD behind KannalaBrandt8::Triangulate against numpy.linalg.svd in double precision.
None of this pins the arithmetic (rounding stays unpinned: DESIGN.md 2); it removes the cheapest ways for the oracle to be wrong about
the optimum: a sign in a Jacobian, a wrong information weight, a mis-ordered se(3) update, a Schur-complement slip."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

scipy_optimize = pytest.importorskip("scipy.optimize")


def _quat_to_R(q):
    x, y, z, w = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _rodrigues(w):
    """rotation vectors [..., 3] -> matrices [..., 3, 3] (scipy.spatial would do; kept local so that the residual function vectorises)"""
    w = np.asarray(w, np.float64)
    th = np.linalg.norm(w, axis=-1)[..., None, None]
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -w[..., 2], w[..., 1], w[..., 2]
    K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -w[..., 0], -w[..., 1], w[..., 0]
    small = th < 1e-9
    ths = np.where(small, 1.0, th)
    A = np.where(small, 1.0, np.sin(ths) / ths)
    B = np.where(small, 0.5, (1 - np.cos(ths)) / ths ** 2)
    return np.eye(3) + A * K + B * (K @ K)


@pytest.mark.parametrize("stereo_frac,seed", [(0.0, 11), (0.5, 12), (1.0, 13)])
def test_pose_optimization_oracle_reaches_scipys_least_squares_optimum(stereo_frac, seed):
    """Optimizer::PoseOptimization (src/Optimizer.cc:854-1168; EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose) on outlier-free
    observations == scipy.optimize.least_squares (MINPACK lmder through method='lm', numeric Jacobian) on the information-weighted
    reprojection residuals, pose parameterised as exp(rotation vector) R0, t0 + dt: rotation to 1e-7, translation to 1e-6 of the scene
    scale (the stereo edge's float 1 / z of types_six_dof_expmap.cpp:339-346 moves u by 1e-7 relative, no more)."""
    import synth_ba
    import oracle_ba_bind as obb
    prob = synth_ba.make_pose_problem(seed, n=240, stereo_frac=stereo_frac, outlier_frac=0.0, noise=False)
    Xw, obs, is2, cam, pose0 = prob["Xw"], prob["obs"].copy(), prob["inv_sigma2"], prob["cam"], prob["pose0"]
    # noise at 0.3 sigma of the keypoint's octave: chi2 = 0.09 chi-square(2 or 3) stays far below the Huber deltas (at 1 sigma one
    # observation in twenty crosses 5.991 by construction and the comparison would be one of robust kernels)
    rng = np.random.default_rng(seed)
    st_mask = obs[:, 2] >= 0
    obs[:, :2] += rng.normal(0, 0.3, (len(obs), 2)) / np.sqrt(is2)[:, None]
    obs[st_mask, 2] += (rng.normal(0, 0.3, len(obs)) / np.sqrt(is2))[st_mask]
    obs = obs.astype(np.float32).astype(np.float64)
    fx, fy, cx, cy, bf = [float(c) for c in cam]
    n_in, pose, out, st = obb.pose_optimization(Xw, obs, is2, cam, pose0)
    assert n_in == len(Xw) and not out.any()
    R0, t0 = _quat_to_R(pose0[:4]), np.asarray(pose0[4:], np.float64)
    stereo = obs[:, 2] >= 0
    w = np.sqrt(is2)

    def residuals(x):
        R = _rodrigues(x[:3]) @ R0
        Xc = Xw @ R.T + t0 + x[3:]
        u = fx * Xc[:, 0] / Xc[:, 2] + cx
        v = fy * Xc[:, 1] / Xc[:, 2] + cy
        r = [w * (obs[:, 0] - u), w * (obs[:, 1] - v), (w * (obs[:, 2] - (u - bf / Xc[:, 2])))[stereo]]
        return np.concatenate(r)

    sol = scipy_optimize.least_squares(residuals, np.zeros(6), method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=2000)
    assert sol.success
    R_s = _rodrigues(sol.x[:3]) @ R0; t_s = t0 + sol.x[3:]
    R_o, t_o = _quat_to_R(pose[:4]), np.asarray(pose[4:])
    # every chi2 at the optimum is far inside the Huber region: the robust kernel never acted
    res = residuals(sol.x)
    chi2 = res[:len(Xw)] ** 2 + res[len(Xw):2 * len(Xw)] ** 2
    assert chi2.max() < 5.0
    assert np.abs(R_o - R_s).max() < 1e-7, np.abs(R_o - R_s).max()
    assert np.abs(t_o - t_s).max() < 1e-6, np.abs(t_o - t_s).max()
    # and the optimum is the TRUE pose up to the noise: neither solver fitted something else
    assert np.abs(t_s - prob["pose_true"][4:]).max() < 0.05


def _ba_residuals_factory(g):
    Rf = np.stack([_quat_to_R(q[:4]) for q in g["poses0"]]); tf = np.asarray(g["poses0"])[:, 4:].copy()
    free = np.flatnonzero(g["pose_fixed"] == 0)
    slot = np.full(g["n_poses"], -1); slot[free] = np.arange(len(free))
    ep, el = g["edge_pose"], g["edge_point"]
    obs, w, st = g["edge_obs"], np.sqrt(g["edge_inv_sigma2"]), g["edge_stereo"] == 1
    fx, fy, cx, cy, bf = g["fx"], g["fy"], g["cx"], g["cy"], g["bf"]
    nfree, L = len(free), g["n_points"]

    def unpack(x):
        R = Rf.copy(); t = tf.copy()
        d = x[:6 * nfree].reshape(nfree, 6)
        R[free] = _rodrigues(d[:, :3]) @ Rf[free]
        t[free] = tf[free] + d[:, 3:]
        X = g["points0"] + x[6 * nfree:].reshape(L, 3)
        return R, t, X

    def residuals(x):
        R, t, X = unpack(x)
        Xc = np.einsum("eij,ej->ei", R[ep], X[el]) + t[ep]
        u = fx * Xc[:, 0] / Xc[:, 2] + cx
        v = fy * Xc[:, 1] / Xc[:, 2] + cy
        return np.concatenate([w * (obs[:, 0] - u), w * (obs[:, 1] - v), (w * (obs[:, 2] - (u - bf / Xc[:, 2])))[st]])
    return residuals, unpack, 6 * nfree + 3 * L


@pytest.mark.parametrize("stereo_frac,seed", [(0.0, 21), (0.4, 22)])
def test_local_ba_oracle_reaches_scipys_least_squares_optimum(stereo_frac, seed):
    """Optimizer::LocalBundleAdjustment's optimisation (src/Optimizer.cc:2046-2122: optimize(5), then optimize(10); block_solver.hpp Schur
    complement; EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ) on an outlier-free window started close to the optimum == one
    scipy.optimize.least_squares fit over ALL free keyframes and points at once (trust-region reflective, dense 2-point Jacobian): same
    final chi2 to 1e-6 relative, same keyframe poses and points to 1e-5."""
    import synth_ba
    import oracle_ba_bind as obb
    g = synth_ba.make_graph(n_kf=6, n_pts=70, obs=4, seed=seed, n_fixed=2, outlier_frac=0.0, stereo_frac=stereo_frac,
                            pose_noise=(0.002, 0.01), point_noise=0.01, pixel_noise=0.3)
    rc, poses, pts, out, st = obb.solve(g)
    assert rc == 0 and not out.any()
    residuals, unpack, nx = _ba_residuals_factory(g)
    sol = scipy_optimize.least_squares(residuals, np.zeros(nx), method="trf", x_scale="jac", xtol=1e-15, ftol=1e-15, gtol=1e-12, max_nfev=400)
    chi2_scipy = 2.0 * sol.cost
    # the same objective to begin with: the oracle reports the Huber-robustified sum (a few perturbed start errors exceed the deltas)
    res0 = residuals(np.zeros(nx))
    E = g["n_edges"]; stm = g["edge_stereo"] == 1
    c0 = res0[:E] ** 2 + res0[E:2 * E] ** 2
    c0[stm] += res0[2 * E:] ** 2
    d2 = np.where(stm, np.float32(np.float32(np.sqrt(7.815)) ** 2), np.float32(np.float32(np.sqrt(5.991)) ** 2)).astype(np.float64)
    rho0 = np.where(c0 <= d2, c0, 2 * np.sqrt(c0 * d2) - d2)
    assert abs(float(rho0.sum()) - st["chi2_initial"]) <= 1e-6 * st["chi2_initial"], (rho0.sum(), st["chi2_initial"])
    # ... and at the optimum no kernel is active, so the final sums are plain sums of squares on both sides
    resf = residuals(sol.x)
    cf = resf[:E] ** 2 + resf[E:2 * E] ** 2
    cf[stm] += resf[2 * E:] ** 2
    assert (cf < d2).all()
    assert st["chi2_final"] < 0.5 * st["chi2_initial"]
    assert abs(st["chi2_final"] - chi2_scipy) <= 1e-6 * chi2_scipy, (st["chi2_final"], chi2_scipy)
    R, t, X = unpack(sol.x)
    Ro = np.stack([_quat_to_R(q[:4]) for q in poses])
    assert np.abs(Ro - R).max() < 1e-5 and np.abs(poses[:, 4:] - t).max() < 1e-5 and np.abs(pts - X).max() < 1e-5
    # the gauge keyframes did not move (the oracle re-normalises their quaternions: 1e-16)
    assert np.abs(poses[:2] - g["poses0"][:2]).max() < 1e-12


def test_kb8_triangulation_svd_against_numpy_svd():
    """KannalaBrandt8::Triangulate (src/CameraModels/KannalaBrandt8.cpp:422-435): the null vector of the 4 x 4 system by the restated
    one-sided Jacobi cv::SVD (oracle/match_oracle.c) == numpy.linalg.svd (LAPACK gesdd) of the same matrix built in double, to the
    float conditioning of the system (relative 2e-3 on the de-homogenised point; both land on the true point within the noise)."""
    import oracle_match_bind as om
    rng = np.random.default_rng(8)
    kb = np.array([190.9, 190.8, 254.9, 256.8, 0.0034, 0.0007, -0.0020, 0.0002], np.float32)
    cam = (1, kb.astype(np.float64))
    checked = 0
    for _ in range(300):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0.03, 0.12)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R12 = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        t12 = rng.uniform(-0.5, 0.5, 3); t12[0] += 0.4 * np.sign(t12[0] or 1)
        X2 = np.array([rng.uniform(-2, 2), rng.uniform(-1.5, 1.5), rng.uniform(2, 8)])
        X1 = R12 @ X2 + t12
        if X1[2] < 0.5:
            continue
        p1 = om.kb8_project_np(cam, X1); p2 = om.kb8_project_np(cam, X2)
        z, x = om.kb8_triangulate_matches(1, kb, 1, kb, p1, p2, R12, t12, 1.0, 1.0)
        if not z > 0:
            continue
        r1 = om.camera_unproject_f(1, kb, np.float32(p1[0]), np.float32(p1[1])).astype(np.float64)
        r2 = om.camera_unproject_f(1, kb, np.float32(p2[0]), np.float32(p2[1])).astype(np.float64)
        R21 = R12.astype(np.float32).astype(np.float64).T; t21 = -R21 @ t12.astype(np.float32).astype(np.float64)
        T1 = np.hstack([np.eye(3), np.zeros((3, 1))]); T2 = np.hstack([R21, t21[:, None]])
        A = np.stack([r1[0] * T1[2] - T1[0], r1[1] * T1[2] - T1[1], r2[0] * T2[2] - T2[0], r2[1] * T2[2] - T2[1]])
        v = np.linalg.svd(A)[2][3]
        x_np = v[:3] / v[3]
        assert np.abs(x - x_np).max() <= 2e-3 * np.abs(x_np).max(), (x, x_np)
        assert np.abs(x_np - X1).max() <= 2e-2 * np.abs(X1).max()
        checked += 1
    assert checked > 150


def test_kb8_unproject_newton_against_scipy_brentq():
    """KannalaBrandt8::unproject (src/CameraModels/KannalaBrandt8.cpp:103-130: ten Newton steps on theta (1 + k1 theta^2 + ... ) = theta_d,
    then tan(theta) / theta_d) restated in the oracle == the root scipy.optimize.brentq brackets for the same polynomial, in double: the
    unit rays agree to 2e-6 (float Newton with the 1e-6 stop), across the image and for both TUM-VI-like coefficient sets."""
    import oracle_match_bind as om
    rng = np.random.default_rng(17)
    for kb in (np.array([190.9, 190.8, 254.9, 256.8, 0.0034, 0.0007, -0.0020, 0.0002], np.float32),
               np.array([190.4, 190.6, 252.7, 255.0, -0.012, 0.031, -0.0019, -0.0003], np.float32)):
        k = kb[4:].astype(np.float64)
        for _ in range(400):
            rad, phi = rng.uniform(0, 250), rng.uniform(0, 2 * np.pi)          # inside the fisheye circle of a 512 x 512 image (theta_d <= 1.32)
            u, v = np.float32(kb[2] + rad * np.cos(phi)), np.float32(kb[3] + rad * np.sin(phi))
            ray = om.camera_unproject_f(1, kb, u, v).astype(np.float64)
            pw = np.array([(float(u) - float(kb[2])) / float(kb[0]), (float(v) - float(kb[3])) / float(kb[1])])
            th_d = min(np.hypot(*pw), np.pi / 2)
            if th_d < 1e-6:
                continue
            f = lambda th: th * (1 + k[0] * th ** 2 + k[1] * th ** 4 + k[2] * th ** 6 + k[3] * th ** 8) - th_d
            th = scipy_optimize.brentq(f, 0.0, 1.56, xtol=1e-14)
            ref = np.array([pw[0] * np.tan(th) / th_d, pw[1] * np.tan(th) / th_d, 1.0])
            a = ray / np.linalg.norm(ray); b = ref / np.linalg.norm(ref)
            assert np.abs(a - b).max() < 2e-6, (u, v, a, b)


def test_undistort_keypoints_against_scipy_root_of_the_brown_model():
    """Frame::UndistortKeyPoints (src/Frame.cc:738-771) = cv::undistortPoints(R = I, P = K): OpenCV's five fixed-point iterations of the
    inverse Brown-Conrady model, restated in the oracle.  Independent check: scipy.optimize.fsolve inverts the FORWARD model (the one
    cv::projectPoints and the calibration define) for every keypoint; with EuRoC's coefficients the five iterations land within 0.002 px
    of the exact inverse inside the central half of the image and within 0.7 px at its corners (3.4.1 stops after five iterations whatever
    the residual: that shortfall is the reference's behaviour and part of the oracle) -- a wrong sign of p1 / p2, a swapped k2 / k3 or
    fx / fy would be off by pixels in the centre as well."""
    import oracle_match_bind as om
    from oracle_bind import KP_DTYPE
    rng = np.random.default_rng(12)
    K = (458.654, 457.296, 367.215, 248.375)
    for dist in ((-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.01)):
        n = 300
        kp = np.zeros(n, KP_DTYPE)
        kp["x"] = rng.uniform(0, 752, n).astype(np.float32); kp["y"] = rng.uniform(0, 480, n).astype(np.float32)
        out = om.undistort_keypoints(kp, K, dist)
        fx, fy, cx, cy = K
        k1, k2, p1, p2 = dist[:4]; k3 = dist[4] if len(dist) > 4 else 0.0

        def forward(xy):
            x, y = xy
            r2 = x * x + y * y
            c = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
            return np.array([x * c + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), y * c + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y])
        worst = 0.0; worst_centre = 0.0
        for i in range(n):
            xd = np.array([(float(kp["x"][i]) - cx) / fx, (float(kp["y"][i]) - cy) / fy])
            sol = scipy_optimize.fsolve(lambda q: forward(q) - xd, xd, xtol=1e-12)
            assert np.abs(forward(sol) - xd).max() < 1e-10
            u, v = fx * sol[0] + cx, fy * sol[1] + cy
            e = float(np.hypot(u - out["x"][i], v - out["y"][i]))
            worst = max(worst, e)
            if abs(kp["x"][i] - cx) < 188 and abs(kp["y"][i] - cy) < 120:
                worst_centre = max(worst_centre, e)
        assert worst < 0.7 and worst_centre < 0.002, (worst, worst_centre)
