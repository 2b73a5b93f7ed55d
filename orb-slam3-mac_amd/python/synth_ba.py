"""Seeded synthetic local-BA graphs (BASELINE config #4).  Host-side input
generator only (numpy); not part of the oracle and not part of the measured path.

P keyframes on a noisy circle (radius 5 m) looking at the centroid, Pinhole fx=fy=458,
cx=320, cy=240; KFs 0,1 fixed; L points uniform in a 4 m cube; each point observed by `obs`
keyframes that see it in-image with positive depth; octave ~ U{0..7} -> invSigma2 =
(float)1.2^-2oct; pixel noise N(0, 1.2^oct); 5 % gross outliers (+-30 px); initial poses
perturbed by exp(N(0,[0.01 rad, 0.05 m])), points by N(0, 0.05 m); everything rounded through
float32 at the boundary.
"""
import numpy as np


def kb8_project(P, fx, fy, cx, cy, k):
    """KannalaBrandt8 projection of camera-frame points P [n,3] (generator side, float64)."""
    r_xy = np.sqrt(P[:, 0] ** 2 + P[:, 1] ** 2)
    th = np.arctan2(r_xy, P[:, 2]); psi = np.arctan2(P[:, 1], P[:, 0])
    r = th + k[0] * th ** 3 + k[1] * th ** 5 + k[2] * th ** 7 + k[3] * th ** 9
    return fx * r * np.cos(psi) + cx, fy * r * np.sin(psi) + cy


def _quat_from_R(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        q = [(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = [0, 0, 0, 0]
        q[i] = 0.5 * s
        s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
    q = np.array(q)
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def _R_from_quat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _exp_so3(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def make_graph(n_kf=50, n_pts=2000, obs=10, seed=0, n_fixed=2, outlier_frac=0.05, stereo_frac=0.0,
               pose_noise=(0.01, 0.05), point_noise=0.05, pixel_noise=1.0, kb8=None, rig2=None, right_frac=0.5, cameras=None, pose_camera=None):
    """kb8 = (k1..k4): the monocular observations come from a KannalaBrandt8 camera (fisheye), else Pinhole.
    rig2 = dict(Trl=(qx,qy,qz,qw,tx,ty,tz), cam=(fx,fy,cx,cy), kb=(k1..k4)|None): a second, rigidly attached camera; a
    fraction right_frac of the observations is also seen there (edge type 2, EdgeSE3ProjectXYZToBody).
    cameras = [dict(fx, fy, cx, cy, bf, kb=None|(k1..k4), rig2=None|dict as above, stereo_frac), ...] + pose_camera [n_kf]: per-keyframe
    calibration (an Atlas window built from several cameras; Optimizer.cc:1961, :1990-1994, :2021-2023): keyframe i observes through
    cameras[pose_camera[i]].  The returned dict then carries `cameras` / `pose_camera` and the single-calibration fields are camera 0's."""
    rng = np.random.default_rng(seed)
    fx = fy = 458.0
    cx, cy, W, H = 320.0, 240.0, 640, 480
    bf = 458.0 * 0.11
    if cameras is not None:
        return _make_graph_multicam(rng, n_kf, n_pts, obs, n_fixed, outlier_frac, pose_noise, point_noise, pixel_noise, right_frac, cameras,
                                    np.asarray(pose_camera, np.int32))
    # ground-truth poses (world -> camera)
    Rs, ts = [], []
    for i in range(n_kf):
        ang = 2 * np.pi * i / n_kf + rng.normal(0, 0.02)
        C = np.array([5 * np.cos(ang), rng.normal(0, 0.2), 5 * np.sin(ang)]) + rng.normal(0, 0.05, 3)
        z = -C / np.linalg.norm(C)
        up = np.array([0.0, -1.0, 0.0])
        x = np.cross(up, z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])                   # rows = camera axes in world
        Rs.append(R); ts.append(-R @ C)
    Rs, ts = np.array(Rs), np.array(ts)
    X = rng.uniform(-2, 2, (n_pts, 3))
    e_pose, e_point, e_obs, e_is2, e_st = [], [], [], [], []
    sig2 = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2))]).astype(np.float32)) ** 2
    inv_sig2 = (np.float32(1.0) / sig2.astype(np.float32)).astype(np.float32)
    for l in range(n_pts):
        Pc = Rs @ X[l] + ts
        if kb8 is not None:
            u, v = kb8_project(Pc, fx, fy, cx, cy, kb8)
        else:
            u = fx * Pc[:, 0] / Pc[:, 2] + cx
            v = fy * Pc[:, 1] / Pc[:, 2] + cy
        vis = np.nonzero((Pc[:, 2] > 0.1) & (u > 0) & (u < W) & (v > 0) & (v < H))[0]
        if len(vis) == 0:
            vis = np.arange(n_kf)
        sel = np.sort(rng.choice(vis, size=min(obs, len(vis)), replace=False))
        for k in sel:
            octv = int(rng.integers(0, 8))
            sd = pixel_noise * 1.2 ** octv
            uu = u[k] + rng.normal(0, sd)
            vv = v[k] + rng.normal(0, sd)
            if rng.random() < outlier_frac:
                uu += rng.choice([-30.0, 30.0]); vv += rng.choice([-30.0, 30.0])
            st = rng.random() < stereo_frac
            ur = uu - bf / Pc[k, 2] + (rng.normal(0, sd) if st else 0.0)
            e_pose.append(k); e_point.append(l); e_st.append(1 if st else 0)
            e_obs.append([np.float32(uu), np.float32(vv), np.float32(ur) if st else 0.0])
            e_is2.append(float(inv_sig2[octv]))
            if rig2 is not None and rng.random() < right_frac:
                Rrl = _R_from_quat(np.array(rig2["Trl"][:4])); trl = np.array(rig2["Trl"][4:])
                Xr = Rrl @ Pc[k] + trl
                f2x, f2y, c2x, c2y = rig2["cam"]
                if rig2.get("kb") is not None:
                    u2, v2 = kb8_project(Xr[None, :], f2x, f2y, c2x, c2y, rig2["kb"]); u2, v2 = float(u2[0]), float(v2[0])
                else:
                    u2, v2 = f2x * Xr[0] / Xr[2] + c2x, f2y * Xr[1] / Xr[2] + c2y
                o2 = int(rng.integers(0, 8)); s2 = pixel_noise * 1.2 ** o2
                u2 += rng.normal(0, s2); v2 += rng.normal(0, s2)
                if rng.random() < outlier_frac:
                    u2 += rng.choice([-30.0, 30.0])
                e_pose.append(k); e_point.append(l); e_st.append(2)
                e_obs.append([np.float32(u2), np.float32(v2), 0.0]); e_is2.append(float(inv_sig2[o2]))
    # perturbed initial estimates, float32-rounded
    poses0 = np.zeros((n_kf, 7))
    poses_gt = np.zeros((n_kf, 7))
    for i in range(n_kf):
        poses_gt[i, :4] = _quat_from_R(Rs[i]); poses_gt[i, 4:] = ts[i]
        if i < n_fixed:
            R0, t0 = Rs[i], ts[i]
        else:
            dR = _exp_so3(rng.normal(0, pose_noise[0], 3))
            R0 = dR @ Rs[i]
            t0 = dR @ ts[i] + rng.normal(0, pose_noise[1], 3)
        R0 = R0.astype(np.float32).astype(np.float64)
        t0 = t0.astype(np.float32).astype(np.float64)
        poses0[i, :4] = _quat_from_R(R0); poses0[i, 4:] = t0
    pts0 = (X + rng.normal(0, point_noise, X.shape)).astype(np.float32).astype(np.float64)
    fixed = np.zeros(n_kf, np.uint8); fixed[:n_fixed] = 1
    return dict(n_poses=n_kf, n_points=n_pts, n_edges=len(e_pose), pose_fixed=fixed,
                edge_pose=np.array(e_pose, np.int32), edge_point=np.array(e_point, np.int32),
                edge_obs=np.array(e_obs, np.float64).reshape(-1, 3), edge_inv_sigma2=np.array(e_is2, np.float64),
                edge_stereo=np.array(e_st, np.uint8), fx=fx, fy=fy, cx=cx, cy=cy, bf=bf, kb=kb8, rig2=rig2,
                poses0=poses0, points0=pts0, poses_gt=poses_gt, points_gt=X.copy())


def _make_graph_multicam(rng, n_kf, n_pts, obs, n_fixed, outlier_frac, pose_noise, point_noise, pixel_noise, right_frac, cameras, pose_camera):
    W, H = 640, 480
    Rs, ts = [], []
    for i in range(n_kf):
        ang = 2 * np.pi * i / n_kf + rng.normal(0, 0.02)
        C = np.array([5 * np.cos(ang), rng.normal(0, 0.2), 5 * np.sin(ang)]) + rng.normal(0, 0.05, 3)
        z = -C / np.linalg.norm(C)
        up = np.array([0.0, -1.0, 0.0])
        x = np.cross(up, z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])
        Rs.append(R); ts.append(-R @ C)
    Rs, ts = np.array(Rs), np.array(ts)
    X = rng.uniform(-2, 2, (n_pts, 3))
    sig2 = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2))]).astype(np.float32)) ** 2
    inv_sig2 = (np.float32(1.0) / sig2.astype(np.float32)).astype(np.float32)

    def proj(c, P):
        if c.get("kb") is not None:
            u, v = kb8_project(P[None, :], c["fx"], c["fy"], c["cx"], c["cy"], c["kb"])
            return float(u[0]), float(v[0])
        return c["fx"] * P[0] / P[2] + c["cx"], c["fy"] * P[1] / P[2] + c["cy"]
    e_pose, e_point, e_obs, e_is2, e_st = [], [], [], [], []
    for l in range(n_pts):
        Pc = Rs @ X[l] + ts
        uv = np.array([proj(cameras[pose_camera[k]], Pc[k]) if Pc[k, 2] > 0.1 else (-1.0, -1.0) for k in range(n_kf)])
        vis = np.nonzero((Pc[:, 2] > 0.1) & (uv[:, 0] > 0) & (uv[:, 0] < W) & (uv[:, 1] > 0) & (uv[:, 1] < H))[0]
        if len(vis) == 0:
            vis = np.nonzero(Pc[:, 2] > 0.1)[0]
        sel = np.sort(rng.choice(vis, size=min(obs, len(vis)), replace=False))
        for k in sel:
            c = cameras[pose_camera[k]]
            octv = int(rng.integers(0, 8))
            sd = pixel_noise * 1.2 ** octv
            uu = uv[k, 0] + rng.normal(0, sd); vv = uv[k, 1] + rng.normal(0, sd)
            if rng.random() < outlier_frac:
                uu += rng.choice([-30.0, 30.0]); vv += rng.choice([-30.0, 30.0])
            st = c.get("kb") is None and rng.random() < c.get("stereo_frac", 0.0)
            ur = uu - c["bf"] / Pc[k, 2] + (rng.normal(0, sd) if st else 0.0)
            e_pose.append(k); e_point.append(l); e_st.append(1 if st else 0)
            e_obs.append([np.float32(uu), np.float32(vv), np.float32(ur) if st else 0.0]); e_is2.append(float(inv_sig2[octv]))
            r2 = c.get("rig2")
            if r2 is not None and rng.random() < right_frac:
                Rrl = _R_from_quat(np.array(r2["Trl"][:4])); trl = np.array(r2["Trl"][4:])
                Xr = Rrl @ Pc[k] + trl
                u2, v2 = proj(dict(fx=r2["cam"][0], fy=r2["cam"][1], cx=r2["cam"][2], cy=r2["cam"][3], kb=r2.get("kb")), Xr)
                o2 = int(rng.integers(0, 8)); s2 = pixel_noise * 1.2 ** o2
                u2 += rng.normal(0, s2); v2 += rng.normal(0, s2)
                e_pose.append(k); e_point.append(l); e_st.append(2)
                e_obs.append([np.float32(u2), np.float32(v2), 0.0]); e_is2.append(float(inv_sig2[o2]))
    poses0 = np.zeros((n_kf, 7)); poses_gt = np.zeros((n_kf, 7))
    for i in range(n_kf):
        poses_gt[i, :4] = _quat_from_R(Rs[i]); poses_gt[i, 4:] = ts[i]
        if i < n_fixed:
            R0, t0 = Rs[i], ts[i]
        else:
            dR = _exp_so3(rng.normal(0, pose_noise[0], 3))
            R0 = dR @ Rs[i]; t0 = dR @ ts[i] + rng.normal(0, pose_noise[1], 3)
        R0 = R0.astype(np.float32).astype(np.float64); t0 = t0.astype(np.float32).astype(np.float64)
        poses0[i, :4] = _quat_from_R(R0); poses0[i, 4:] = t0
    pts0 = (X + rng.normal(0, point_noise, X.shape)).astype(np.float32).astype(np.float64)
    fixed = np.zeros(n_kf, np.uint8); fixed[:n_fixed] = 1
    c0 = cameras[0]
    return dict(n_poses=n_kf, n_points=n_pts, n_edges=len(e_pose), pose_fixed=fixed,
                edge_pose=np.array(e_pose, np.int32), edge_point=np.array(e_point, np.int32),
                edge_obs=np.array(e_obs, np.float64).reshape(-1, 3), edge_inv_sigma2=np.array(e_is2, np.float64),
                edge_stereo=np.array(e_st, np.uint8), fx=c0["fx"], fy=c0["fy"], cx=c0["cx"], cy=c0["cy"], bf=c0["bf"], kb=c0.get("kb"), rig2=c0.get("rig2"),
                cameras=cameras, pose_camera=pose_camera, poses0=poses0, points0=pts0, poses_gt=poses_gt, points_gt=X.copy())


def make_pose_problem(seed, n=800, stereo_frac=0.0, outlier_frac=0.1, noise=True, perturb=(0.02, 0.08), kb8=None, rig2=None, right_frac=0.4):
    """Seeded synthetic Optimizer::PoseOptimization input (host-side generator, numpy only).

    Camera at a random pose looking at a cloud of n map points (float32-rounded, as Xw.at<float>), Pinhole
    fx=fy=458, cx=320, cy=240, bf=40; observations = projections + N(0, 1.2^octave) px, octave ~ U{0..7},
    invSigma2 = (float)1.2^-2oct; a fraction are gross outliers (+-40 px); a fraction carry a right-image
    coordinate (stereo edge).  Returns dict(Xw, obs [n,3], inv_sigma2, cam, pose0 (perturbed), pose_true)."""
    rng = np.random.default_rng(seed)
    fx = fy = 458.0; cx, cy, bf = 320.0, 240.0, 40.0
    ang = rng.normal(0, 0.3, 3)
    th = np.linalg.norm(ang)
    K = np.array([[0, -ang[2], ang[1]], [ang[2], 0, -ang[0]], [-ang[1], ang[0], 0]])
    R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
    t = rng.normal(0, 1.0, 3)
    # points in the camera frustum, then to world
    z = rng.uniform(2.0, 12.0, n)
    u = rng.uniform(20, 620, n); v = rng.uniform(20, 460, n)
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    Xw = ((Xc - t) @ R).astype(np.float32).astype(np.float64)            # Xw = R^T (Xc - t)
    Xc = Xw @ R.T + t
    octv = rng.integers(0, 8, n)
    sig = 1.2 ** octv
    uu = fx * Xc[:, 0] / Xc[:, 2] + cx; vv = fy * Xc[:, 1] / Xc[:, 2] + cy
    if kb8 is not None:
        uu, vv = kb8_project(Xc, fx, fy, cx, cy, kb8)
    ur = uu - bf / Xc[:, 2]
    if noise:
        uu = uu + rng.normal(0, 1, n) * sig; vv = vv + rng.normal(0, 1, n) * sig; ur = ur + rng.normal(0, 1, n) * sig
    bad = rng.random(n) < outlier_frac
    uu = np.where(bad, uu + rng.choice([-40, 40], n), uu); vv = np.where(bad, vv + rng.choice([-40, 40], n), vv)
    is_st = rng.random(n) < stereo_frac
    right = np.zeros(n, np.uint8)
    if rig2 is not None:                                             # a fraction of the points is observed in the second camera instead
        right = (rng.random(n) < right_frac).astype(np.uint8)
        Rrl = _R_from_quat(np.array(rig2["Trl"][:4])); Xr = Xc @ Rrl.T + np.array(rig2["Trl"][4:])
        f2x, f2y, c2x, c2y = rig2["cam"]
        if rig2.get("kb") is not None:
            u2, v2 = kb8_project(Xr, f2x, f2y, c2x, c2y, rig2["kb"])
        else:
            u2, v2 = f2x * Xr[:, 0] / Xr[:, 2] + c2x, f2y * Xr[:, 1] / Xr[:, 2] + c2y
        if noise:
            u2 = u2 + rng.normal(0, 1, n) * sig; v2 = v2 + rng.normal(0, 1, n) * sig
        u2 = np.where(bad, u2 + 40.0, u2)
        uu = np.where(right == 1, u2, uu); vv = np.where(right == 1, v2, vv)
        is_st = is_st & (right == 0)
    obs = np.stack([uu, vv, np.where(is_st, np.maximum(ur, 0.5), -1.0)], 1).astype(np.float32).astype(np.float64)
    inv_s2 = (1.0 / (np.float32(1.2) ** octv.astype(np.float32)) ** 2).astype(np.float32).astype(np.float64)
    q_true = _quat_from_R(R)
    d = np.concatenate([rng.normal(0, perturb[0], 3), rng.normal(0, perturb[1], 3)])
    th = np.linalg.norm(d[:3])
    Kd = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
    Rd = np.eye(3) + (np.sin(th) / th * Kd + (1 - np.cos(th)) / th ** 2 * Kd @ Kd if th > 0 else 0)
    R0 = (Rd @ R).astype(np.float32).astype(np.float64); t0 = (Rd @ t + d[3:]).astype(np.float32).astype(np.float64)
    U, _, Vt = np.linalg.svd(R0); R0 = U @ Vt                               # Converter::toSE3Quat gets a float Tcw
    return dict(Xw=Xw, obs=obs, inv_sigma2=inv_s2, cam=(fx, fy, cx, cy, bf), kb8=kb8, rig2=rig2, right=right if rig2 is not None else None, pose0=np.concatenate([_quat_from_R(R0), t0]),
                pose_true=np.concatenate([q_true, t]), outlier_true=bad)
