"""Synthetic visual-inertial windows for the inertial local-BA tests and the bench (no oracle, no GPU code in here): the arrays of
one Optimizer::LocalInertialBA window in the layout of orbhip_iba_window (include/orbhip.h)."""
import numpy as np

KF, PREINT = 21, 67


class Window:
    """Arrays of one problem (kept alive here) + the ctypes struct view."""

    def __init__(self, d):
        self.d = d
        a = self.arrays = {
            "kf_fixed": np.ascontiguousarray(d["kf_fixed"], np.uint8), "kf_imu": np.ascontiguousarray(d["kf_imu"], np.uint8),
            "edge_kf": np.ascontiguousarray(d["edge_kf"], np.int32), "edge_point": np.ascontiguousarray(d["edge_point"], np.int32),
            "edge_obs": np.ascontiguousarray(d["edge_obs"], np.float64), "edge_stereo": np.ascontiguousarray(d["edge_stereo"], np.uint8),
            "edge_inv_sigma2": np.ascontiguousarray(d["edge_inv_sigma2"], np.float64),
            "edge_close": np.ascontiguousarray(d["edge_close"], np.uint8),
            "in_kf1": np.ascontiguousarray(d["in_kf1"], np.int32), "in_kf2": np.ascontiguousarray(d["in_kf2"], np.int32),
            "in_preint": np.ascontiguousarray(d["in_preint"], np.float64), "in_info": np.ascontiguousarray(d["in_info"], np.float64),
            "in_info_g": np.ascontiguousarray(d["in_info_g"], np.float64), "in_info_a": np.ascontiguousarray(d["in_info_a"], np.float64),
            "in_robust": np.ascontiguousarray(d["in_robust"], np.uint8)}
        self.kf0 = np.ascontiguousarray(d["kf_state"], np.float64)
        self.pts0 = np.ascontiguousarray(d["points"], np.float64)
        self.cam = d["cam"]
        self.Rcb = np.ascontiguousarray(d["Rcb"], np.float64)
        self.tcb = np.ascontiguousarray(d["tcb"], np.float64)
        self.n_kf, self.n_points, self.n_edges, self.n_inertial = len(a["kf_fixed"]), len(self.pts0), len(a["edge_kf"]), len(a["in_kf1"])

    def struct(self, cls):
        p = cls()
        p.n_kf, p.n_points, p.n_edges, p.n_inertial = self.n_kf, self.n_points, self.n_edges, self.n_inertial
        for k, v in self.arrays.items():
            setattr(p, k, v.ctypes.data)
        for i in range(9):
            p.Rcb[i] = float(self.Rcb.reshape(-1)[i])
        for i in range(3):
            p.tcb[i] = float(self.tcb[i])
        p.fx, p.fy, p.cx, p.cy, p.bf = [float(x) for x in self.cam]
        p.camera_model = int(self.d.get("camera_model", 0))
        for i in range(4):
            p.kb[i] = float(self.d.get("kb", (0, 0, 0, 0))[i])
        p.has_cam2 = 1 if "Trl" in self.d else 0
        if p.has_cam2:
            for i in range(12):
                p.Trl[i] = float(np.asarray(self.d["Trl"]).reshape(-1)[i])
            p.fx2, p.fy2, p.cx2, p.cy2 = [float(x) for x in self.d["cam2"]]
            p.camera2_model = int(self.d.get("camera2_model", 0))
            for i in range(4):
                p.kb2[i] = float(self.d.get("kb2", (0, 0, 0, 0))[i])
        return p


# ---------------------------------------------------------------------------------------------- synthetic windows
def _rot(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def make_window(seed, n_opt=8, n_fixed_vis=12, n_points=400, stereo_frac=0.5, outlier_frac=0.03, noise_px=0.6,
                state_noise=1.0, large=False, fisheye_rig=False):
    """A temporal window as LocalInertialBA builds it (Optimizer.cc:4574-4868): n_opt consecutive keyframes with IMU states, the
    keyframe before them fixed (with IMU states), n_fixed_vis older fixed keyframes that only see the points.  Keyframe order in
    the arrays: optimizable newest first (vpOptimizableKFs), then the fixed previous keyframe, then the visual-only fixed ones."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy, bf = 458.0, 457.0, 367.0, 248.0, 47.9
    kb = (0.0035, 0.0007, -0.0021, 0.0002)               # KannalaBrandt8 k1..k4 (TUM-VI-like)
    if fisheye_rig:                                      # two KannalaBrandt8 cameras, no rectified stereo (config #5's rig)
        fx, fy, cx, cy, bf = 190.9, 190.8, 254.9, 256.9, 0.0
        stereo_frac = 0.0
        Rrl = _rot(np.array([0.004, -0.012, 0.003]))
        Trl = np.concatenate([Rrl, np.array([[-0.101], [0.0019], [0.0012]])], 1)
        cam2 = (190.4, 190.3, 252.6, 255.0)
        kb2 = (0.0034, 0.0008, -0.0020, 0.0002)

    def kb_project(Xc, f, kk):
        th = np.arctan2(np.hypot(Xc[0], Xc[1]), Xc[2]); psi = np.arctan2(Xc[1], Xc[0])
        r = th + kk[0] * th ** 3 + kk[1] * th ** 5 + kk[2] * th ** 7 + kk[3] * th ** 9
        return f[0] * r * np.cos(psi) + f[2], f[1] * r * np.sin(psi) + f[3]
    Rcb = _rot(np.array([0.02, -0.01, 1.55]))            # camera/body extrinsics of an EuRoC-like rig
    tcb = np.array([0.065, -0.02, 0.01])
    dt_kf = 0.25 if not large else 0.3
    nT = n_opt + 1                                       # temporal keyframes incl. the fixed previous one, oldest first in time
    g = np.array([0, 0, -float(np.float32(9.81))])
    # smooth body trajectory (analytic position -> exact velocity), looking roughly along +x of the world
    tt = np.arange(nT) * dt_kf
    A = rng.uniform(0.3, 0.8, 3) * np.array([1.0, 1.0, 0.3])
    om = rng.uniform(0.5, 1.1, 3)
    ph = rng.uniform(0, 6.28, 3)
    pos = lambda t: np.array([0.6 * t, 0, 0]) + A * np.sin(om * t + ph)
    vel = lambda t: np.array([0.6, 0, 0]) + A * om * np.cos(om * t + ph)
    # body frame: camera looks along world +x; camera z = Rcb-row... build Rwb from a desired camera orientation
    Rwc0 = np.array([[0, 0, 1.0], [-1.0, 0, 0], [0, -1.0, 0]])           # camera z -> world x, camera x -> world -y, camera y -> world -z
    Rwb = [Rwc0 @ _rot(0.12 * np.sin(om * t + ph[::-1])) @ Rcb for t in tt]
    P = [pos(t) for t in tt]
    V = [vel(t) for t in tt]
    bg_true = rng.normal(0, 0.01, 3)
    ba_true = rng.normal(0, 0.05, 3)
    # visual-only fixed keyframes: earlier poses scattered behind the window
    fixedR, fixedP = [], []
    for i in range(n_fixed_vis):
        t = -0.4 * (i + 1)
        fixedR.append(Rwc0 @ _rot(rng.normal(0, 0.08, 3)) @ Rcb)
        fixedP.append(pos(0) + np.array([0.5 * t, 0, 0]) + rng.normal(0, 0.25, 3))
    # array order
    order_t = list(range(nT - 1, 0, -1))                 # optimizable, newest first
    n_kf = n_opt + 1 + n_fixed_vis
    kf_true = np.zeros((n_kf, KF))
    kf_fixed = np.zeros(n_kf, np.uint8)
    kf_imu = np.zeros(n_kf, np.uint8)
    idx_of_t = {}
    for a, ti in enumerate(order_t + [0]):
        idx_of_t[ti] = a
        kf_true[a, 0:9] = Rwb[ti].reshape(-1); kf_true[a, 9:12] = P[ti]; kf_true[a, 12:15] = V[ti]
        kf_true[a, 15:18] = bg_true + rng.normal(0, 1e-4, 3); kf_true[a, 18:21] = ba_true + rng.normal(0, 1e-3, 3)
        kf_imu[a] = 1
    kf_fixed[n_opt] = 1
    for i in range(n_fixed_vis):
        a = n_opt + 1 + i
        kf_true[a, 0:9] = fixedR[i].reshape(-1); kf_true[a, 9:12] = fixedP[i]
        kf_fixed[a] = 1
    # points in front of the cameras
    pts_true = np.stack([pos(tt[-1] / 2)[0] + rng.uniform(3, 14, n_points), rng.uniform(-5, 5, n_points), rng.uniform(-2.5, 2.5, n_points)], 1)
    inv_sigma2_levels = 1.0 / (1.2 ** (2 * np.arange(8)))
    ekf, ept, eobs, est, eis2, eclose = [], [], [], [], [], []
    for l in range(n_points):
        X = pts_true[l]
        seen = 0
        for a in range(n_kf):
            R = kf_true[a, 0:9].reshape(3, 3); t = kf_true[a, 9:12]
            Xc = Rcb @ (R.T @ (X - t)) + tcb
            if Xc[2] < 0.5:
                continue
            if fisheye_rig:
                u, v = kb_project(Xc, (fx, fy, cx, cy), kb)
                if not (0 < u < 512 and 0 < v < 512):
                    continue
            else:
                u, v = fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy
                if not (0 < u < 752 and 0 < v < 480):
                    continue
            octv = int(rng.integers(0, 8))
            s = noise_px * 1.2 ** octv
            close = 1 if Xc[2] < 10 else 0
            if rng.random() >= 0.25:                                     # left / main camera observation
                is_st = rng.random() < stereo_frac and Xc[2] < 40 * bf / fx
                ob = np.array([u + rng.normal(0, s), v + rng.normal(0, s), (u - bf / Xc[2] + rng.normal(0, s)) if is_st else -1.0])
                if rng.random() < outlier_frac:
                    ob[:2] += rng.uniform(-25, 25, 2)
                ob = ob.astype(np.float32).astype(np.float64)            # cv::KeyPoint / mvuRight are float
                ekf.append(a); ept.append(l); eobs.append(ob); est.append(1 if is_st else 0)
                eis2.append(float(np.float32(inv_sigma2_levels[octv]))); eclose.append(close)
                seen += 1
            if fisheye_rig and rng.random() < 0.6:                       # right camera: EdgeMono(1) on the same pose vertex
                Xr = Trl[:, :3] @ Xc + Trl[:, 3]
                if Xr[2] > 0.5:
                    ur, vr = kb_project(Xr, cam2, kb2)
                    if 0 < ur < 512 and 0 < vr < 512:
                        ob = np.array([ur + rng.normal(0, s), vr + rng.normal(0, s), -1.0])
                        if rng.random() < outlier_frac:
                            ob[:2] += rng.uniform(-25, 25, 2)
                        ob = ob.astype(np.float32).astype(np.float64)
                        ekf.append(a); ept.append(l); eobs.append(ob); est.append(2)
                        eis2.append(float(np.float32(inv_sigma2_levels[octv]))); eclose.append(close)
    # inertial edges: keyframe at time ti (ti >= 1) to ti - 1; exact preintegrated deltas + noise, integrated at a bias slightly off
    in1, in2, pre, info, infog, infoa, rob = [], [], [], [], [], [], []
    for a, ti in enumerate(order_t):
        k2, k1 = idx_of_t[ti], idx_of_t[ti - 1]
        R1, R2 = Rwb[ti - 1], Rwb[ti]
        dt = dt_kf
        b_lin_g = bg_true + rng.normal(0, 2e-3, 3)
        b_lin_a = ba_true + rng.normal(0, 1e-2, 3)
        JRg = -dt * (np.eye(3) + rng.normal(0, 0.03, (3, 3)))
        JVa = -dt * (np.eye(3) + rng.normal(0, 0.05, (3, 3)))
        JPa = -0.5 * dt * dt * (np.eye(3) + rng.normal(0, 0.05, (3, 3)))
        JVg = rng.normal(0, 0.3 * dt * dt, (3, 3))
        JPg = rng.normal(0, 0.1 * dt ** 3, (3, 3))
        # "true" deltas at the true bias of k1, then moved back to the linearisation bias so that the corrected deltas are exact + noise
        dbg = kf_true[k1, 15:18] - b_lin_g
        dba = kf_true[k1, 18:21] - b_lin_a
        dR_true = R1.T @ R2 @ _rot(rng.normal(0, 4e-4, 3))
        dV_true = R1.T @ (V[ti] - V[ti - 1] - g * dt) + rng.normal(0, 2e-3, 3)
        dP_true = R1.T @ (P[ti] - P[ti - 1] - V[ti - 1] * dt - 0.5 * g * dt * dt) + rng.normal(0, 1e-3, 3)
        dR0 = dR_true @ _rot(JRg @ dbg).T
        dV0 = dV_true - JVg @ dbg - JVa @ dba
        dP0 = dP_true - JPg @ dbg - JPa @ dba
        rec = np.concatenate([[dt], dR0.reshape(-1), dV0, dP0, JRg.reshape(-1), JVg.reshape(-1), JVa.reshape(-1), JPg.reshape(-1),
                              JPa.reshape(-1), b_lin_g, b_lin_a])
        rec = rec.astype(np.float32).astype(np.float64)                  # IMU::Preintegrated holds float cv::Mat
        Rq, _ = np.linalg.qr(rng.normal(0, 1, (9, 9)))
        ev = np.concatenate([rng.uniform(2e5, 4e6, 3), rng.uniform(2e4, 4e5, 3), rng.uniform(1e5, 3e6, 3)])
        Cmix = np.eye(9) + 0.15 * (Rq - np.eye(9))
        I9 = Cmix @ np.diag(ev) @ Cmix.T
        I9 = 0.5 * (I9 + I9.T)
        last = a == n_opt - 1                                            # i == N-1: the edge into the fixed keyframe
        if last:
            I9 = I9 * 1e-2
        in1.append(k1); in2.append(k2); pre.append(rec); info.append(I9.reshape(-1))
        infog.append((np.eye(3) * rng.uniform(2e7, 2e8)).reshape(-1)); infoa.append((np.eye(3) * rng.uniform(2e4, 2e5)).reshape(-1))
        rob.append(1 if last else 0)
    # initial estimates: truth + noise on the optimizable states and the points
    kf0 = kf_true.copy()
    for a in range(n_opt):
        R = kf0[a, 0:9].reshape(3, 3) @ _rot(rng.normal(0, 0.004 * state_noise, 3))
        kf0[a, 0:9] = R.reshape(-1)
        kf0[a, 9:12] += rng.normal(0, 0.02 * state_noise, 3)
        kf0[a, 12:15] += rng.normal(0, 0.03 * state_noise, 3)
        kf0[a, 15:18] += rng.normal(0, 5e-4 * state_noise, 3)
        kf0[a, 18:21] += rng.normal(0, 5e-3 * state_noise, 3)
    pts0 = (pts_true + rng.normal(0, 0.04 * state_noise, pts_true.shape)).astype(np.float32).astype(np.float64)
    d = dict(kf_fixed=kf_fixed, kf_imu=kf_imu, kf_state=kf0, points=pts0, cam=(fx, fy, cx, cy, bf), Rcb=Rcb, tcb=tcb,
             edge_kf=ekf, edge_point=ept, edge_obs=np.array(eobs).reshape(-1, 3), edge_stereo=est, edge_inv_sigma2=eis2, edge_close=eclose,
             in_kf1=in1, in_kf2=in2, in_preint=np.array(pre), in_info=np.array(info), in_info_g=np.array(infog), in_info_a=np.array(infoa),
             in_robust=rob, kf_true=kf_true, pts_true=pts_true)
    if fisheye_rig:
        d.update(camera_model=1, kb=kb, Trl=Trl, cam2=cam2, camera2_model=1, kb2=kb2)
    return Window(d)


def load_golden_windows():
    """tests/golden/iba_golden.npz (made by tools/gen_golden.py from the oracle): [(Window, expected dict)]."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "iba_golden.npz"))
    res = []
    for name in ("w0", "w1"):
        pre = name + "_in_"
        d = {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}
        for k in ("camera_model", "camera2_model"):
            if k in d:
                d[k] = int(d[k])
        res.append((Window(d), {"kf": g[name + "_kf"], "pts": g[name + "_pts"], "outlier": g[name + "_outlier"],
                                "stats": g[name + "_stats"], "err": g[name + "_err"]}))
    return res
