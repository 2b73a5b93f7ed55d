"""The keyframe-side ORBmatcher methods, Optimizer::PoseOptimization(Frame*) and the map-merge Optimizer::LocalBundleAdjustment of
orb-slam3-mac_amd/host (reference signatures) run end to end on the GPU over KeyFrame / MapPoint pointer graphs, each checked against
an INDEPENDENT Python model of the host geometry (numpy float32 with OpenCV's rounding points, host/cvmath.h) on top of the matcher /
BA oracles: bit-exact match vectors and bookkeeping, poses <= 1e-9 / 1e-4 as the solvers' own tests."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "orb-slam3-mac_amd", "lib", "host_smoke")
F32 = np.float32


# ------------------------------------------------------------------ named flat arrays (host/flatfile.h)
def write_flat(path, arrays):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(arrays)))
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            if a.dtype.fields is not None or a.dtype == np.uint8:
                kind, raw = 2, a.view(np.uint8).reshape(-1)
            elif a.dtype.kind == "f":
                kind, raw = 1, a.astype(np.float32).reshape(-1)
            else:
                kind, raw = 0, a.astype(np.int32).reshape(-1)
            f.write(name.encode().ljust(24, b"\0")[:24]); f.write(struct.pack("<ii", kind, raw.size)); f.write(raw.tobytes())


def read_flat(path):
    out = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<i", f.read(4))
        for _ in range(n):
            name = f.read(24).split(b"\0")[0].decode()
            kind, cnt = struct.unpack("<ii", f.read(8))
            dt = (np.int32, np.float32, np.uint8)[kind]
            out[name] = np.frombuffer(f.read(cnt * np.dtype(dt).itemsize), dt).copy()
    return out


def run_smoke(mode, tmp_path, arrays, token):
    fin, fout = str(tmp_path / (mode + ".in")), str(tmp_path / (mode + ".out"))
    write_flat(fin, arrays)
    r = subprocess.run([EXE, mode, fin, fout], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=dict(os.environ, ORBHIP_SHIM_DEBUG="1"))
    assert r.returncode == 0 and token in r.stdout, r.stdout[-3000:]
    out = read_flat(fout)
    out["_log"] = r.stdout
    return out


# ------------------------------------------------------------------ host/cvmath.h in numpy (scalar loops: the point sets are small)
def gemm_small(R, x, t=None, alpha=1.0):
    out = np.zeros(3, F32)
    for i in range(3):
        s = F32(F32(F32(R[i][0]) * F32(x[0]) + F32(R[i][1]) * F32(x[1])) + F32(R[i][2]) * F32(x[2]))
        out[i] = F32(np.float64(s) * alpha + (np.float64(t[i]) if t is not None else 0.0))
    return out


def gemm_t(R, x, t=None, alpha=1.0):                     # R^T x (+ t): double sums
    out = np.zeros(3, F32)
    for i in range(3):
        s = 0.0
        for k in range(3):
            s += np.float64(R[k][i]) * np.float64(x[k])
        out[i] = F32(s * alpha + (np.float64(t[i]) if t is not None else 0.0))
    return out


def mat_mul_t(A, tA, B, tB, alpha=1.0):                  # a transposed operand: double sums
    D = np.zeros((3, 3), F32)
    for i in range(3):
        for j in range(3):
            s = 0.0
            for k in range(3):
                s += np.float64(A[k][i] if tA else A[i][k]) * np.float64(B[j][k] if tB else B[k][j])
            D[i, j] = F32(s * alpha)
    return D


def dotd(a, b):
    s = 0.0
    for k in range(3):
        s += np.float64(a[k]) * np.float64(b[k])
    return s


def normd(a):
    return np.sqrt(dotd(a, a))


def project(cam_type, cam, P):
    """GeometricCamera::project(cv::Point3f) of the stand-in / reference: Pinhole.cpp:34-37 (float)"""
    import oracle_match_bind as om
    if cam_type == 0:
        return F32(F32(F32(cam[0]) * P[0]) / P[2] + F32(cam[2])), F32(F32(F32(cam[1]) * P[1]) / P[2] + F32(cam[3]))
    uv = om.camera_project_f(1, cam, P)                  # fisheye scenes go through the deterministic restatement on both sides
    return uv[0], uv[1]


def predict_scale(max_dist, dist, nlevels=8):
    ratio = F32(F32(max_dist) / F32(dist))
    n = int(np.ceil(F32(np.log(ratio)) / F32(np.log(F32(1.2)))))
    return 0 if n < 0 else nlevels - 1 if n >= nlevels else n


def scale_and_angle(W, l, p3Dw, Ow, check_normal):
    maxD = F32(F32(1.2) * W["pt_dist"][l, 1]); minD = F32(F32(0.8) * W["pt_dist"][l, 0])
    PO = (p3Dw - Ow).astype(F32)
    dist = F32(normd(PO))
    if dist < minD or dist > maxD:
        return None
    if check_normal and dotd(PO, W["pt_n"][l]) < 0.5 * np.float64(dist):
        return None
    return predict_scale(W["pt_dist"][l, 1], dist)


# ------------------------------------------------------------------ the world
def make_world(seed, rig=False, fisheye=False, n1=700, n2=800, M=900):
    import oracle_match_bind as om
    from oracle_bind import KP_DTYPE
    rng = np.random.default_rng(seed)
    if fisheye or rig:
        camL = np.array([190.9, 190.8, 254.9, 256.8, 0.0034, 0.0007, -0.0020, 0.0002], F32)
        camR = np.array([190.4, 190.6, 252.7, 255.0, 0.0031, 0.0009, -0.0019, 0.0003], F32)
        bounds = (0, 0, 512, 512); ctype = 1
    else:
        camL = np.array([458.0, 457.0, 320.0, 240.0, 0, 0, 0, 0], F32); camR = camL.copy()
        bounds = (0, 0, 640, 480); ctype = 0
    scale = np.cumprod(np.concatenate([[F32(1.0)], np.full(7, F32(1.2))]).astype(F32)).astype(F32)
    sigma2 = (scale * scale).astype(F32); invsigma2 = (F32(1.0) / sigma2).astype(F32)
    mbf = F32(40.0)

    def pose(ax, ang, t):
        T = np.eye(4, dtype=F32); T[:3, :3] = om._rot(ax, ang).astype(F32); T[:3, 3] = np.asarray(t, F32); return T
    Tcw = [pose([0.1, 1.0, 0.0], 0.05, [0.3, -0.02, 0.1]), pose([0.0, 1.0, 0.1], -0.03, [-0.25, 0.03, 0.05])]
    Rrl = om._rot([0.1, 1.0, 0.05], 0.02); trl = np.array([-0.101, 0.0007, 0.0012])
    Trl = np.zeros((3, 4), F32); Trl[:, :3] = Rrl.astype(F32); Trl[:, 3] = trl.astype(F32)
    Tlr = np.zeros((3, 4), F32); Tlr[:, :3] = Rrl.T.astype(F32); Tlr[:, 3] = (-Rrl.T @ trl).astype(F32)
    # points in front of both cameras, world coordinates
    Xc2 = np.stack([rng.uniform(-2.5, 2.5, M), rng.uniform(-1.8, 1.8, M), rng.uniform(2.5, 9.0, M)], 1)
    R2, t2 = Tcw[1][:3, :3].astype(np.float64), Tcw[1][:3, 3].astype(np.float64)
    X = ((Xc2 - t2) @ R2).astype(F32)
    desc = rng.integers(0, 256, (M, 32), dtype=np.uint8)
    level = rng.integers(0, 7, M)
    W = dict(rig=int(rig), cam_type=ctype, cam1=camL, cam2=camR, bounds=bounds, scale=scale, sigma2=sigma2, invsigma2=invsigma2, mbf=mbf,
             Tlr=Tlr, Trl=Trl, pt_X=X, pt_desc=desc, Tcw=Tcw)
    Ow = [(-(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))) for T in Tcw]
    d2 = np.linalg.norm(X.astype(np.float64) - Ow[1], axis=1)
    maxD = (d2 * (1.2 ** (level + 0.5))).astype(F32)
    W["pt_dist"] = np.stack([(maxD / F32(1.2 ** 7)).astype(F32), maxD], 1).astype(F32)
    nrm = (X.astype(np.float64) - Ow[1]); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)     # mean viewing direction: camera -> point (MapPoint.cc:443-452)
    nrm[rng.random(M) < 0.05] *= -1                                          # some seen from behind: the viewing-angle test
    W["pt_n"] = nrm.astype(F32)
    W["pt_obs"] = rng.choice([1, 2, 3, 5, 8], M).astype(np.int32)
    W["pt_bad"] = (rng.random(M) < 0.04).astype(np.int32)
    node_of_pt = (np.arange(M) % 70) * 3 + 100
    for k, n in ((0, n1), (1, n2)):
        T = Tcw[k]; R, t = T[:3, :3].astype(np.float64), T[:3, 3].astype(np.float64)
        src = rng.integers(0, M, n)
        Xc = X[src].astype(np.float64) @ R.T + t
        nleft = int(n * 0.55) if rig else n
        right = np.arange(n) >= nleft
        Xr = Xc @ Rrl.T + trl
        uvL = om.kb8_project_np((ctype, camL.astype(np.float64)), Xc); uvR = om.kb8_project_np((ctype, camR.astype(np.float64)), Xr)
        uv = np.where(right[:, None], uvR, uvL) + rng.normal(0, 0.6, (n, 2)) * rng.choice([1, 1, 5], n)[:, None]
        kp = np.zeros(n, KP_DTYPE)
        kp["x"], kp["y"] = uv[:, 0], uv[:, 1]
        stray = rng.random(n) < 0.1
        kp["x"][stray] = rng.uniform(bounds[0] + 5, bounds[2] - 5, stray.sum()); kp["y"][stray] = rng.uniform(bounds[1] + 5, bounds[3] - 5, stray.sum())
        dcur = np.linalg.norm(X[src].astype(np.float64) - Ow[k], axis=1)
        pred = np.clip(np.ceil(np.log(maxD[src] / dcur) / np.log(1.2)), 0, 7).astype(int)
        kp["octave"] = np.clip(pred - rng.integers(0, 2, n) + (rng.random(n) < 0.05) * 3, 0, 7)
        kp["angle"] = ((src * 37 % 360) + rng.normal(0, 3, n) + rng.choice([0, 0, 0, 120], n)) % 360
        kp["size"] = 31; kp["class_id"] = -1
        noise = rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8)
        noise[rng.random(n) < 0.4] = 0
        d = desc[src] ^ noise
        d[stray] = rng.integers(0, 256, (int(stray.sum()), 32), dtype=np.uint8)
        ur = np.full(n, -1.0, F32)
        if not rig and not fisheye:
            st = rng.random(n) < 0.3
            ur[st] = (kp["x"][st] - mbf / Xc[st, 2]).astype(F32)
        mp = np.full(n, -1, np.int32)
        hold = (rng.random(n) < 0.4) & ~stray
        seen = set()
        for i in np.flatnonzero(hold):                                       # a map point sits at ONE keypoint of a keyframe
            if src[i] not in seen:
                mp[i] = src[i]; seen.add(int(src[i]))
        nid = np.where(rng.random(n) < 0.9, node_of_pt[src], rng.integers(0, 80, n) * 3 + 101).astype(np.int32)
        nid[rng.random(n) < 0.03] = -1
        W["kf%d" % (k + 1)] = dict(kp=kp, desc=d, ur=ur, mp=mp, nid=nid, nleft=nleft, src=src)
    # duplicates of points keyframe 2 already holds (what Fuse exists for): same place and descriptor, another identity, held by nobody
    held = np.unique(W["kf2"]["mp"][W["kf2"]["mp"] >= 0])
    dup = rng.choice(held, min(150, len(held)), replace=False)
    W["pt_X"] = np.concatenate([W["pt_X"], (W["pt_X"][dup] + rng.normal(0, 0.002, (len(dup), 3))).astype(F32)])
    for k in ("pt_desc", "pt_dist", "pt_n"):
        W[k] = np.concatenate([W[k], W[k][dup]])
    W["pt_obs"] = np.concatenate([W["pt_obs"], rng.choice([1, 2, 3, 5, 8], len(dup)).astype(np.int32)])
    W["pt_bad"] = np.concatenate([W["pt_bad"], np.zeros(len(dup), np.int32)])
    return W, rng


def world_arrays(W):
    a = dict(rig=[W["rig"]], cam_type=[W["cam_type"]], cam1=W["cam1"], cam2=W["cam2"], bounds=np.array(W["bounds"], F32), scale=W["scale"], sigma2=W["sigma2"],
             invsigma2=W["invsigma2"], mbf=[W["mbf"]], Tlr=W["Tlr"], Trl=W["Trl"], pt_X=W["pt_X"], pt_n=W["pt_n"], pt_dist=W["pt_dist"], pt_desc=W["pt_desc"],
             pt_obs=W["pt_obs"], pt_bad=W["pt_bad"])
    for k in (1, 2):
        kf = W["kf%d" % k]
        a.update({"kf%d_Tcw" % k: W["Tcw"][k - 1], "kf%d_kp" % k: kf["kp"], "kf%d_desc" % k: kf["desc"], "kf%d_ur" % k: kf["ur"], "kf%d_mp" % k: kf["mp"],
                  "kf%d_nid" % k: kf["nid"], "kf%d_nleft" % k: [kf["nleft"]]})
    return a


def in_image(W, u, v):
    b = W["bounds"]
    return u >= b[0] and u < b[2] and v >= b[1] and v < b[3]


def kf_pose_parts(T):
    R = T[:3, :3]; t = T[:3, 3]
    Ow = gemm_t(R, t, None, -1.0)
    return R, t, Ow


def right_pose_parts(W, T):
    """KeyFrame::GetRightRotation / GetRightTranslation / GetRightCameraCenter as host/slam_types.h states them (double sums)"""
    Tlr = W["Tlr"]
    R = np.zeros((3, 3), F32)
    for i in range(3):
        for j in range(3):
            R[i, j] = F32(sum(np.float64(Tlr[k, i]) * np.float64(T[k, j]) for k in range(3)))
    trl = np.array([F32(-1.0 * sum(np.float64(Tlr[k, i]) * np.float64(Tlr[k, 3]) for k in range(3))) for i in range(3)], F32)
    t = np.array([F32(sum(np.float64(Tlr[k, i]) * np.float64(T[k, 3]) for k in range(3)) + np.float64(trl[i])) for i in range(3)], F32)
    Ow = kf_pose_parts(T)[2]
    Owr = np.array([F32(sum(np.float64(T[k, i]) * np.float64(Tlr[k, 3]) for k in range(3)) + np.float64(Ow[i])) for i in range(3)], F32)
    return R, t, Owr


def query(u, v, radius, ur, angle, lo, hi, has_obs):
    return (u, v, radius, ur, angle, lo, hi, has_obs)


def as_queries(q, dq):
    import oracle_match_bind as om
    return np.array(q, om.PROJ_QUERY_DTYPE), np.array(dq, np.uint8).reshape(-1, 32)


# ------------------------------------------------------------------ expected results
def expected_fuse(W, cand, th, right):
    import oracle_match_bind as om
    kf = W["kf2"]; T = W["Tcw"][1]
    R, t, Ow = right_pose_parts(W, T) if right else kf_pose_parts(T)
    cam = W["cam2"] if right else W["cam1"]
    in_kf = set(int(x) for x in kf["mp"] if x >= 0)
    q, dq, owner = [], [], []
    for i, l in enumerate(cand):
        if l < 0 or W["pt_bad"][l] or l in in_kf:
            continue
        p3Dw = W["pt_X"][l]
        p3Dc = gemm_small(R, p3Dw, t)
        if p3Dc[2] < 0.0:
            continue
        invz = F32(F32(1) / p3Dc[2])
        u, v = project(W["cam_type"], cam, p3Dc)
        if not in_image(W, u, v):
            continue
        ur = F32(u - F32(W["mbf"] * invz))
        lvl = scale_and_angle(W, l, p3Dw, Ow, True)
        if lvl is None:
            continue
        q.append(query(u, v, F32(F32(th) * W["scale"][lvl]), ur, 0.0, lvl - 1, lvl, 0)); dq.append(W["pt_desc"][l]); owner.append(i)
    q, dq = as_queries(q, dq)
    nleft = kf["nleft"]
    if W["rig"]:
        sl = slice(nleft, None) if right else slice(0, nleft)
        kps, ds = kf["kp"][sl], kf["desc"][sl]
        urs = kf["ur"][:len(kps)]
        row0 = nleft if right else 0
    else:
        kps, ds, urs, row0 = kf["kp"], kf["desc"], kf["ur"], 0
    bi, bd = om.fuse_search(q, dq, kps, ds, urs, W["invsigma2"], W["bounds"])
    kfmp = kf["mp"].copy(); obs = W["pt_obs"].copy(); bad = W["pt_bad"].copy().astype(bool); replaced = np.full(len(obs), -1, np.int32)
    in_kf_now = set(in_kf)
    nfused = 0
    for t_, i in enumerate(owner):
        l = cand[i]
        if bad[l] or l in in_kf_now or bd[t_] > 50:
            continue
        idx = int(bi[t_]) + row0
        other = kfmp[idx]
        if other >= 0:
            if not bad[other]:
                if obs[other] > obs[l]:
                    replaced[l] = other; bad[l] = True
                else:
                    replaced[other] = l; bad[other] = True
        else:
            in_kf_now.add(int(l)); kfmp[idx] = l
            obs[l] += 2 if (not W["rig"] and kf["ur"][idx] >= 0) else 1
        nfused += 1
    return nfused, kfmp, replaced, obs, len(q)


def sim3_parts(Scw):
    sR = Scw[:3, :3]
    scw = F32(np.sqrt(dotd(sR[0], sR[0])))
    f = F32(1.0 / np.float64(scw))
    R = (sR * f).astype(F32); t = (Scw[:3, 3] * f).astype(F32)
    return R, t, gemm_t(R, t, None, -1.0)


def sim3_queries(W, cand, Scw, th, skip, project_camera, has_obs):
    kf = W["kf2"]
    R, t, Ow = sim3_parts(Scw)
    fx, fy, cx, cy = (F32(v) for v in W["cam1"][:4])
    q, dq, owner = [], [], []
    for i, l in enumerate(cand):
        if W["pt_bad"][l] or l in skip:
            continue
        p3Dw = W["pt_X"][l]
        p3Dc = gemm_small(R, p3Dw, t)
        if p3Dc[2] < 0.0:
            continue
        if project_camera:
            u, v = project(W["cam_type"], W["cam1"], p3Dc)
        else:
            invz = F32(F32(1) / p3Dc[2]); x = F32(p3Dc[0] * invz); y = F32(p3Dc[1] * invz)
            u = F32(F32(fx * x) + cx); v = F32(F32(fy * y) + cy)
        if not in_image(W, u, v):
            continue
        lvl = scale_and_angle(W, l, p3Dw, Ow, True)
        if lvl is None:
            continue
        q.append(query(u, v, F32(F32(th) * W["scale"][lvl]), -1.0, 0.0, lvl - 1, lvl, has_obs)); dq.append(W["pt_desc"][l]); owner.append(i)
    return as_queries(q, dq) + (owner,)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n1,n2,npts", [(11, 700, 800, 900), (12, 700, 800, 900), (13, 5200, 5600, 6000)])
def test_keyframe_matchers_pinhole(tmp_path, seed, n1, n2, npts):
    """Pinhole keyframes (monocular + stereo keypoints): every keyframe-side method against the Python model + oracles.
    The third case has keyframes of 5200 / 5600 keypoints -- what the two first keyframes of a monocular map carry (the initialisation
    extractor runs 5 x nFeatures, Tracking.cc:210): round 3's kernels refused them (LDS capacities of 2048 / 2900 / 4096 keypoints per
    keyframe) and the class methods then found nothing; round 4 gives Fuse / SearchBySim3, SearchForTriangulation, SearchByBoW and the
    Sim3 / relocalisation projections a form that keeps only the grid in LDS."""
    import oracle_match_bind as om
    W, rng = make_world(seed, n1=n1, n2=n2, M=npts)
    M = len(W["pt_obs"]); kf1, kf2 = W["kf1"], W["kf2"]
    A = world_arrays(W)
    # candidate lists: map points incl. NULL entries, duplicates of points the keyframe already holds (-> Replace), points it holds (skipped)
    fuse_list = rng.integers(-1, M, 600).astype(np.int32)
    sim3_list = rng.permutation(M)[:500].astype(np.int32)
    Scw = W["Tcw"][1].copy(); Scw[:3, :3] *= F32(1.1); Scw[:3, 3] *= F32(1.1)
    T1, T2 = W["Tcw"][0].astype(np.float64), W["Tcw"][1].astype(np.float64)
    R12 = (T1[:3, :3] @ T2[:3, :3].T); t12 = T1[:3, 3] - R12 @ T2[:3, 3]
    K = np.array([[W["cam1"][0], 0, W["cam1"][2]], [0, W["cam1"][1], W["cam1"][3]], [0, 0, 1]], np.float64)
    Kinv = np.linalg.inv(K)
    F12 = (Kinv.T @ om.skew(t12.astype(F32)).astype(np.float64) @ R12 @ Kinv).astype(F32)
    s3_pre = np.full(len(kf1["kp"]), -1, np.int32)
    held2 = kf2["mp"][kf2["mp"] >= 0]
    s3_pre[rng.choice(len(s3_pre), 40, replace=False)] = rng.choice(held2, 40)
    fr_mp = np.where(rng.random(len(kf2["kp"])) < 0.15, rng.integers(0, M, len(kf2["kp"])), -1).astype(np.int32)
    already = rng.choice(kf1["mp"][kf1["mp"] >= 0], 30, replace=False).astype(np.int32)
    A.update(fuse_list=fuse_list, th_fuse=[3.0], sim3_list=sim3_list, Scw=Scw, s12=[1.0], R12=R12.astype(F32), t12=t12.astype(F32), F12=F12, s3_pre=s3_pre,
             fr_kp=kf2["kp"], fr_desc=kf2["desc"], fr_Tcw=W["Tcw"][1], fr_mp=fr_mp, already=already)
    out = run_smoke("kfmatch", tmp_path, A, "HOST_KF_OK")

    # ---- Fuse(pKF, vpMapPoints, th)
    n, kfmp, repl, obs, nq = expected_fuse(W, fuse_list, 3.0, False)
    assert nq > 100 and n > 40 and (repl >= 0).sum() > 3
    assert out["fuse_n"][0] == n
    np.testing.assert_array_equal(out["fuse_kfmp"], kfmp); np.testing.assert_array_equal(out["fuse_replaced"], repl); np.testing.assert_array_equal(out["fuse_nobs"], obs)

    # ---- Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)
    in2 = set(int(x) for x in kf2["mp"] if x >= 0 and not W["pt_bad"][x])                    # KeyFrame::GetMapPoints: the not-bad ones
    q, dq, owner = sim3_queries(W, sim3_list, Scw, 4, in2, True, 0)
    bi, bd = om.window_best(q, dq, kf2["kp"], kf2["desc"], W["bounds"])
    kfmp = kf2["mp"].copy(); rp = np.full(len(sim3_list), -1, np.int32); nf = 0
    for t_, i in enumerate(owner):
        if bd[t_] > 50:
            continue
        other = kfmp[bi[t_]]
        if other >= 0:
            if not W["pt_bad"][other]:
                rp[i] = other
        else:
            kfmp[bi[t_]] = sim3_list[i]
        nf += 1
    assert nf > 30 and out["fs_n"][0] == nf
    np.testing.assert_array_equal(out["fs_replace"], rp); np.testing.assert_array_equal(out["fs_kfmp"], kfmp)

    # ---- SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming) and the overload with keyframes
    for variant, th, key in ((0, 3, "sp"), (1, 8, "spk")):
        matched0 = np.where(np.arange(len(kf2["mp"])) % 3 == 0, kf2["mp"], -1).astype(np.int32)
        found = set(int(x) for x in matched0 if x >= 0)
        q, dq, owner = sim3_queries(W, sim3_list, Scw, th, found, variant == 0, 1)
        nm, tm = om.search_by_projection_sim3(q, dq, kf2["kp"], kf2["desc"], W["bounds"], np.where(matched0 >= 0, -2, -1), 1.5)
        exp = matched0.copy(); who = np.full(len(exp), -1, np.int32)
        for i in np.flatnonzero(tm >= 0):
            exp[i] = sim3_list[owner[tm[i]]]; who[i] = owner[tm[i]] & 1
        assert nm > 30 and out[key + "_n"][0] == nm
        np.testing.assert_array_equal(out[key + "_matched"], exp)
        if variant:
            np.testing.assert_array_equal(out["spk_kf"], who)

    # ---- SearchBySim3
    sR12 = (R12.astype(F32) * F32(1.0)).astype(F32); sR21 = (R12.astype(F32).T * F32(1.0 / np.float64(F32(1.0)))).astype(F32)
    t12f = t12.astype(F32); t21 = gemm_small(sR21, t12f, None, -1.0)
    fx, fy, cx, cy = (F32(v) for v in W["cam1"][:4])

    def direction(kfa, kfb, Ta, sRba, tba, done):
        Ra, ta = Ta[:3, :3], Ta[:3, 3]
        q, dq, owner = [], [], []
        for i, l in enumerate(kfa["mp"]):
            if l < 0 or done[i] or W["pt_bad"][l]:
                continue
            pA = gemm_small(Ra, W["pt_X"][l], ta); pB = gemm_small(sRba, pA, tba)
            if pB[2] < 0.0:
                continue
            invz = F32(1.0 / np.float64(pB[2])); x = F32(pB[0] * invz); y = F32(pB[1] * invz)
            u = F32(F32(fx * x) + cx); v = F32(F32(fy * y) + cy)
            if not in_image(W, u, v):
                continue
            maxD = F32(F32(1.2) * W["pt_dist"][l, 1]); minD = F32(F32(0.8) * W["pt_dist"][l, 0]); d3 = F32(normd(pB))
            if d3 < minD or d3 > maxD:
                continue
            lvl = predict_scale(W["pt_dist"][l, 1], d3)
            q.append(query(u, v, F32(F32(7.5) * W["scale"][lvl]), -1.0, 0.0, lvl - 1, lvl, 0)); dq.append(W["pt_desc"][l]); owner.append(i)
        q, dq = as_queries(q, dq)
        bi, bd = om.window_best(q, dq, kfb["kp"], kfb["desc"], W["bounds"])
        m = np.full(len(kfa["mp"]), -1, np.int32)
        for t_, i in enumerate(owner):
            if bd[t_] <= 100:
                m[i] = bi[t_]
        return m, len(q)
    done1 = s3_pre >= 0
    done2 = np.zeros(len(kf2["mp"]), bool)
    idx_in_kf2 = {int(l): i for i, l in enumerate(kf2["mp"]) if l >= 0}
    for l in s3_pre[s3_pre >= 0]:
        if int(l) in idx_in_kf2:
            done2[idx_in_kf2[int(l)]] = True
    m1, nq1 = direction(kf1, kf2, W["Tcw"][0], sR21, t21, done1)
    m2, nq2 = direction(kf2, kf1, W["Tcw"][1], sR12, t12f, done2)
    exp = s3_pre.copy(); nfound = 0
    for i1 in range(len(m1)):
        if m1[i1] >= 0 and m2[m1[i1]] == i1:
            exp[i1] = kf2["mp"][m1[i1]]; nfound += 1
    assert nq1 > 50 and nq2 > 50 and nfound > 5
    assert out["s3_n"][0] == nfound
    np.testing.assert_array_equal(out["s3_matches"], exp)

    # ---- SearchByBoW(pKF1, pKF2)
    c = dict(nid_k=np.where(kf1["nid"] >= 0, kf1["nid"], 10 ** 6 + np.arange(len(kf1["nid"]))), kp_k=kf1["kp"], d_k=kf1["desc"],
             valid=((kf1["mp"] >= 0) & (W["pt_bad"][np.clip(kf1["mp"], 0, None)] == 0)).astype(np.uint8),
             nid_f=np.where(kf2["nid"] >= 0, kf2["nid"], 2 * 10 ** 6 + np.arange(len(kf2["nid"]))), kp_f=kf2["kp"], d_f=kf2["desc"],
             valid2=((kf2["mp"] >= 0) & (W["pt_bad"][np.clip(kf2["mp"], 0, None)] == 0)).astype(np.uint8))
    nm, m12 = om.search_by_bow_kf(c, 0.9, True)
    exp = np.where(m12 >= 0, kf2["mp"][np.clip(m12, 0, None)], -1)
    assert nm > 20 and out["bow_n"][0] == nm
    np.testing.assert_array_equal(out["bow_matches"], exp)

    # ---- SearchForTriangulation
    R1w, t1w, Cw = kf_pose_parts(W["Tcw"][0]); R2w, t2w, _ = kf_pose_parts(W["Tcw"][1])
    C2 = gemm_small(R2w, Cw, t2w)
    ep = project(0, W["cam1"], C2)
    g = np.zeros(1, om.TRI_GENERAL_DTYPE)[0]
    g["R12"][0] = mat_mul_t(R1w, False, R2w, True).reshape(9); g["t12"][0] = gemm_small(mat_mul_t(R1w, False, R2w, True, -1.0), t2w, t1w)
    g["F12"][0] = F12.reshape(9); g["cam1"][0] = W["cam1"]; g["cam2"][0] = W["cam1"]; g["ep_x"], g["ep_y"] = ep; g["nleft1"] = g["nleft2"] = -1
    base = dict(kp1=kf1["kp"], d1=kf1["desc"], nid1=kf1["nid"], mp1=(kf1["mp"] >= 0).astype(np.uint8), ur1=kf1["ur"], kp2=kf2["kp"], d2=kf2["desc"],
                nid2=np.where(kf2["nid"] >= 0, kf2["nid"], 2 * 10 ** 6 + np.arange(len(kf2["nid"]))), mp2=(kf2["mp"] >= 0).astype(np.uint8), ur2=kf2["ur"],
                scale=W["scale"], sigma2=W["sigma2"], sigma2_1=W["sigma2"])
    for key, coarse, ori in (("tri", 0, False), ("tric", 1, False), ("trio", 0, True)):
        gg = g.copy(); gg["coarse"] = coarse
        nm, m = om.search_for_triangulation_general(dict(base, geom=gg), ori)
        pairs = np.stack([np.flatnonzero(m >= 0), m[m >= 0]], 1).reshape(-1).astype(np.int32)
        assert nm > 20 and out[key + "_n"][0] == nm, (key, nm, out[key + "_n"])
        np.testing.assert_array_equal(out[key + "_pairs"], pairs)
    # the overload that returns the points: Pinhole::matchAndtriangulate is { return false; } (include/CameraModels/Pinhole.h:91-94)
    for key in ("trp", "trpo"):
        assert out[key + "_n"][0] == 0 and len(out[key + "_pairs"]) == 0 and len(out[key + "_points"]) == 0

    # ---- SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist)
    Rc, tc, Owc = kf_pose_parts(W["Tcw"][1])
    q, dq, owner = [], [], []
    alr = set(int(x) for x in already)
    for i, l in enumerate(kf1["mp"]):
        if l < 0 or W["pt_bad"][l] or int(l) in alr:
            continue
        xw = W["pt_X"][l]; xc = gemm_small(Rc, xw, tc)
        u, v = project(0, W["cam1"], xc)
        b = W["bounds"]
        if u < b[0] or u > b[2] or v < b[1] or v > b[3]:
            continue
        lvl = scale_and_angle(W, l, xw, Owc, False)
        if lvl is None:
            continue
        q.append(query(u, v, F32(F32(10) * W["scale"][lvl]), -1.0, kf1["kp"]["angle"][i], lvl - 1, lvl + 1, 1)); dq.append(W["pt_desc"][l]); owner.append(i)
    q, dq = as_queries(q, dq)
    nm, tm = om.search_by_projection(q, dq, kf2["kp"], kf2["desc"], None, W["bounds"], np.where(fr_mp >= 0, -2, -1), 100, True)
    exp = fr_mp.copy()
    for i in np.flatnonzero(tm >= 0):
        exp[i] = kf1["mp"][owner[tm[i]]]
    assert len(q) > 100 and nm > 20 and out["rl_n"][0] == nm
    np.testing.assert_array_equal(out["rl_mp"], exp)


@pytest.mark.gpu
def test_keyframe_matchers_fisheye_rig(tmp_path):
    """Keyframes of a two-camera KannalaBrandt8 rig (NLeft != -1, BASELINE config #5): Fuse in both cameras (bRight), SearchForTriangulation over
    the four camera pairs with the triangulation constraint, SearchByBoW(KF, KF) restricted to the left camera's features."""
    import oracle_match_bind as om
    W, rng = make_world(21, rig=True)
    M = len(W["pt_obs"]); kf1, kf2 = W["kf1"], W["kf2"]
    A = world_arrays(W)
    fuse_list = rng.integers(-1, M, 700).astype(np.int32)
    Scw = W["Tcw"][1].copy()
    A.update(fuse_list=fuse_list, th_fuse=[3.0], sim3_list=rng.permutation(M)[:50].astype(np.int32), Scw=Scw, s12=[1.0], R12=np.eye(3, dtype=F32), t12=np.zeros(3, F32),
             F12=np.zeros((3, 3), F32), s3_pre=np.full(len(kf1["kp"]), -1, np.int32))
    out = run_smoke("kfmatch", tmp_path, A, "HOST_KF_OK")
    for right, key in ((False, "fuse"), (True, "fuser")):
        n, kfmp, repl, obs, nq = expected_fuse(W, fuse_list, 3.0, right)
        assert nq > 60 and n > 15, (key, nq, n)
        assert out[key + "_n"][0] == n
        np.testing.assert_array_equal(out[key + "_kfmp"], kfmp); np.testing.assert_array_equal(out[key + "_replaced"], repl)
        np.testing.assert_array_equal(out[key + "_nobs"], obs)
    # SearchByBoW(KF, KF): only indices below mvKeysUn.size() == NLeft take part (ORBmatcher.cc:862-864, 882-884)
    v1 = ((kf1["mp"] >= 0) & (W["pt_bad"][np.clip(kf1["mp"], 0, None)] == 0) & (np.arange(len(kf1["mp"])) < kf1["nleft"])).astype(np.uint8)
    v2 = ((kf2["mp"] >= 0) & (W["pt_bad"][np.clip(kf2["mp"], 0, None)] == 0) & (np.arange(len(kf2["mp"])) < kf2["nleft"])).astype(np.uint8)
    c = dict(nid_k=np.where(kf1["nid"] >= 0, kf1["nid"], 10 ** 6 + np.arange(len(kf1["nid"]))), kp_k=kf1["kp"], d_k=kf1["desc"], valid=v1,
             nid_f=np.where(kf2["nid"] >= 0, kf2["nid"], 2 * 10 ** 6 + np.arange(len(kf2["nid"]))), kp_f=kf2["kp"], d_f=kf2["desc"], valid2=v2)
    nm, m12 = om.search_by_bow_kf(c, 0.9, True)
    assert nm > 5 and out["bow_n"][0] == nm
    np.testing.assert_array_equal(out["bow_matches"], np.where(m12 >= 0, kf2["mp"][np.clip(m12, 0, None)], -1))
    # SearchForTriangulation: relative poses of the four camera pairs as the method builds them from the keyframe poses
    Rl = [kf_pose_parts(W["Tcw"][k])[0] for k in range(2)]; tl = [kf_pose_parts(W["Tcw"][k])[1] for k in range(2)]
    Rr = [right_pose_parts(W, W["Tcw"][k])[0] for k in range(2)]; tr = [right_pose_parts(W, W["Tcw"][k])[1] for k in range(2)]
    g = np.zeros(1, om.TRI_GENERAL_DTYPE)[0]

    def rel(Ra, ta, Rb, tb):
        return mat_mul_t(Ra, False, Rb, True).reshape(9), gemm_small(Ra, gemm_t(Rb, tb, None, -1.0), ta)
    for c_, (a, b_) in enumerate((("l", "l"), ("l", "r"), ("r", "l"), ("r", "r"))):
        Ra, ta = (Rl[0], tl[0]) if a == "l" else (Rr[0], tr[0]); Rb, tb = (Rl[1], tl[1]) if b_ == "l" else (Rr[1], tr[1])
        g["R12"][c_], g["t12"][c_] = rel(Ra, ta, Rb, tb)
    g["cam1"][0] = W["cam1"]; g["cam1"][1] = W["cam2"]; g["cam2"][0] = W["cam1"]; g["cam2"][1] = W["cam2"]; g["cam1_type"][:] = 1; g["cam2_type"][:] = 1
    Cw = kf_pose_parts(W["Tcw"][0])[2]; C2 = gemm_small(Rl[1], Cw, tl[1])
    # (the epipole goes through the stand-in camera's libm on the C++ side; rig pairs never test it, ORBmatcher.cc:1091)
    g["nleft1"], g["nleft2"] = kf1["nleft"], kf2["nleft"]
    base = dict(kp1=kf1["kp"], d1=kf1["desc"], nid1=kf1["nid"], mp1=(kf1["mp"] >= 0).astype(np.uint8), ur1=kf1["ur"], kp2=kf2["kp"], d2=kf2["desc"],
                nid2=np.where(kf2["nid"] >= 0, kf2["nid"], 2 * 10 ** 6 + np.arange(len(kf2["nid"]))), mp2=(kf2["mp"] >= 0).astype(np.uint8), ur2=kf2["ur"],
                scale=W["scale"], sigma2=W["sigma2"], sigma2_1=W["sigma2"])
    combos = set()
    for key, coarse, ori in (("tri", 0, False), ("tric", 1, False), ("trio", 0, True)):
        gg = g.copy(); gg["coarse"] = coarse
        nm, m = om.search_for_triangulation_general(dict(base, geom=gg), ori)
        pairs = np.stack([np.flatnonzero(m >= 0), m[m >= 0]], 1).reshape(-1).astype(np.int32)
        assert nm > 15 and out[key + "_n"][0] == nm, (key, nm, out[key + "_n"])
        np.testing.assert_array_equal(out[key + "_pairs"], pairs)
        i1 = np.flatnonzero(m >= 0)
        combos |= set((2 * (i1 >= kf1["nleft"]) + (m[i1] >= kf2["nleft"])).tolist())
    assert combos == {0, 1, 2, 3}
    # SearchForTriangulation(..., vMatchedPoints) (ORBmatcher.cc:1212-1402): absolute poses GetPose() / GetRightPose() per camera, bOnlyStereo (passed as
    # true by the smoke) not read; pairs AND world points bit-exact
    P = np.zeros(1, om.TRI_POSES_DTYPE)[0]
    for k, name in enumerate(("Tcw1", "Tcw2")):
        P[name][0] = W["Tcw"][k][:3, :].reshape(12)
        P[name][1] = np.concatenate([Rr[k], tr[k].reshape(3, 1)], 1).reshape(12)
    combos = set()
    for key, ori in (("trp", False), ("trpo", True)):
        nm, m, pts = om.search_for_triangulation_points(dict(base, geom=g), P, ori)
        pairs = np.stack([np.flatnonzero(m >= 0), m[m >= 0]], 1).reshape(-1).astype(np.int32)
        assert nm > 15 and out[key + "_n"][0] == nm, (key, nm, out[key + "_n"])
        np.testing.assert_array_equal(out[key + "_pairs"], pairs)
        np.testing.assert_array_equal(np.asarray(out[key + "_points"], F32).view(np.uint32), pts[m >= 0].reshape(-1).view(np.uint32))
        i1 = np.flatnonzero(m >= 0)
        combos |= set((2 * (i1 >= kf1["nleft"]) + (m[i1] >= kf2["nleft"])).tolist())
        # the points are the world's: each lies within a few centimetres of a map point position or at least in front of both left cameras
        for k in range(2):
            Xc = pts[m >= 0].astype(np.float64) @ W["Tcw"][k][:3, :3].astype(np.float64).T + W["Tcw"][k][:3, 3].astype(np.float64)
            assert (Xc[:, 2] > 0).all()
    assert combos == {0, 1, 2, 3}


# ------------------------------------------------------------------ Optimizer::PoseOptimization(Frame*)
def _quat_pose_from_T(T):
    """Converter::toSE3Quat as host/optimizer_common.h: float 4x4 -> (qx,qy,qz,qw,tx,ty,tz) double, w >= 0, unit norm"""
    R = T[:3, :3].astype(np.float64)
    tr = np.trace(R)
    q = np.zeros(4)
    if tr > 0:
        s = np.sqrt(tr + 1.0); q[3] = 0.5 * s; s = 0.5 / s
        q[0] = (R[2, 1] - R[1, 2]) * s; q[1] = (R[0, 2] - R[2, 0]) * s; q[2] = (R[1, 0] - R[0, 1]) * s
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3; k = (j + 1) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0); q[i] = 0.5 * s; s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s; q[j] = (R[j, i] + R[i, j]) * s; q[k] = (R[k, i] + R[i, k]) * s
    if q[3] < 0:
        q = -q
    q /= np.linalg.norm(q)
    return np.concatenate([q, T[:3, 3].astype(np.float64)])


def _T_from_quat_pose(p):
    import synth_ba
    T = np.eye(4, dtype=F32)
    x, y, z, w = p[:4]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w; txx, txy, txz = tx * x, ty * x, tz * x; tyy, tyz, tzz = ty * y, tz * y, tz * z
    R = np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]])
    T[:3, :3] = R.astype(F32); T[:3, 3] = p[4:].astype(F32)
    return T


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["mono_stereo", "fisheye_rig", "two_points"])
def test_pose_optimization_drop_in(tmp_path, variant):
    """Optimizer::PoseOptimization(Frame*) over a Frame with map points at some keypoints: mvbOutlier, the pose and the return value against the
    oracle run on the edges the reference would create (Optimizer.cc:897-1037)."""
    import oracle_ba_bind as ob
    import synth_ba
    from oracle_bind import KP_DTYPE
    rng = np.random.default_rng(5 if variant != "fisheye_rig" else 6)
    kb8 = (-0.0034, 0.0007, -0.0021, 0.0002) if variant == "fisheye_rig" else None
    rig2 = dict(Trl=(0.004, -0.012, 0.002, 0.99991, -0.101, 0.0007, 0.0012), cam=(190.4, 190.6, 252.7, 255.0), kb=(0.0031, 0.0007, -0.0019, 0.0003)) if kb8 else None
    n = 900 if variant != "two_points" else 40
    pr = synth_ba.make_pose_problem(77, n=n, stereo_frac=0.0 if kb8 else 0.4, outlier_frac=0.12, kb8=kb8, rig2=rig2, right_frac=0.45 if kb8 else 0.0)
    cam = pr["cam"]
    # float32 boundary as the reference's containers: keypoints, mvuRight, world positions, pose
    obs = pr["obs"].astype(F32); Xw = pr["Xw"].astype(F32)
    inv_tab = (F32(1.0) / (np.cumprod(np.concatenate([[F32(1.0)], np.full(7, F32(1.2))]).astype(F32)) ** 2)).astype(F32)
    octave = np.round(-0.5 * np.log(pr["inv_sigma2"]) / np.log(1.2)).astype(np.int64)
    right = pr.get("right")
    right = np.zeros(n, np.uint8) if right is None else np.asarray(right, np.uint8)
    order = np.argsort(right, kind="stable")                      # frame layout of a rig: left keypoints first, then right
    obs, Xw, octave, right = obs[order], Xw[order], octave[order], right[order]
    has = rng.random(n) < (0.85 if variant != "two_points" else 0.05)
    if variant == "two_points":
        has[:] = False; has[[3, 17]] = True
    kp = np.zeros(n, KP_DTYPE); kp["x"], kp["y"], kp["octave"] = obs[:, 0], obs[:, 1], octave
    T0 = _T_from_quat_pose(pr["pose0"])
    Xin = Xw.copy(); Xin[~has] = np.nan
    cam8 = np.zeros(8, F32); cam8[:4] = cam[:4]
    if kb8:
        cam8[4:] = kb8
    cam2 = np.zeros(8, F32)
    A = dict(cam=cam8, cam_type=[1 if kb8 else 0], rig=[1 if kb8 else 0], cam2=cam2, Tcw=T0, kp=kp, ur=np.where(obs[:, 2] >= 0, obs[:, 2], -1).astype(F32), X=Xin,
             invsigma2=inv_tab, mbf=[cam[4]], nleft=[int((right == 0).sum())], Trl=np.zeros((3, 4), F32))
    if kb8:
        cam2[:4] = rig2["cam"]; cam2[4:] = rig2["kb"]
        q = np.array(rig2["Trl"][:4]); R = synth_ba._R_from_quat(q)
        A["Trl"] = np.concatenate([R.astype(F32), np.array(rig2["Trl"][4:], F32)[:, None]], 1)
    out = run_smoke("poseopt", tmp_path, A, "HOST_POSEOPT_OK")
    sel = np.flatnonzero(has)
    if len(sel) < 3:
        assert out["n_inliers"][0] == 0 and np.array_equal(out["Tcw"].reshape(4, 4), T0)
        assert not out["outlier"][sel].any()                      # reset even though nothing is optimised (Optimizer.cc:906, :1040)
        return
    pose0 = _quat_pose_from_T(T0)
    rig_o = None
    if kb8:
        rig_o = dict(rig2); rig_o["Trl"] = tuple(_quat_pose_from_T(np.vstack([A["Trl"], [0, 0, 0, 1]]).astype(F32)))
        rig_o["cam"] = tuple(float(v) for v in cam2[:4]); rig_o["kb"] = tuple(float(v) for v in cam2[4:])
    o_obs = np.stack([obs[sel, 0], obs[sel, 1], A["ur"][sel]], 1).astype(np.float64)
    n_in, pose, outl, st = ob.pose_optimization(Xw[sel].astype(np.float64), o_obs, inv_tab[octave[sel]].astype(np.float64),
                                                tuple(float(v) for v in cam8[:4]) + (float(F32(cam[4])),), pose0,
                                                kb8=tuple(float(v) for v in cam8[4:]) if kb8 else None, rig2=rig_o, right=right[sel] if kb8 else None)
    assert out["n_inliers"][0] == n_in and 50 < n_in < len(sel)
    np.testing.assert_array_equal(out["outlier"][sel], outl)
    assert out["outlier"][~has].all()                             # keypoints without a map point keep their (stale) flag: untouched by the reference too
    Texp = _T_from_quat_pose(pose)
    assert np.abs(out["Tcw"].reshape(4, 4) - Texp).max() <= 1e-6, np.abs(out["Tcw"].reshape(4, 4) - Texp).max()


# ------------------------------------------------------------------ map-merge Optimizer::LocalBundleAdjustment
@pytest.mark.gpu
@pytest.mark.parametrize("marked", [True, False])
def test_merge_local_bundle_adjustment_drop_in(tmp_path, marked):
    """Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, pbStopFlag) over a welding window.  marked: the caller set
    mnBALocalForMerge on the adjustable keyframes (edges to all keyframes); otherwise -- the reference as written -- only observations in FIXED
    keyframes become edges, adjustable poses stay put, points move."""
    import oracle_ba_bind as ob
    import synth_ba
    import test_gpu_host_cpp as hc
    n_kf, n_pts = 16, 400
    g = synth_ba.make_graph(n_kf=n_kf, n_pts=n_pts, obs=10, seed=404, stereo_frac=0.3, n_fixed=0, pose_noise=(0.0004, 0.002), outlier_frac=0.03)
    ids = (np.arange(n_kf) * 3 + 2).astype(np.int32)
    T = np.zeros((n_kf, 4, 4), F32)
    for i in range(n_kf):
        T[i, :3, :3] = synth_ba._R_from_quat(g["poses0"][i, :4]).astype(F32); T[i, :3, 3] = g["poses0"][i, 4:].astype(F32); T[i, 3, 3] = 1
    fixed = (np.arange(n_kf) % 4 != 3).astype(np.int32)             # 12 fixed keyframes: every point keeps edges also when only they carry them
    octave = np.round(-0.5 * np.log(g["edge_inv_sigma2"]) / np.log(1.2)).astype(np.int32)
    inv_s2 = (F32(1.0) / (np.cumprod(np.concatenate([[F32(1.0)], np.full(7, F32(1.2))]).astype(F32)) ** 2)).astype(F32)
    obs3 = g["edge_obs"].astype(F32).copy()
    g["edge_stereo"] = ((g["edge_stereo"] == 1) & (obs3[:, 2] >= 0)).astype(np.uint8)     # a right coordinate left of the image is no stereo keypoint (mvuRight < 0)
    obs3[g["edge_stereo"] == 0, 2] = -1.0
    X = g["points0"].astype(F32)
    A = dict(ids=ids, Tcw=T, fixed=fixed, marked=(1 - fixed) * int(marked), X=X, eKF=g["edge_pose"], eMP=g["edge_point"], eObs=obs3, eOct=octave, invsigma2=inv_s2,
             cam=np.array([g["fx"], g["fy"], g["cx"], g["cy"], g["bf"]], F32), abort=[0], main=[n_kf - 1])
    out = run_smoke("mergeba", tmp_path, A, "HOST_MERGEBA_OK")
    # expected: the oracle under the merge parameters on the edges the filter of Optimizer.cc:6424 lets through
    keep = np.ones(g["n_edges"], bool) if marked else fixed[g["edge_pose"]].astype(bool)
    pts_with_edges = np.unique(g["edge_point"][keep])
    remap = np.full(n_pts, -1, np.int64); remap[pts_with_edges] = np.arange(len(pts_with_edges))
    order = np.lexsort((np.arange(g["n_edges"])[keep], remap[g["edge_point"][keep]]))          # point-major, observation order inside a point
    ek = np.flatnonzero(keep)[order]
    # the shim walks std::map<KeyFrame*, ...>: pointer order of the keyframes == creation order here (ascending index), which is the input order per point
    poses0 = np.stack([_quat_pose_from_T(T[i]) for i in range(n_kf)])
    sub = dict(n_poses=n_kf, n_points=len(pts_with_edges), n_edges=len(ek), pose_fixed=fixed.astype(np.uint8), edge_pose=g["edge_pose"][ek].astype(np.int32),
               edge_point=remap[g["edge_point"][ek]].astype(np.int32), edge_obs=np.concatenate([obs3[ek, :2], np.where(g["edge_stereo"][ek, None] == 1, obs3[ek, 2:3], 0)], 1).astype(np.float64),
               edge_inv_sigma2=inv_s2[octave[ek]].astype(np.float64), edge_stereo=g["edge_stereo"][ek].astype(np.uint8),
               fx=float(F32(g["fx"])), fy=float(F32(g["fy"])), cx=float(F32(g["cx"])), cy=float(F32(g["cy"])), bf=float(F32(g["bf"])),
               poses0=poses0, points0=X[pts_with_edges].astype(np.float64))
    prm = ob.merge_params(); prm.iters2 = 0                          # premise of the merge parity (test_gpu_ba.py::test_ba_merge_variant)
    first = ob.solve(sub, prm)[3]
    assert np.bincount(sub["edge_point"][first == 0], minlength=sub["n_points"]).min() >= 2
    rc, o_poses, o_pts, o_out, o_st = ob.solve(sub, ob.merge_params())
    assert rc == 0 and o_st["discarded"] == 0
    Texp = T.copy()
    for i in np.flatnonzero(fixed == 0):
        Texp[i] = _T_from_quat_pose(o_poses[i])
    er = out["erased"].reshape(-1, 2)
    exp_er = np.stack([g["edge_pose"][ek][o_out == 1], g["edge_point"][ek][o_out == 1]], 1)
    assert len(exp_er) > 5
    assert sorted(map(tuple, er.tolist())) == sorted(map(tuple, exp_er.tolist()))
    # a point that lost observations down to <= 2 goes bad in EraseObservation and is skipped by the write-back (:6892-6893)
    nobs = np.bincount(g["edge_point"], weights=1 + (g["edge_stereo"] == 1), minlength=n_pts)
    lost = np.bincount(exp_er[:, 1], weights=1 + (g["edge_stereo"][ek][o_out == 1] == 1), minlength=n_pts)
    bad = (lost > 0) & (nobs - lost <= 2)
    np.testing.assert_array_equal(out["bad"], bad.astype(np.int32))
    Xexp = X.copy(); Xexp[pts_with_edges] = o_pts.astype(F32); Xexp[bad] = X[bad]
    assert np.abs(out["Tcw"].reshape(n_kf, 4, 4) - Texp).max() <= 2e-5, np.abs(out["Tcw"].reshape(n_kf, 4, 4) - Texp).max()
    dX = np.abs(out["X"].reshape(n_pts, 3) - Xexp).max(1)
    worst = np.argsort(-dX)[:5]
    kept = np.bincount(sub["edge_point"][first == 0], minlength=sub["n_points"])
    diag = [(int(w), float(dX[w]), int(kept[remap[w]]) if remap[w] >= 0 else -1, int(np.bincount(sub["edge_point"], minlength=sub["n_points"])[remap[w]]) if remap[w] >= 0 else -1,
             float(np.abs(Xexp[w] - X[w]).max())) for w in worst]
    assert dX.max() <= 1e-4, (diag, out["_log"])
    assert out["updates"][0] == n_pts - bad.sum()                  # UpdateNormalAndDepth on every (good) map point of the window (:6889-6899)
    if not marked:                                                 # adjustable keyframes have no edges: the pose only takes the Converter round trip
        for i in np.flatnonzero(fixed == 0):
            assert np.abs(out["Tcw"].reshape(n_kf, 4, 4)[i] - T[i]).max() < 1e-6
        assert np.abs(out["X"].reshape(n_pts, 3) - X).max() > 1e-5


# ------------------------------------------------------------------ Frame::ComputeStereoMatches()
@pytest.mark.gpu
@pytest.mark.parametrize("w,h,disp,nfeat,mb,mbf,seed", [(640, 480, 12, 1000, 40.0 / 458.0, 40.0, 72), (640, 480, 3, 1000, 40.0 / 458.0, 40.0, 63),
                                                        (752, 480, 8, 1500, 4.79, 47.9, 88), (640, 480, -1, 1000, 40.0 / 458.0, 40.0, 60)])
def test_frame_compute_stereo_matches_drop_in(tmp_path, w, h, disp, nfeat, mb, mbf, seed):
    """Frame::ComputeStereoMatches() (src/Frame.cc:802-980) through the class, in the rectified-stereo constructor's sequence (:109-130):
    two ORBextractor objects run on two threads, then the member function fills mvuRight / mvDepth from the two device-resident
    extractions -- bit-exact against the oracle's extraction + association of the same images; calling it again changes nothing; a frame
    whose extractor has moved on is refused loudly (everything -1) instead of being matched against another image.  disp -1: a
    featureless right image (no candidates)."""
    import oracle_bind as ob
    from test_oracle_orb import make_stereo_pair
    left, right = make_stereo_pair(w, h, max(disp, 0), seed=seed)
    if disp < 0:
        right = np.full((h, w), 128, np.uint8)
    out = run_smoke("stereo", tmp_path, dict(dims=np.array([w, h, nfeat], np.int32), left=left.reshape(-1), right=right.reshape(-1),
                                             cal=np.array([mb, mbf], np.float32)), "HOST_STEREO_OK")
    eL = ob.OracleExtractor(nfeat, 1.2, 8, 20, 7); eR = ob.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    kpL, dL, _ = eL.extract(left, (0, 0)); kpR, dR, _ = eR.extract(right, (0, 0))
    assert out["n"][0] == len(kpL) and out["nr"][0] == len(kpR)
    assert out["kpx"].tobytes() == kpL["x"].tobytes() and out["kpy"].tobytes() == kpL["y"].tobytes()
    np.testing.assert_array_equal(out["kpo"], kpL["octave"])
    if len(kpR):
        kept, ur_ref, dp_ref, _ = ob.compute_stereo_matches(eL, eR, kpL, dL, kpR, dR, np.float32(mb), np.float32(mbf))
    else:
        kept, ur_ref, dp_ref = 0, np.full(len(kpL), -1, np.float32), np.full(len(kpL), -1, np.float32)
    assert out["uright"].tobytes() == ur_ref.tobytes()
    assert out["depth"].tobytes() == dp_ref.tobytes()
    assert (out["uright"] >= 0).sum() == kept
    if disp >= 0:
        assert kept > (50 if mb > 1 else 300)
    else:
        assert kept == 0
    assert out["same"][0] == 1
    assert out["stale"][0] == 1, out["_log"][-800:]
    assert "not the latest extractions" in out["_log"]
