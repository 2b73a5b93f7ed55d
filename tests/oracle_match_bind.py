"""ctypes binding of the matching ORACLE (oracle/match_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import numpy as np
from oracle_bind import lib, KP_DTYPE

vp, ci, cf, cd = C.c_void_p, C.c_int, C.c_float, C.c_double
lib.orc_descriptor_distance.argtypes = [vp, vp]
lib.orc_bf2nn.argtypes = [vp, ci, vp, ci, cd, vp, vp, vp]
lib.orc_search_for_initialization.argtypes = [vp, vp, ci, vp, vp, ci, cf, cf, cf, cf, ci, cf, ci, vp, vp]
lib.orc_features_in_area.argtypes = [vp, ci, cf, cf, cf, cf, cf, cf, cf, ci, ci, vp, ci]
lib.orc_search_by_projection.argtypes = [vp, vp, ci, vp, vp, vp, ci, cf, cf, cf, cf, ci, ci, vp]
lib.orc_search_by_projection.restype = ci
lib.orc_search_by_projection_map.argtypes = [vp, vp, ci, vp, vp, vp, ci, cf, cf, cf, cf, ci, cf, vp]
lib.orc_search_by_projection_map.restype = ci
lib.orc_fuse_search.argtypes = [vp, vp, ci, vp, vp, vp, ci, vp, cf, cf, cf, cf, vp, vp]
lib.orc_fuse_search.restype = None
lib.orc_search_by_bow.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, ci, cf, ci, vp]
lib.orc_search_by_bow.restype = ci
lib.orc_distinctive_descriptor.argtypes = [vp, ci]
lib.orc_distinctive_descriptor.restype = ci
lib.orc_bow_transform.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp]
lib.orc_bow_transform.restype = None

PROJ_QUERY_DTYPE = np.dtype([("u", np.float32), ("v", np.float32), ("radius", np.float32), ("ur", np.float32),
                             ("angle", np.float32), ("min_level", np.int32), ("max_level", np.int32), ("has_obs", np.int32)])


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib.orc_descriptor_distance(a.ctypes.data, b.ctypes.data)


def bf2nn(A, B, ratio=0.7):
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    na, nb = len(A), len(B)
    idx = np.zeros((max(na, 1), 2), np.int32)
    dist = np.zeros((max(na, 1), 2), np.int32)
    acc = np.zeros(max(na, 1), np.uint8)
    lib.orc_bf2nn(A.ctypes.data, na, B.ctypes.data, nb, ratio, idx.ctypes.data, dist.ctypes.data, acc.ctypes.data)
    return idx[:na], dist[:na], acc[:na]


def search_for_initialization(kpA, dA, kpB, dB, bounds, prev, window=100, ratio=0.9, check_ori=True):
    kpA = np.ascontiguousarray(kpA, KP_DTYPE)
    kpB = np.ascontiguousarray(kpB, KP_DTYPE)
    dA = np.ascontiguousarray(dA, np.uint8)
    dB = np.ascontiguousarray(dB, np.uint8)
    prev = np.ascontiguousarray(prev, np.float32).copy()
    m12 = np.zeros(max(len(kpA), 1), np.int32)
    n = lib.orc_search_for_initialization(kpA.ctypes.data, dA.ctypes.data, len(kpA), kpB.ctypes.data, dB.ctypes.data,
                                          len(kpB), bounds[0], bounds[1], bounds[2], bounds[3], window, ratio,
                                          1 if check_ori else 0, prev.ctypes.data, m12.ctypes.data)
    return n, m12[:len(kpA)], prev


def features_in_area(kp, bounds, x, y, r, min_level, max_level):
    kp = np.ascontiguousarray(kp, KP_DTYPE)
    out = np.zeros(max(len(kp), 1), np.int32)
    n = lib.orc_features_in_area(kp.ctypes.data, len(kp), bounds[0], bounds[1], bounds[2], bounds[3], x, y, r,
                                 min_level, max_level, out.ctypes.data, len(out))
    return out[:n]


def search_by_projection(q, dq, kp, d, u_right, bounds, train_match, th_high=100, check_ori=True):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, ...) restated; returns (nmatches, train_match)."""
    q = np.ascontiguousarray(q, PROJ_QUERY_DTYPE)
    dq = np.ascontiguousarray(dq, np.uint8)
    kp = np.ascontiguousarray(kp, KP_DTYPE)
    d = np.ascontiguousarray(d, np.uint8)
    tm = np.ascontiguousarray(train_match, np.int32).copy()
    if len(tm) == 0:
        tm = np.zeros(1, np.int32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    n = lib.orc_search_by_projection(q.ctypes.data, dq.ctypes.data, len(q), kp.ctypes.data, d.ctypes.data,
                                     None if ur is None else ur.ctypes.data, len(kp), bounds[0], bounds[1], bounds[2],
                                     bounds[3], th_high, 1 if check_ori else 0, tm.ctypes.data)
    return n, tm[:len(kp)]


def search_by_projection_map(q, dq, kp, d, u_right, bounds, train_match, th_high=100, nn_ratio=0.8):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) restated; returns (nmatches, train_match)."""
    q = np.ascontiguousarray(q, PROJ_QUERY_DTYPE)
    dq = np.ascontiguousarray(dq, np.uint8)
    kp = np.ascontiguousarray(kp, KP_DTYPE)
    d = np.ascontiguousarray(d, np.uint8)
    tm = np.ascontiguousarray(train_match, np.int32).copy()
    if len(tm) == 0:
        tm = np.zeros(1, np.int32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    n = lib.orc_search_by_projection_map(q.ctypes.data, dq.ctypes.data, len(q), kp.ctypes.data, d.ctypes.data,
                                         None if ur is None else ur.ctypes.data, len(kp), bounds[0], bounds[1], bounds[2],
                                         bounds[3], th_high, nn_ratio, tm.ctypes.data)
    return n, tm[:len(kp)]


def distinctive_descriptor(desc):
    desc = np.ascontiguousarray(desc, np.uint8)
    return lib.orc_distinctive_descriptor(desc.ctypes.data, len(desc))


def make_vocabulary(rng, k, L, ragged=False):
    """Synthetic DBoW2-style tree (flat CSR): k children per node, L levels below the root; ragged = some inner nodes
    get fewer children and some branches end early (leaves at different depths)."""
    desc = [np.zeros(32, np.uint8)]; children = [[]]; depth = [0]
    frontier = [0]
    for lev in range(1, L + 1):
        nxt = []
        for parent in frontier:
            nk = k if not ragged else int(rng.integers(2, k + 1))
            if ragged and lev > 1 and rng.random() < 0.15:
                continue                                          # this node stays a leaf
            for _ in range(nk):
                nid = len(desc)
                d = desc[parent] ^ rng.integers(0, 256, 32, dtype=np.uint8) if lev == 1 else desc[parent] ^ (rng.integers(0, 256, 32, dtype=np.uint8) & rng.integers(0, 256, 32, dtype=np.uint8) & rng.integers(0, 256, 32, dtype=np.uint8))
                desc.append(d.astype(np.uint8)); children.append([]); depth.append(lev)
                children[parent].append(nid); nxt.append(nid)
        frontier = nxt
    n = len(desc)
    child_start = np.zeros(n + 1, np.int32); child_ids = []
    for i in range(n):
        child_ids += children[i]; child_start[i + 1] = len(child_ids)
    word = np.full(n, -1, np.int32); w = np.zeros(n, np.float64); nw = 0
    for i in range(n):
        if not children[i]:
            word[i] = nw; nw += 1; w[i] = float(rng.uniform(0.1, 9.0))
    return dict(node_desc=np.stack(desc), child_start=child_start, child_ids=np.array(child_ids, np.int32), node_word=word,
                node_weight=w, L=L, n_words=nw)


def bow_transform(feature, voc, levelsup):
    f = np.ascontiguousarray(feature, np.uint8)
    wid, nid = C.c_int32(), C.c_int32(); w = C.c_double()
    nd = np.ascontiguousarray(voc["node_desc"], np.uint8)
    lib.orc_bow_transform(f.ctypes.data, nd.ctypes.data, voc["child_start"].ctypes.data, voc["child_ids"].ctypes.data,
                          voc["node_word"].ctypes.data, voc["node_weight"].ctypes.data, voc["L"], levelsup,
                          C.byref(wid), C.byref(w), C.byref(nid))
    return wid.value, w.value, nid.value


def fuse_search(q, dq, kp, d, u_right, inv_level_sigma2, bounds):
    """Search part of ORBmatcher::Fuse restated; returns (best_idx [nq], best_dist [nq])."""
    q = np.ascontiguousarray(q, PROJ_QUERY_DTYPE); dq = np.ascontiguousarray(dq, np.uint8)
    kp = np.ascontiguousarray(kp, KP_DTYPE); d = np.ascontiguousarray(d, np.uint8)
    sig = np.ascontiguousarray(inv_level_sigma2, np.float32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    bi = np.zeros(max(len(q), 1), np.int32); bd = np.zeros(max(len(q), 1), np.int32)
    lib.orc_fuse_search(q.ctypes.data, dq.ctypes.data, len(q), kp.ctypes.data, d.ctypes.data, None if ur is None else ur.ctypes.data,
                        len(kp), sig.ctypes.data, bounds[0], bounds[1], bounds[2], bounds[3], bi.ctypes.data, bd.ctypes.data)
    return bi[:len(q)], bd[:len(q)]


def feature_vector_csr(nids):
    """Flatten a DBoW2 FeatureVector (node of every feature -> map<node, [feature indices in order]>): ids asc, start, feat."""
    nids = np.asarray(nids, np.int64)
    ids = np.unique(nids).astype(np.int32)
    order = np.argsort(nids, kind="stable").astype(np.int32)
    counts = np.array([(nids == i).sum() for i in ids], np.int32)
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return ids, start, order


def make_bow_case(rng, nk, nf, n_nodes=40):
    """Keyframe / frame with features spread over shared and private vocabulary nodes; frame descriptors are noisy copies of
    keyframe descriptors (same node) plus distractors and exact duplicates (ties)."""
    kp_k = np.zeros(nk, KP_DTYPE); kp_f = np.zeros(nf, KP_DTYPE)
    kp_k["angle"] = rng.uniform(0, 360, nk).astype(np.float32)
    d_k = rng.integers(0, 256, (nk, 32), dtype=np.uint8)
    nid_k = rng.integers(0, n_nodes, nk) * 3 + 100
    src = rng.integers(0, max(nk, 1), nf) if nk else np.zeros(nf, np.int64)
    if nk:
        d_f = d_k[src] ^ (rng.integers(0, 256, (nf, 32), dtype=np.uint8) & rng.integers(0, 256, (nf, 32), dtype=np.uint8) & rng.integers(0, 256, (nf, 32), dtype=np.uint8))
        nid_f = np.where(rng.random(nf) < 0.8, nid_k[src], rng.integers(0, n_nodes + 10, nf) * 3 + 101)
        kp_f["angle"] = (kp_k["angle"][src] + rng.choice([0, 0, 0, 120], nf) + rng.normal(0, 4, nf)).astype(np.float32) % np.float32(360)
        if nf > 10:
            d_f[nf // 2] = d_f[nf // 2 - 1]; nid_f[nf // 2] = nid_f[nf // 2 - 1]
    else:
        d_f = rng.integers(0, 256, (nf, 32), dtype=np.uint8); nid_f = rng.integers(0, n_nodes, nf) * 3 + 100
    valid = (rng.random(nk) < 0.85).astype(np.uint8)
    return dict(kp_k=kp_k, d_k=d_k, nid_k=nid_k, valid=valid, kp_f=kp_f, d_f=d_f, nid_f=nid_f)


def search_by_bow(c, nn_ratio=0.7, check_ori=True):
    ki, ks, kf = feature_vector_csr(c["nid_k"]); fi, fs, ff = feature_vector_csr(c["nid_f"])
    nF = len(c["kp_f"])
    m = np.zeros(max(nF, 1), np.int32)
    kpk = np.ascontiguousarray(c["kp_k"], KP_DTYPE); kpf = np.ascontiguousarray(c["kp_f"], KP_DTYPE)
    dk = np.ascontiguousarray(c["d_k"], np.uint8); df = np.ascontiguousarray(c["d_f"], np.uint8)
    va = np.ascontiguousarray(c["valid"], np.uint8)
    n = lib.orc_search_by_bow(ki.ctypes.data, ks.ctypes.data, kf.ctypes.data, len(ki), va.ctypes.data, kpk.ctypes.data, dk.ctypes.data,
                              fi.ctypes.data, fs.ctypes.data, ff.ctypes.data, len(fi), kpf.ctypes.data, df.ctypes.data, nF,
                              nn_ratio, 1 if check_ori else 0, m.ctypes.data)
    return n, m[:nF]


# ------------------------------------------------------------------ ORBmatcher::SearchForTriangulation oracle
def skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]], np.float32)


def make_tri_case(rng, n1, n2, n_nodes=60, stereo_frac=0.0, only_stereo=False, coarse=False):
    """Two keyframes seeing the same synthetic points: KF2's keypoints are reprojections (plus outliers off the epipolar line,
    points near the epipole, exact duplicate descriptors for the equal-distance rule)."""
    fx, fy, cx, cy = 458.0, 457.0, 367.0, 248.0
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
    ang = 0.05
    R12 = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)   # X1 = R12 X2 + t12
    t12 = np.array([0.35, 0.02, 0.9], np.float32)
    # float32 arithmetic throughout, as the cv::Mat (CV_32F) chain of Pinhole.cpp:124-127
    Kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    F12 = (Kinv.T @ skew(t12) @ R12 @ Kinv).astype(np.float32)
    # epipole: KF1's centre in KF2 (ORBm:978-992): C2 = R2w*Cw+t2w = R12^T (0 - t12)
    C2 = (-(R12.T @ t12)).astype(np.float32)
    ep = (np.float32(fx) * C2[0] / C2[2] + np.float32(cx), np.float32(fy) * C2[1] / C2[2] + np.float32(cy))
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    sigma2 = (scale * scale).astype(np.float32)
    kp1 = np.zeros(n1, KP_DTYPE); kp2 = np.zeros(n2, KP_DTYPE)
    X2 = np.stack([rng.uniform(-4, 4, n1), rng.uniform(-3, 3, n1), rng.uniform(2, 12, n1)], 1).astype(np.float32)
    X1 = (X2 @ R12.T + t12).astype(np.float32)
    kp1["x"] = fx * X1[:, 0] / X1[:, 2] + cx; kp1["y"] = fy * X1[:, 1] / X1[:, 2] + cy
    kp1["octave"] = rng.integers(0, 8, n1); kp1["angle"] = rng.uniform(0, 360, n1).astype(np.float32)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    nid1 = (rng.integers(0, n_nodes, n1) * 3 + 100).astype(np.int32)
    src = rng.integers(0, max(n1, 1), n2) if n1 else np.zeros(n2, np.int64)
    if n1:
        noise = rng.integers(0, 256, (n2, 32), dtype=np.uint8) & rng.integers(0, 256, (n2, 32), dtype=np.uint8) & rng.integers(0, 256, (n2, 32), dtype=np.uint8)
        noise[rng.random(n2) < 0.3] = 0                                   # exact copies: equal distances inside a node
        d2 = d1[src] ^ noise
        nid2 = np.where(rng.random(n2) < 0.85, nid1[src], rng.integers(0, n_nodes + 10, n2) * 3 + 101).astype(np.int32)
        kp2["x"] = fx * X2[src, 0] / X2[src, 2] + cx + rng.normal(0, 1.0, n2) * rng.choice([0.3, 1, 4], n2)
        kp2["y"] = fy * X2[src, 1] / X2[src, 2] + cy + rng.normal(0, 1.0, n2) * rng.choice([0.3, 1, 4], n2)
        near = rng.random(n2) < 0.05                                      # a few keypoints next to the epipole
        kp2["x"][near] = ep[0] + rng.uniform(-12, 12, near.sum()); kp2["y"][near] = ep[1] + rng.uniform(-12, 12, near.sum())
        kp2["angle"] = (kp1["angle"][src] + rng.choice([0, 0, 0, 90], n2) + rng.normal(0, 4, n2)).astype(np.float32) % np.float32(360)
    else:
        d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8); nid2 = (rng.integers(0, n_nodes, n2) * 3 + 100).astype(np.int32)
    kp2["octave"] = rng.integers(0, 8, n2)
    ur1 = np.where(rng.random(n1) < stereo_frac, kp1["x"] - 5, -1).astype(np.float32)
    ur2 = np.where(rng.random(n2) < stereo_frac, kp2["x"] - 5, -1).astype(np.float32)
    return dict(kp1=kp1, d1=d1, nid1=nid1, mp1=(rng.random(n1) < 0.3).astype(np.uint8), ur1=ur1,
                kp2=kp2, d2=d2, nid2=nid2, mp2=(rng.random(n2) < 0.2).astype(np.uint8), ur2=ur2,
                F12=F12.reshape(9).copy(), ep=ep, scale=scale, sigma2=sigma2, only_stereo=only_stereo, coarse=coarse)


lib.orc_search_for_triangulation.restype = ci
lib.orc_search_for_triangulation.argtypes = [vp] * 5 + [ci] + [vp] * 3 + [ci] + [vp] * 5 + [C.c_float, C.c_float, vp, vp, ci, ci, ci, vp]


def search_for_triangulation(c, check_ori=True, mono=False):
    i2, s2, f2 = feature_vector_csr(c["nid2"])
    n1 = len(c["kp1"])
    m = np.zeros(max(n1, 1), np.int32)
    a = {k: np.ascontiguousarray(c[k]) for k in ("nid1", "mp1", "kp1", "d1", "ur1", "mp2", "kp2", "d2", "ur2", "F12", "scale", "sigma2")}
    n = lib.orc_search_for_triangulation(a["nid1"].ctypes.data, a["mp1"].ctypes.data, a["kp1"].ctypes.data, a["d1"].ctypes.data,
                                         None if mono else a["ur1"].ctypes.data, n1, i2.ctypes.data, s2.ctypes.data, f2.ctypes.data, len(i2),
                                         a["mp2"].ctypes.data, a["kp2"].ctypes.data, a["d2"].ctypes.data, None if mono else a["ur2"].ctypes.data,
                                         a["F12"].ctypes.data, float(c["ep"][0]), float(c["ep"][1]), a["scale"].ctypes.data, a["sigma2"].ctypes.data,
                                         1 if c["only_stereo"] else 0, 1 if c["coarse"] else 0, 1 if check_ori else 0, m.ctypes.data)
    return n, m[:n1]


# ------------------------------------------------------------------ KF-KF SearchByBoW and the Sim3 searches
lib.orc_search_by_bow_kf.restype = ci
lib.orc_search_by_bow_kf.argtypes = [vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, cf, ci, vp]


def search_by_bow_kf(c, nn_ratio=0.75, check_ori=True):
    """c: make_bow_case dict + 'valid2'.  Returns (nmatches, matches12 [n1])."""
    i1, s1, f1 = feature_vector_csr(c["nid_k"]); i2, s2, f2 = feature_vector_csr(c["nid_f"])
    n1, n2 = len(c["kp_k"]), len(c["kp_f"])
    m = np.zeros(max(n1, 1), np.int32)
    a = {k: np.ascontiguousarray(c[k]) for k in ("kp_k", "d_k", "valid", "kp_f", "d_f", "valid2")}
    n = lib.orc_search_by_bow_kf(i1.ctypes.data, s1.ctypes.data, f1.ctypes.data, len(i1), a["valid"].ctypes.data, a["kp_k"].ctypes.data,
                                 a["d_k"].ctypes.data, n1, i2.ctypes.data, s2.ctypes.data, f2.ctypes.data, len(i2), a["valid2"].ctypes.data,
                                 a["kp_f"].ctypes.data, a["d_f"].ctypes.data, n2, nn_ratio, 1 if check_ori else 0, m.ctypes.data)
    return n, m[:n1]


lib.orc_search_by_projection_sim3.restype = ci
lib.orc_search_by_projection_sim3.argtypes = [vp, vp, ci, vp, vp, ci, cf, cf, cf, cf, cf, vp]
lib.orc_window_best.argtypes = [vp, vp, ci, vp, vp, ci, cf, cf, cf, cf, vp, vp]


def search_by_projection_sim3(q, dq, kp, d, bounds, matched, ratio_hamming):
    q = np.ascontiguousarray(q, PROJ_QUERY_DTYPE); dq = np.ascontiguousarray(dq, np.uint8)
    kp = np.ascontiguousarray(kp, KP_DTYPE); d = np.ascontiguousarray(d, np.uint8)
    tm = np.ascontiguousarray(matched, np.int32).copy()
    if len(tm) == 0:
        tm = np.zeros(1, np.int32)
    n = lib.orc_search_by_projection_sim3(q.ctypes.data, dq.ctypes.data, len(q), kp.ctypes.data, d.ctypes.data, len(kp),
                                          bounds[0], bounds[1], bounds[2], bounds[3], ratio_hamming, tm.ctypes.data)
    return n, tm[:len(kp)]


def window_best(q, dq, kp, d, bounds):
    q = np.ascontiguousarray(q, PROJ_QUERY_DTYPE); dq = np.ascontiguousarray(dq, np.uint8)
    kp = np.ascontiguousarray(kp, KP_DTYPE); d = np.ascontiguousarray(d, np.uint8)
    bi = np.zeros(max(len(q), 1), np.int32); bd = np.zeros(max(len(q), 1), np.int32)
    lib.orc_window_best(q.ctypes.data, dq.ctypes.data, len(q), kp.ctypes.data, d.ctypes.data, len(kp), bounds[0], bounds[1], bounds[2],
                        bounds[3], bi.ctypes.data, bd.ctypes.data)
    return bi[:len(q)], bd[:len(q)]


# ------------------------------------------------------------------ Frame glue
lib.orc_assign_features_to_grid.argtypes = [vp, ci, cf, cf, cf, cf, vp, vp]
lib.orc_undistort_keypoints.argtypes = [vp, ci, cf, cf, cf, cf, vp, ci, vp]


def assign_features_to_grid(kp, bounds):
    kp = np.ascontiguousarray(kp, KP_DTYPE)
    cs = np.zeros(64 * 48 + 1, np.int32); it = np.zeros(max(len(kp), 1), np.int32)
    lib.orc_assign_features_to_grid(kp.ctypes.data, len(kp), bounds[0], bounds[1], bounds[2], bounds[3], cs.ctypes.data, it.ctypes.data)
    return cs, it[:cs[-1]]


def undistort_keypoints(kp, K, dist):
    kp = np.ascontiguousarray(kp, KP_DTYPE); dc = np.ascontiguousarray(dist, np.float32)
    out = np.zeros(max(len(kp), 1), KP_DTYPE)
    lib.orc_undistort_keypoints(kp.ctypes.data, len(kp), K[0], K[1], K[2], K[3], dc.ctypes.data, len(dc), out.ctypes.data)
    return out[:len(kp)]


lib.orc_bow_vectors.argtypes = [vp, vp, vp, ci] + [vp] * 7


def bow_vectors(wid, w, nid):
    wid = np.ascontiguousarray(wid, np.int32); w = np.ascontiguousarray(w, np.float64); nid = np.ascontiguousarray(nid, np.int32)
    n = len(wid); m = max(n, 1)
    ni = np.zeros(m, np.int32); ns = np.zeros(m + 1, np.int32); ft = np.zeros(m, np.int32); nn = ci()
    bw = np.zeros(m, np.int32); bv = np.zeros(m, np.float64); nw = ci()
    lib.orc_bow_vectors(wid.ctypes.data, w.ctypes.data, nid.ctypes.data, n, ni.ctypes.data, ns.ctypes.data, ft.ctypes.data, C.byref(nn),
                        bw.ctypes.data, bv.ctypes.data, C.byref(nw))
    return ni[:nn.value], ns[:nn.value + 1], ft[:ns[nn.value]], bw[:nw.value], bv[:nw.value]


# ------------------------------------------------------------------ two-camera rig frames (Nleft != -1)
lib.orc_search_by_projection_rig.restype = ci
lib.orc_search_by_projection_rig.argtypes = [ci, vp, vp, ci, vp, vp, ci, ci, vp, C.c_float, C.c_float, C.c_float, C.c_float, ci, C.c_float, ci, vp]
lib.orc_search_by_bow_rig.restype = ci
lib.orc_search_by_bow_rig.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, ci, ci, C.c_float, ci, vp]


def search_by_projection_rig(mode, q, dq, kp, d, nleft, mirror, bounds, train_match, th_high=100, nn_ratio=0.8, check_ori=True):
    """ORBmatcher::SearchByProjection on a rig frame (mode 0: from the last frame, 1: local map points); returns (nmatches, train_match)."""
    q = np.ascontiguousarray(q, PROJ_QUERY_DTYPE); dq = np.ascontiguousarray(dq, np.uint8)
    kp = np.ascontiguousarray(kp, KP_DTYPE); d = np.ascontiguousarray(d, np.uint8)
    tm = np.ascontiguousarray(train_match, np.int32).copy()
    if len(tm) == 0:
        tm = np.zeros(1, np.int32)
    mi = None if mirror is None else np.ascontiguousarray(mirror, np.int32)
    n = lib.orc_search_by_projection_rig(mode, q.ctypes.data, dq.ctypes.data, len(q), kp.ctypes.data, d.ctypes.data, len(kp), nleft,
                                         None if mi is None else mi.ctypes.data, bounds[0], bounds[1], bounds[2], bounds[3], th_high, nn_ratio,
                                         1 if check_ori else 0, tm.ctypes.data)
    return n, tm[:len(kp)]


def search_by_bow_rig(c, nleft, nn_ratio=0.7, check_ori=True):
    ki, ks, kf = feature_vector_csr(c["nid_k"]); fi, fs, ff = feature_vector_csr(c["nid_f"])
    nF = len(c["kp_f"])
    m = np.zeros(max(nF, 1), np.int32)
    kpk = np.ascontiguousarray(c["kp_k"], KP_DTYPE); kpf = np.ascontiguousarray(c["kp_f"], KP_DTYPE)
    dk = np.ascontiguousarray(c["d_k"], np.uint8); df = np.ascontiguousarray(c["d_f"], np.uint8)
    va = np.ascontiguousarray(c["valid"], np.uint8)
    n = lib.orc_search_by_bow_rig(ki.ctypes.data, ks.ctypes.data, kf.ctypes.data, len(ki), va.ctypes.data, kpk.ctypes.data, dk.ctypes.data,
                                  fi.ctypes.data, fs.ctypes.data, ff.ctypes.data, len(fi), kpf.ctypes.data, df.ctypes.data, nF, nleft,
                                  nn_ratio, 1 if check_ori else 0, m.ctypes.data)
    return n, m[:nF]
