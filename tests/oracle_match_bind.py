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


# ------------------------------------------------------------------ SearchForTriangulation, every camera combination
TRI_GENERAL_DTYPE = np.dtype([("R12", "<f4", (4, 9)), ("t12", "<f4", (4, 3)), ("F12", "<f4", (4, 9)), ("cam1", "<f4", (2, 8)), ("cam2", "<f4", (2, 8)),
                              ("cam1_type", "<i4", (2,)), ("cam2_type", "<i4", (2,)), ("ep_x", "<f4"), ("ep_y", "<f4"), ("nleft1", "<i4"), ("nleft2", "<i4"),
                              ("only_stereo", "<i4"), ("coarse", "<i4")])
lib.orc_search_for_triangulation_general.restype = ci
lib.orc_search_for_triangulation_general.argtypes = [vp] * 5 + [ci] + [vp] * 3 + [ci] + [vp] * 4 + [vp, vp, vp, vp, ci, vp]
TRI_POSES_DTYPE = np.dtype([("Tcw1", "<f4", (2, 12)), ("Tcw2", "<f4", (2, 12))])
lib.orc_search_for_triangulation_points.restype = ci
lib.orc_search_for_triangulation_points.argtypes = [vp] * 4 + [ci] + [vp] * 3 + [ci] + [vp] * 3 + [vp, vp, vp, vp, ci, vp, vp]
lib.orc_kb8_match_and_triangulate.restype = ci
lib.orc_kb8_match_and_triangulate.argtypes = [vp, ci, vp, cf, cf, cf, cf, vp, vp, cf, cf, vp]
lib.orc_kb8_triangulate_matches.restype = cf
lib.orc_kb8_triangulate_matches.argtypes = [ci, vp, ci, vp, cf, cf, cf, cf, vp, vp, cf, cf, vp]
lib.orc_camera_project_f.argtypes = [ci, vp, vp, vp]
lib.orc_camera_unproject_f.argtypes = [ci, vp, cf, cf, vp]


def camera_project_f(cam_type, params, P):
    p = np.ascontiguousarray(params, np.float32); x = np.ascontiguousarray(P, np.float32); uv = np.zeros(2, np.float32)
    lib.orc_camera_project_f(cam_type, p.ctypes.data, x.ctypes.data, uv.ctypes.data)
    return uv


def camera_unproject_f(cam_type, params, u, v):
    p = np.ascontiguousarray(params, np.float32); r = np.zeros(3, np.float32)
    lib.orc_camera_unproject_f(cam_type, p.ctypes.data, float(u), float(v), r.ctypes.data)
    return r


def kb8_triangulate_matches(type1, cam1, type2, cam2, p1, p2, R12, t12, sigma1, sigma2):
    """KannalaBrandt8::TriangulateMatches restated.  Returns (z1 or -1, x3D)."""
    c1 = np.ascontiguousarray(cam1, np.float32); c2 = np.ascontiguousarray(cam2, np.float32)
    R = np.ascontiguousarray(R12, np.float32).reshape(9); t = np.ascontiguousarray(t12, np.float32)
    x = np.zeros(3, np.float32)
    z = lib.orc_kb8_triangulate_matches(type1, c1.ctypes.data, type2, c2.ctypes.data, float(p1[0]), float(p1[1]), float(p2[0]), float(p2[1]),
                                        R.ctypes.data, t.ctypes.data, float(sigma1), float(sigma2), x.ctypes.data)
    return z, x


def kb8_project_np(cam, P):
    """double-precision KannalaBrandt8 / Pinhole projection for building test scenes (cam = (type, 8 params))"""
    t, p = cam
    P = np.asarray(P, np.float64)
    if t == 0:
        return np.stack([p[0] * P[..., 0] / P[..., 2] + p[2], p[1] * P[..., 1] / P[..., 2] + p[3]], -1)
    th = np.arctan2(np.hypot(P[..., 0], P[..., 1]), P[..., 2]); psi = np.arctan2(P[..., 1], P[..., 0])
    r = th + p[4] * th ** 3 + p[5] * th ** 5 + p[6] * th ** 7 + p[7] * th ** 9
    return np.stack([p[0] * r * np.cos(psi) + p[2], p[1] * r * np.sin(psi) + p[3]], -1)


def _rot(ax, ang):
    ax = np.asarray(ax, np.float64); ax = ax / np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def make_tri_general_case(rng, n1, n2, mode, n_nodes=60, only_stereo=False, coarse=False):
    """Two keyframes seeing the same synthetic points, in one of the reference's camera set-ups:
      'pinhole'  single Pinhole cameras (must agree with the fast-path kernel / oracle),
      'kb8'      single KannalaBrandt8 cameras (monocular fisheye: epipole test + triangulation constraint),
      'rig'      two-camera KannalaBrandt8 rigs (NLeft != -1: keypoints left | right, relative pose per camera pair).
    KF2's keypoints are reprojections of KF1's points into (one of) its camera(s), plus outliers off the epipolar geometry."""
    pin = np.array([458.0, 457.0, 367.0, 248.0, 0, 0, 0, 0], np.float32)
    kbl = np.array([190.9, 190.8, 254.9, 256.8, 0.0034, 0.0007, -0.0020, 0.0002], np.float32)
    kbr = np.array([190.4, 190.6, 252.7, 255.0, 0.0031, 0.0009, -0.0019, 0.0003], np.float32)
    rig = mode == "rig"
    ctype = 0 if mode == "pinhole" else 1
    camL = pin if mode == "pinhole" else kbl
    camR = kbr
    # X_right = Rrl X_left + trl (mTrl); relative pose of the left cameras X1l = Rll X2l + tll
    Rrl = _rot([0.1, 1.0, 0.05], 0.02); trl = np.array([-0.101, 0.0007, 0.0012])
    Rll = _rot([0.0, 1.0, 0.1], 0.06); tll = np.array([0.35, 0.02, 0.15])
    combos = {}
    combos[0] = (Rll, tll)
    combos[1] = (Rll @ Rrl.T, tll - Rll @ Rrl.T @ trl)                       # lr: X1l from X2r
    combos[2] = (Rrl @ Rll, Rrl @ tll + trl)                                 # rl: X1r from X2l
    combos[3] = (Rrl @ combos[1][0], Rrl @ combos[1][1] + trl)               # rr
    g = np.zeros(1, TRI_GENERAL_DTYPE)[0]
    for c in range(4 if rig else 1):
        g["R12"][c] = combos[c][0].astype(np.float32).reshape(9); g["t12"][c] = combos[c][1].astype(np.float32)
    g["cam1"][0] = camL; g["cam2"][0] = camL; g["cam1"][1] = camR; g["cam2"][1] = camR
    g["cam1_type"][:] = ctype; g["cam2_type"][:] = ctype
    if mode == "pinhole":
        K = np.array([[pin[0], 0, pin[2]], [0, pin[1], pin[3]], [0, 0, 1]], np.float32)
        Kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
        g["F12"][0] = (Kinv.T @ skew(g["t12"][0]) @ g["R12"][0].reshape(3, 3) @ Kinv).astype(np.float32).reshape(9)
    # epipole: KF1's (left) centre in KF2's left camera
    C2 = -(Rll.T @ tll)
    ep = kb8_project_np((ctype, camL.astype(np.float64)), C2)
    g["ep_x"], g["ep_y"] = np.float32(ep[0]), np.float32(ep[1])
    g["only_stereo"], g["coarse"] = int(only_stereo), int(coarse)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32); sigma2 = (scale * scale).astype(np.float32)
    # points in front of KF2's left camera; keypoint side (left / right camera) per keypoint
    nl1 = int(n1 * 0.55) if rig else n1
    nl2 = int(n2 * 0.5) if rig else n2
    g["nleft1"], g["nleft2"] = (nl1, nl2) if rig else (-1, -1)
    X2l = np.stack([rng.uniform(-3, 3, n1), rng.uniform(-2, 2, n1), rng.uniform(2, 10, n1)], 1)
    X1l = X2l @ Rll.T + tll
    right1 = np.arange(n1) >= nl1
    X1 = np.where(right1[:, None], X1l @ Rrl.T + trl, X1l)
    kp1 = np.zeros(n1, KP_DTYPE); kp2 = np.zeros(n2, KP_DTYPE)
    uv1 = np.where(right1[:, None], kb8_project_np((ctype, camR.astype(np.float64)), X1), kb8_project_np((ctype, camL.astype(np.float64)), X1))
    kp1["x"], kp1["y"] = uv1[:, 0], uv1[:, 1]
    kp1["octave"] = rng.integers(0, 8, n1); kp1["angle"] = rng.uniform(0, 360, n1).astype(np.float32)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    nid1 = (rng.integers(0, n_nodes, n1) * 3 + 100).astype(np.int32)
    nid1[rng.random(n1) < 0.03] = -1                                          # features in no vocabulary node
    src = rng.integers(0, max(n1, 1), n2) if n1 else np.zeros(n2, np.int64)
    right2 = np.arange(n2) >= nl2
    if n1:
        noise = rng.integers(0, 256, (n2, 32), dtype=np.uint8) & rng.integers(0, 256, (n2, 32), dtype=np.uint8) & rng.integers(0, 256, (n2, 32), dtype=np.uint8)
        noise[rng.random(n2) < 0.3] = 0
        d2 = d1[src] ^ noise
        nid2 = np.where(rng.random(n2) < 0.85, np.maximum(nid1[src], 100), rng.integers(0, n_nodes + 10, n2) * 3 + 101).astype(np.int32)
        X2 = np.where(right2[:, None], X2l[src] @ Rrl.T + trl, X2l[src])
        uv2 = np.where(right2[:, None], kb8_project_np((ctype, camR.astype(np.float64)), X2), kb8_project_np((ctype, camL.astype(np.float64)), X2))
        jit = rng.normal(0, 1.0, (n2, 2)) * rng.choice([0.2, 1.0, 6.0], n2)[:, None]
        kp2["x"], kp2["y"] = uv2[:, 0] + jit[:, 0], uv2[:, 1] + jit[:, 1]
        near = rng.random(n2) < 0.05
        kp2["x"][near] = g["ep_x"] + rng.uniform(-12, 12, near.sum()); kp2["y"][near] = g["ep_y"] + rng.uniform(-12, 12, near.sum())
        kp2["angle"] = (kp1["angle"][src] + rng.choice([0, 0, 0, 90], n2) + rng.normal(0, 4, n2)).astype(np.float32) % np.float32(360)
    else:
        d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8); nid2 = (rng.integers(0, n_nodes, n2) * 3 + 100).astype(np.int32)
    kp2["octave"] = rng.integers(0, 8, n2)
    sf = 0.0 if rig else 0.3                                                  # rig keyframes carry no stereo keypoints (mvuRight all -1)
    ur1 = np.where(rng.random(n1) < sf, kp1["x"] - 5, -1).astype(np.float32)
    ur2 = np.where(rng.random(n2) < sf, kp2["x"] - 5, -1).astype(np.float32)
    return dict(kp1=kp1, d1=d1, nid1=nid1, mp1=(rng.random(n1) < 0.3).astype(np.uint8), ur1=ur1,
                kp2=kp2, d2=d2, nid2=nid2, mp2=(rng.random(n2) < 0.2).astype(np.uint8), ur2=ur2, geom=g, scale=scale, sigma2=sigma2,
                sigma2_1=(sigma2 * np.float32(1.0)).astype(np.float32), mode=mode, rel=(Rll, tll, Rrl, trl))


def tri_case_poses(c, seed=0):
    """Absolute poses for a make_tri_general_case scene (the overload that returns the points works with GetPose() / GetRightPose(),
    ORBmatcher.cc:1307-1321): KF2's left camera is placed at an arbitrary world pose, the others follow from the case's relative
    poses.  Returns one TRI_POSES_DTYPE record (rows 0..2 of Tcw, row-major) -- [0] left, [1] right (zeros without a second camera)."""
    Rll, tll, Rrl, trl = c["rel"]
    rng = np.random.default_rng(4000 + seed)
    R2w = _rot(rng.normal(size=3), rng.uniform(0.2, 1.0)); t2w = rng.uniform(-2, 2, 3)
    R1w = Rll @ R2w; t1w = Rll @ t2w + tll                                    # X1l = Rll X2l + tll
    P = np.zeros(1, TRI_POSES_DTYPE)[0]
    rig = c["geom"]["nleft1"] != -1
    for name, (R, t) in (("Tcw1", (R1w, t1w)), ("Tcw2", (R2w, t2w))):
        P[name][0] = np.concatenate([R, t[:, None]], 1).astype(np.float32).reshape(12)
        if rig:
            P[name][1] = np.concatenate([Rrl @ R, (Rrl @ t + trl)[:, None]], 1).astype(np.float32).reshape(12)
    return P


def search_for_triangulation_points(c, poses, check_ori=True):
    """ORBmatcher::SearchForTriangulation(..., vMatchedPoints) (ORBmatcher.cc:1212-1402) on a make_tri_general_case scene.
    Returns (nmatches, matches12 [n1], points12 [n1][3])."""
    i2, s2, f2 = feature_vector_csr(c["nid2"])
    n1 = len(c["kp1"])
    m = np.zeros(max(n1, 1), np.int32); pts = np.zeros((max(n1, 1), 3), np.float32)
    a = {k: np.ascontiguousarray(c[k]) for k in ("nid1", "mp1", "kp1", "d1", "mp2", "kp2", "d2", "sigma2", "sigma2_1")}
    g = np.ascontiguousarray(np.array([c["geom"]], TRI_GENERAL_DTYPE)); P = np.ascontiguousarray(np.array([poses], TRI_POSES_DTYPE))
    n = lib.orc_search_for_triangulation_points(a["nid1"].ctypes.data, a["mp1"].ctypes.data, a["kp1"].ctypes.data, a["d1"].ctypes.data, n1,
                                                i2.ctypes.data, s2.ctypes.data, f2.ctypes.data, len(i2), a["mp2"].ctypes.data, a["kp2"].ctypes.data,
                                                a["d2"].ctypes.data, g.ctypes.data, P.ctypes.data, a["sigma2_1"].ctypes.data, a["sigma2"].ctypes.data,
                                                1 if check_ori else 0, m.ctypes.data, pts.ctypes.data)
    return n, m[:n1], pts[:n1]


def search_for_triangulation_general(c, check_ori=True):
    i2, s2, f2 = feature_vector_csr(c["nid2"])
    n1 = len(c["kp1"])
    m = np.zeros(max(n1, 1), np.int32)
    a = {k: np.ascontiguousarray(c[k]) for k in ("nid1", "mp1", "kp1", "d1", "ur1", "mp2", "kp2", "d2", "ur2", "scale", "sigma2", "sigma2_1")}
    g = np.ascontiguousarray(np.array([c["geom"]], TRI_GENERAL_DTYPE))
    n = lib.orc_search_for_triangulation_general(a["nid1"].ctypes.data, a["mp1"].ctypes.data, a["kp1"].ctypes.data, a["d1"].ctypes.data,
                                                 a["ur1"].ctypes.data, n1, i2.ctypes.data, s2.ctypes.data, f2.ctypes.data, len(i2),
                                                 a["mp2"].ctypes.data, a["kp2"].ctypes.data, a["d2"].ctypes.data, a["ur2"].ctypes.data,
                                                 g.ctypes.data, a["sigma2_1"].ctypes.data, a["scale"].ctypes.data, a["sigma2"].ctypes.data,
                                                 1 if check_ori else 0, m.ctypes.data)
    return n, m[:n1]
