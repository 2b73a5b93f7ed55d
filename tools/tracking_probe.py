#!/usr/bin/env python3
"""Throughput of the per-frame tracking kernels on the bench workload (GPU box): SearchByProjection (last frame),
SearchLocalPoints variant, PoseOptimization -- 1023 consecutive frame pairs of the 1024-frame synthetic batch."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
import numpy as np
import torch
import orbhip

B, W, H = 1024, 640, 480
ctx = orbhip.Context(0)
ext = orbhip.Extractor(ctx, 1000, 1.2, 8, 20, 7)
imgs = orbhip.synth_frames(W, H, B, seed=20241004)
d_imgs = torch.from_numpy(imgs).cuda()
ext.extract_device(d_imgs.data_ptr(), W, H, W, W * H, B, (0, 0))
ctx.synchronize()
kp_p, desc_p, cnt_p, _ = ext.results_device()
M = ext.max_keypoints
# queries = keypoints of frame f (device->host once, then packed): u,v = position, radius 15*scale[oct], levels (o-1, o+1)
import ctypes as C
hip = C.CDLL("libamdhip64.so"); hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
kp = np.zeros((B, M), orbhip.KP_DTYPE); cnt = np.zeros(B, np.int32)
hip.hipMemcpy(kp.ctypes.data, kp_p, kp.nbytes, 2); hip.hipMemcpy(cnt.ctypes.data, cnt_p, cnt.nbytes, 2)
sf = ext.table(0)
q = np.zeros((B, M), orbhip.PROJ_QUERY_DTYPE)
q["u"] = kp["x"]; q["v"] = kp["y"]; q["angle"] = kp["angle"]; q["radius"] = np.float32(15.0) * sf[np.clip(kp["octave"], 0, 7)]
q["min_level"] = kp["octave"] - 1; q["max_level"] = kp["octave"] + 1; q["has_obs"] = 1; q["ur"] = -1
d_q = torch.from_numpy(q.view(np.uint8)).cuda()
out = {}
tm = torch.full((B, M), -1, dtype=torch.int32, device="cuda"); nm = torch.zeros((B,), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for P in (B - 1, 256, 1):
  for name, fn in (("search_by_projection", lambda: orbhip.search_by_projection_device(ctx, d_q.data_ptr(), desc_p, cnt_p, M, kp_p + M * 28, desc_p + M * 32, None, cnt_p + 4, M, M, P, (0.0, 0.0, float(W), float(H)), 100, True, tm.data_ptr(), nm.data_ptr())),
                 ("search_local_map", lambda: orbhip.search_local_map_device(ctx, d_q.data_ptr(), desc_p, cnt_p, M, kp_p + M * 28, desc_p + M * 32, None, cnt_p + 4, M, M, P, (0.0, 0.0, float(W), float(H)), 100, 0.8, tm.data_ptr(), nm.data_ptr()))):
    for it in range(4):
        tm.fill_(-1); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); ctx.synchronize(); dt = time.perf_counter() - t0
    out["%s_%dpairs_ms" % (name, P)] = round(dt * 1e3, 3); out[name + "_matches_per_pair"] = round(float(nm[:P].float().mean().item()), 1)
# ---- BoW tree descent + SearchByBoW + Fuse search + pose-only BA on the same batch
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_match_bind as om            # only its host-side synthetic-vocabulary / CSR helpers are used here
rng = np.random.default_rng(1)
voc = om.make_vocabulary(rng, 10, 4)
dv = [torch.from_numpy(np.ascontiguousarray(voc[k])).cuda() for k in ("node_desc", "child_start", "child_ids", "node_word", "node_weight")]
wid = torch.zeros((B, M), dtype=torch.int32, device="cuda"); ww = torch.zeros((B, M), dtype=torch.float64, device="cuda"); nid = torch.zeros((B, M), dtype=torch.int32, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    orbhip.bow_transform_device(ctx, desc_p, cnt_p, B, M, M, [t.data_ptr() for t in dv], 4, 2, wid.data_ptr(), ww.data_ptr(), nid.data_ptr())
    ctx.synchronize(); dt = time.perf_counter() - t0
out["bow_transform_1024frames_ms"] = round(dt * 1e3, 3)
h_nid = nid.cpu().numpy()
P = 256
arr = dict(ki=np.zeros((P, M), np.int32), ks=np.zeros((P, M + 1), np.int32), kf=np.zeros((P, M), np.int32), kn=np.zeros(P, np.int32),
           fi=np.zeros((P, M), np.int32), fs=np.zeros((P, M + 1), np.int32), ff=np.zeros((P, M), np.int32), fn=np.zeros(P, np.int32))
for p in range(P):
    for side, f in (("k", p), ("f", p + 1)):
        ids, st, fe = om.feature_vector_csr(h_nid[f, :cnt[f]])
        arr[side + "i"][p, :len(ids)] = ids; arr[side + "s"][p, :len(st)] = st; arr[side + ("f" if side == "k" else "f")][p, :len(fe)] = fe
        arr[side + "n"][p] = len(ids)
t = {k: torch.from_numpy(v).cuda() for k, v in arr.items()}
valid = torch.ones((P, M), dtype=torch.uint8, device="cuda"); mf = torch.zeros((P, M), dtype=torch.int32, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    orbhip.search_by_bow_device(ctx, [t["ki"].data_ptr(), t["ks"].data_ptr(), t["kf"].data_ptr(), t["kn"].data_ptr(), valid.data_ptr(), kp_p, desc_p],
                                [t["fi"].data_ptr(), t["fs"].data_ptr(), t["ff"].data_ptr(), t["fn"].data_ptr(), kp_p + M * 28, desc_p + M * 32], cnt_p + 4,
                                P, M, M, M, 0.7, True, mf.data_ptr(), nm.data_ptr())
    ctx.synchronize(); dt = time.perf_counter() - t0
out["search_by_bow_256pairs_ms"] = round(dt * 1e3, 3); out["search_by_bow_matches_per_pair"] = round(float(nm[:P].float().mean().item()), 1)
geom = np.zeros(P, orbhip.TRI_PAIR_DTYPE); geom["F12"] = np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32); geom["ep_x"] = 1e6; geom["ep_y"] = 1e6
d_geom = torch.from_numpy(geom.view(np.uint8)).cuda(); nomp = torch.zeros((B, M), dtype=torch.uint8, device="cuda")
m12 = torch.zeros((P, M), dtype=torch.int32, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    orbhip.search_for_triangulation_device(ctx, [nid.data_ptr(), nomp.data_ptr(), kp_p, desc_p, 0, cnt_p],
                                           [t["fi"].data_ptr(), t["fs"].data_ptr(), t["ff"].data_ptr(), t["fn"].data_ptr(), nomp.data_ptr(),
                                            kp_p + M * 28, desc_p + M * 32, 0, cnt_p + 4], d_geom.data_ptr(), P, M, M, M, sf, sf * sf, True,
                                           m12.data_ptr(), nm.data_ptr())
    ctx.synchronize(); dt = time.perf_counter() - t0
out["search_for_triangulation_256pairs_ms"] = round(dt * 1e3, 3); out["search_for_triangulation_matches_per_pair"] = round(float(nm[:P].float().mean().item()), 1)
q2 = q.copy(); q2["min_level"] = kp["octave"] - 1; q2["max_level"] = kp["octave"]; q2["radius"] = np.float32(3.0) * sf[np.clip(kp["octave"], 0, 7)]
d_q2 = torch.from_numpy(q2.view(np.uint8)).cuda()
bi = torch.zeros((B, M), dtype=torch.int32, device="cuda"); bd = torch.zeros((B, M), dtype=torch.int32, device="cuda")
sig = (1.0 / sf.astype(np.float32) ** 2).astype(np.float32)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    orbhip.fuse_search_device(ctx, d_q2.data_ptr(), desc_p, cnt_p, M, kp_p + M * 28, desc_p + M * 32, None, cnt_p + 4, M, M, B - 1, sig,
                              (0.0, 0.0, float(W), float(H)), bi.data_ptr(), bd.data_ptr())
    ctx.synchronize(); dt = time.perf_counter() - t0
out["fuse_search_1023pairs_ms"] = round(dt * 1e3, 3)
print(json.dumps(out))
