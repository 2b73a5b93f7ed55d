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
print(json.dumps(out))
