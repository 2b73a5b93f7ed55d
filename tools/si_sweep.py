#!/usr/bin/env python3
"""SearchForInitialization: time per call against the number of frame pairs, replay form vs sequential form (GPU box)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
import numpy as np, torch, ctypes as C
import orbhip

B, W, H = 1024, 640, 480
ctx = orbhip.Context(0)
ext = orbhip.Extractor(ctx, 1000, 1.2, 8, 20, 7)
imgs = orbhip.synth_frames(W, H, B, seed=20241004)
d_imgs = torch.from_numpy(imgs).cuda()
ext.reserve(W, H, B)
ext.extract_device(d_imgs.data_ptr(), W, H, W, W * H, B, (0, 0)); ctx.synchronize()
kp_p, desc_p, cnt_p, _ = ext.results_device()
mk = ext.max_keypoints; ds = mk * 32
d_prev = torch.zeros((B, mk, 2), dtype=torch.float32, device="cuda")
d_m12 = torch.zeros((B, mk), dtype=torch.int32, device="cuda"); d_nm = torch.zeros((B,), dtype=torch.int32, device="cuda")
bnds = (0.0, 0.0, float(W), float(H))
out = {}
for form, env in (("replay", "1048576"), ("sequential", "0")):
    os.environ["ORBHIP_SI_PARALLEL_MAX_PAIRS"] = env
    for P in (1, 8, 32, 64, 128, 256, 512, 1023):
        ts = []
        for it in range(8):
            orbhip.prev_matched_init_device(ctx, kp_p, mk, P, mk, d_prev.data_ptr()); ctx.synchronize()
            t0 = time.perf_counter()
            orbhip.search_for_initialization_device(ctx, kp_p, desc_p, cnt_p, kp_p + mk * 28, desc_p + ds, cnt_p + 4, P, mk, mk, bnds, 100, 0.9, True,
                                                    d_prev.data_ptr(), d_m12.data_ptr(), d_nm.data_ptr())
            ctx.synchronize(); ts.append(time.perf_counter() - t0)
        out["%s_%d" % (form, P)] = round(min(ts[2:]) * 1e3, 4)
        out["%s_%d_matches" % (form, P)] = int(d_nm[:P].sum().item())
print(json.dumps(out))
