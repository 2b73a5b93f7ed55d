#!/usr/bin/env python3
"""SearchByProjection: time per call against the number of frame pairs, replay form vs sequential form (GPU box)."""
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
hip = C.CDLL("libamdhip64.so"); hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
kp_h = np.zeros((B, mk), orbhip.KP_DTYPE); hip.hipMemcpy(kp_h.ctypes.data, kp_p, kp_h.nbytes, 2)
sf = ext.table(0)
q = np.zeros((B, mk), orbhip.PROJ_QUERY_DTYPE)
q["u"] = kp_h["x"]; q["v"] = kp_h["y"]; q["angle"] = kp_h["angle"]; q["radius"] = np.float32(15.0) * sf[np.clip(kp_h["octave"], 0, 7)]
q["min_level"] = kp_h["octave"] - 1; q["max_level"] = kp_h["octave"] + 1; q["has_obs"] = 1; q["ur"] = -1
d_q = torch.from_numpy(q.view(np.uint8)).cuda()
d_tm = torch.full((B, mk), -1, dtype=torch.int32, device="cuda"); d_tn = torch.zeros((B,), dtype=torch.int32, device="cuda")
bnds = (0.0, 0.0, float(W), float(H))
out = {}
for mode in ("last_frame", "local_map"):
    for form, env in (("replay", "100000"), ("sequential", "0")):
        os.environ["ORBHIP_SBP_PARALLEL_MAX_PAIRS"] = env
        for P in (1, 8, 32, 64, 128, 256, 512, 1023):
            def call():
                if mode == "last_frame":
                    orbhip.search_by_projection_device(ctx, d_q.data_ptr(), desc_p, cnt_p, mk, kp_p + mk * 28, desc_p + ds, None, cnt_p + 4, mk, mk, P, bnds, 100, True, d_tm.data_ptr(), d_tn.data_ptr())
                else:
                    orbhip.search_local_map_device(ctx, d_q.data_ptr(), desc_p, cnt_p, mk, kp_p + mk * 28, desc_p + ds, None, cnt_p + 4, mk, mk, P, bnds, 100, 0.8, d_tm.data_ptr(), d_tn.data_ptr())
            ts = []
            for it in range(8):
                d_tm.fill_(-1); torch.cuda.synchronize()
                t0 = time.perf_counter(); call(); ctx.synchronize(); ts.append(time.perf_counter() - t0)
            out["%s_%s_%d" % (mode, form, P)] = round(min(ts[2:]) * 1e3, 4)
print(json.dumps(out))
