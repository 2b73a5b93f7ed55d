"""GPU-box probe (debug build: make EXTRA=-DSBP_PROF): shader cycles of k_search_by_projection's phases for frame pair 0 of a VGA batch."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "orb-slam3-mac_amd", "python"))
import numpy as np, torch, orbhip
B, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 640, 480
ctx = orbhip.Context(0); ext = orbhip.Extractor(ctx, 1000, 1.2, 8, 20, 7); ext.reserve(W, H, B)
imgs = torch.from_numpy(orbhip.synth_frames(W, H, B, seed=7)).cuda()
ext.extract_device(imgs.data_ptr(), W, H, W, W * H, B, (0, 0)); ctx.synchronize()
kp_p, desc_p, cnt_p, _ = ext.results_device(); M = ext.max_keypoints
hip = C.CDLL("libamdhip64.so"); hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
kp = np.zeros((B, M), orbhip.KP_DTYPE); hip.hipMemcpy(kp.ctypes.data, kp_p, kp.nbytes, 2)
sf = ext.table(0)
q = np.zeros((B, M), orbhip.PROJ_QUERY_DTYPE)
q["u"] = kp["x"]; q["v"] = kp["y"]; q["angle"] = kp["angle"]; q["radius"] = np.float32(15.0) * sf[np.clip(kp["octave"], 0, 7)]
q["min_level"] = kp["octave"] - 1; q["max_level"] = kp["octave"] + 1; q["has_obs"] = 1; q["ur"] = -1
d_q = torch.from_numpy(q.view(np.uint8)).cuda()
tm = torch.full((B, M), -1, dtype=torch.int32, device="cuda"); nm = torch.zeros((B,), dtype=torch.int32, device="cuda")
buf = (C.c_longlong * 8)()
import time
for it in range(3):
    tm.fill_(-1); torch.cuda.synchronize()
    orbhip.lib.orbhip_debug_sbp_prof(buf, 1)
    t0 = time.perf_counter()
    orbhip.search_by_projection_device(ctx, d_q.data_ptr(), desc_p, cnt_p, M, kp_p + M * 28, desc_p + M * 32, None, cnt_p + 4, M, M, B - 1, (0.0, 0.0, float(W), float(H)), 100, True, tm.data_ptr(), nm.data_ptr())
    ctx.synchronize(); dt = time.perf_counter() - t0
    orbhip.lib.orbhip_debug_sbp_prof(buf, 0)
    v = list(buf)
    print("%.3f ms | setup %d  list %d  eval %d  update %d  tail %d | queries %d  candidates %d  matches(pair0) %d" % (dt * 1e3, v[0], v[1], v[2], v[3], v[4], v[5], v[6], int(nm[0].item())))
