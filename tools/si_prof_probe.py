"""GPU-box probe (debug build: make EXTRA=-DSI_PROF): shader cycles of k_search_init's phases for frame pair 0 of a VGA batch."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "orb-slam3-mac_amd", "python"))
import numpy as np, torch, orbhip
B, W, H = 64, 640, 480
ctx = orbhip.Context(0); ext = orbhip.Extractor(ctx, 1000, 1.2, 8, 20, 7); ext.reserve(W, H, B)
imgs = torch.from_numpy(orbhip.synth_frames(W, H, B, seed=7)).cuda()
ext.extract_device(imgs.data_ptr(), W, H, W, W * H, B, (0, 0)); ctx.synchronize()
kp, desc, cnt, _ = ext.results_device(); max_kp = ext.max_keypoints; ds = max_kp * 32
prev = torch.zeros((B, max_kp, 2), dtype=torch.float32, device="cuda"); m12 = torch.empty((B, max_kp), dtype=torch.int32, device="cuda"); nm = torch.zeros((B,), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
buf = (C.c_longlong * 8)()
for it in range(2):
    orbhip.prev_matched_init_device(ctx, kp, max_kp, B - 1, max_kp, prev.data_ptr()); ctx.synchronize()
    orbhip.lib.orbhip_debug_si_prof(buf, 1)
    orbhip.search_for_initialization_device(ctx, kp, desc, cnt, kp + max_kp * 28, desc + ds, cnt + 4, B - 1, max_kp, max_kp, (0.0, 0.0, float(W), float(H)), 100, 0.9, True, prev.data_ptr(), m12.data_ptr(), nm.data_ptr())
    ctx.synchronize(); orbhip.lib.orbhip_debug_si_prof(buf, 0)
    v = list(buf)
    print("setup %d  scan %d  dist+reduce %d  update %d  tail %d | iterations %d  candidates %d  matches(pair0) %d" % (v[0], v[1], v[2], v[3], v[4], v[5], v[6], int(nm[0].item())))
