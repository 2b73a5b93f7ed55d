"""GPU-box probe (debug build: make EXTRA=-DOCT_PROF): shader cycles of k_octree's phases for workgroup 0 (level 0 of frame 0)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "orb-slam3-mac_amd", "python"))
import numpy as np, torch, orbhip
B, W, H = 256, 640, 480
ctx = orbhip.Context(0); ext = orbhip.Extractor(ctx, 1000, 1.2, 8, 20, 7); ext.reserve(W, H, B)
imgs = torch.from_numpy(orbhip.synth_frames(W, H, B, seed=7)).cuda()
buf = (C.c_longlong * 8)()
for it in range(2):
    orbhip.lib.orbhip_debug_oct_prof(buf, 1)
    ext.extract_device(imgs.data_ptr(), W, H, W, W * H, B, (0, 0)); ctx.synchronize()
    orbhip.lib.orbhip_debug_oct_prof(buf, 0)
    v = list(buf)
    print("gather %d  roots %d  subdivision %d  best %d  output+perm %d | passes %d" % (v[0], v[1], v[2], v[3], v[4], v[5]))
