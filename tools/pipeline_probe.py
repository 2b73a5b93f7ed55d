#!/usr/bin/env python3
"""ORB extract + match throughput with 1 or 2 independent pipelines (context + extractor + match buffers each, own HIP stream),
batches alternating between them: how much of the latency-bound kernels' time (k_octree, k_fast_cells, k_search_init) hides behind
the VALU-bound ones of the other pipeline.  Not the bench default (per-kernel durations would no longer be one kernel's own).
usage: pipeline_probe.py [pipelines=2] [steps=10]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
import numpy as np, torch, orbhip
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, W, H = 1024, 640, 480
imgs = orbhip.synth_frames(W, H, B, seed=20241004)
d_imgs = torch.from_numpy(imgs).cuda()


class Pipe:
    def __init__(self):
        self.ctx = orbhip.Context(0)
        self.ext = orbhip.Extractor(self.ctx, 1000, 1.2, 8, 20, 7)
        self.ext.reserve(W, H, B)
        m = self.ext.max_keypoints
        self.m = m
        self.idx2 = torch.empty((B, m, 2), dtype=torch.int32, device="cuda"); self.dist2 = torch.empty((B, m, 2), dtype=torch.int32, device="cuda")
        self.acc = torch.zeros((B, m), dtype=torch.uint8, device="cuda"); self.prev = torch.zeros((B, m, 2), dtype=torch.float32, device="cuda")
        self.m12 = torch.empty((B, m), dtype=torch.int32, device="cuda"); self.nm = torch.empty((B,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        self.kp, self.desc, self.cnt, self.mono = self.ext.results_device()

    def step(self):
        c, m, ds = self.ctx, self.m, self.m * 32
        self.ext.extract_device(d_imgs.data_ptr(), W, H, W, W * H, B, (0, 0))
        orbhip.match_bf2nn_device(c, self.desc, self.cnt, ds, self.desc + ds, self.cnt + 4, ds, B - 1, m, 0.7, self.idx2.data_ptr(), self.dist2.data_ptr(), self.acc.data_ptr())
        orbhip.match_bf2nn_device(c, self.desc + (B - 1) * ds, self.cnt + 4 * (B - 1), ds, self.desc, self.cnt, ds, 1, m, 0.7,
                                  self.idx2.data_ptr() + (B - 1) * m * 8, self.dist2.data_ptr() + (B - 1) * m * 8, self.acc.data_ptr() + (B - 1) * m)
        orbhip.prev_matched_init_device(c, self.kp, m, B - 1, m, self.prev.data_ptr())
        orbhip.search_for_initialization_device(c, self.kp, self.desc, self.cnt, self.kp + m * 28, self.desc + ds, self.cnt + 4, B - 1, m, m,
                                                (0.0, 0.0, float(W), float(H)), 100, 0.9, True, self.prev.data_ptr(), self.m12.data_ptr(), self.nm.data_ptr())


pipes = [Pipe() for _ in range(NP)]
for p in pipes:
    p.step()
for p in pipes:
    p.ctx.synchronize()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    pipes[i % NP].step()
for p in pipes:
    p.ctx.synchronize()
dt = time.perf_counter() - t0
for p in pipes:
    p.ctx.check_status()
print(json.dumps({"pipelines": NP, "steps": K, "frames_per_s": round(B * K / dt, 1), "ms_per_step": round(dt / K * 1e3, 3)}))
