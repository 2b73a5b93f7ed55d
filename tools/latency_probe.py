#!/usr/bin/env python3
"""Per-call latency of orbhip_extract_batch_device at small batches, eager launches vs hipGraph replay (GPU box)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
import torch
import orbhip

ctx = orbhip.Context(0)
out = {}
for B in (1, 2, 8, 32):
    imgs = orbhip.synth_frames(640, 480, B, seed=9)
    d = torch.from_numpy(imgs).cuda()
    for mode in ("eager", "graph"):
        ext = orbhip.Extractor(ctx, 1000, 1.2, 8, 20, 7)
        ext.set_graph_mode(mode == "graph")
        for _ in range(20):
            ext.extract_device(d.data_ptr(), 640, 480, 640, 640 * 480, B, (0, 0))
        ctx.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            ext.extract_device(d.data_ptr(), 640, 480, 640, 640 * 480, B, (0, 0))
            ctx.synchronize()                      # latency: each call waited for, as Tracking does
        dt = (time.perf_counter() - t0) / n
        out["B%d_%s_us" % (B, mode)] = round(dt * 1e6, 1)
        ext.close()
print(json.dumps(out))
