#!/usr/bin/env python3
"""Local-BA latency at small batches: ms per solve of G concurrent 50 KF x 2000 pt x 10 obs windows (GPU box)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
import torch  # noqa: F401
import orbhip, synth_ba

ctx = orbhip.Context(0)
out = {}
for G in (1, 2, 8, 64):
    gs = [synth_ba.make_graph(seed=50 + i) for i in range(min(G, 4))]
    bb = orbhip.BaBatch(ctx, [gs[i % len(gs)] for i in range(G)])
    bb.solve(); ctx.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        bb.solve()
    ctx.synchronize()
    out["G%d_ms_per_solve" % G] = round((time.perf_counter() - t0) / n * 1e3, 3)
    out["G%d_ticks" % G] = bb.ticks
    st = bb.download()[3][0]
    out["G%d_trials0" % G] = st["lm_trials"]
    bb.close()
print(json.dumps(out))
