#!/usr/bin/env python3
"""Kernel trace helper: 6 single-window local-BA solves (50 KF x 2000 pt x 10 obs) -- run under rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import orbhip, synth_ba
ctx = orbhip.Context(0)
bb = orbhip.BaBatch(ctx, [synth_ba.make_graph(seed=50)])
for _ in range(6):
    bb.solve()
ctx.synchronize()
print("ticks", bb.ticks)
bb.close(); ctx.close()
