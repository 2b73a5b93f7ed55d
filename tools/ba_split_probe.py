#!/usr/bin/env python3
"""Local-BA leg of bench.py as ONE batch of 256 windows on one context against TWO batches of 128 on two contexts of the same GPU
(the kernels of a batch are a dependent chain with latency-bound members -- k_ba_ldlt is one workgroup per window)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "synth"))
import torch, orbhip, synth_ba
G = int(os.environ.get("BA_G", 256)); STEPS = 3
graphs = [synth_ba.make_graph(seed=i) for i in range(8)]
glist = [graphs[i % 8] for i in range(G)]
c1, c2 = orbhip.Context(0), orbhip.Context(0)
def sync(): c1.synchronize(); c2.synchronize(); torch.cuda.synchronize()
def timed(fn):
    fn(); sync(); t0 = time.perf_counter()
    for _ in range(STEPS): fn()
    sync(); return (time.perf_counter() - t0) / STEPS * 1e3
for rep in range(2):
    b = orbhip.BaBatch(c1, glist)
    t_one = timed(lambda: b.solve()); b.close()
    for parts in (2, 4):
        cs = [c1, c2] + [orbhip.Context(0) for _ in range(parts - 2)]
        bs = [orbhip.BaBatch(cs[i], glist[i * G // parts:(i + 1) * G // parts]) for i in range(parts)]
        t = timed(lambda: [x.solve() for x in bs])
        print("rep %d: one batch of %d: %.2f ms; %d batches of %d on %d contexts: %.2f ms" % (rep, G, t_one, parts, G // parts, parts, t), flush=True)
        for x in bs: x.close()
        for c in cs[2:]: c.close()
