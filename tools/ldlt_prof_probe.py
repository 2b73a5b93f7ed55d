#!/usr/bin/env python3
"""Debug helper (library built with EXTRA=-DLDLT_PROF): cycle counters of ldlt_solve_wg's phases over single-window BA solves."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import orbhip, synth_ba
ctx = orbhip.Context(0)
bb = orbhip.BaBatch(ctx, [synth_ba.make_graph(seed=50)])
bb.solve(); ctx.synchronize()
buf = (C.c_longlong * 8)()
orbhip.lib.orbhip_debug_ldlt_prof(buf, 1)
for _ in range(4):
    bb.solve()
ctx.synchronize()
orbhip.lib.orbhip_debug_ldlt_prof(buf, 0)
n = 4 * bb.ticks
print("per LDLT call (cycles): load %.0f diag %.0f rows %.0f trailing %.0f backsub %.0f  (ticks %d)" % tuple([buf[i] / n for i in range(5)] + [bb.ticks]))
