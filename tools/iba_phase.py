"""Phase cycle counters of one inertial window (ORBHIP_IBA_PROF=1 prints them from the library).  usage: python tools/iba_phase.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ORBHIP_IBA_PROF"] = "1"
import orbhip
import synth_iba

ctx = orbhip.Context(0)
win = synth_iba.make_window(9100, n_opt=10, n_fixed_vis=20, n_points=600)
b = orbhip.IbaBatch(ctx, [win.struct(orbhip.IbaWindow)], [win.kf0], [win.pts0])
for _ in range(3):
    b.solve()
print(b.download()[3])
b.close()
ctx.close()
