"""Timing probe for the inertial local BA: single-window latency and batched throughput on the GPU, the CPU oracle beside it.
Run on the GPU box: python tools/iba_probe.py [n_batch]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orbhip
import oracle_iba_bind as ib


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ctx = orbhip.Context(0)
    cases = [("10 KF + 40 fixed, 1500 pts", dict(n_opt=10, n_fixed_vis=40, n_points=1500), False),
             ("10 KF + 20 fixed, 600 pts", dict(n_opt=10, n_fixed_vis=20, n_points=600), False),
             ("25 KF + 60 fixed, 2000 pts (bLarge)", dict(n_opt=25, n_fixed_vis=60, n_points=2000, large=True), True)]
    for name, kw, large in cases:
        win = ib.make_window(77, **kw)
        s = win.struct(orbhip.IbaWindow)
        p = orbhip.iba_default_params(large)
        orbhip.inertial_ba_solve_batch(ctx, [s], [win.kf0], [win.pts0], p)
        t0 = time.perf_counter()
        for _ in range(5):
            r = orbhip.inertial_ba_solve_batch(ctx, [s], [win.kf0], [win.pts0], p)
        t1 = (time.perf_counter() - t0) / 5
        tc = float("nan")
        if not os.environ.get("ORBHIP_PROBE_NO_CPU"):       # (the counter passes only want the kernel)
            t0 = time.perf_counter()
            ib.solve(win, ib.default_params(large))
            tc = time.perf_counter() - t0
        print("%s: edges %d, n=%d unknowns; GPU single window %.2f ms (host packing + H2D + kernel + D2H), trials %d; CPU oracle %.1f ms"
              % (name, win.n_edges, 15 * kw["n_opt"], t1 * 1e3, r[3][0]["lm_trials"], tc * 1e3), flush=True)
        wins = [ib.make_window(100 + i, **kw) for i in range(8)]
        structs = [wins[i % 8].struct(orbhip.IbaWindow) for i in range(nb)]
        kfs = [wins[i % 8].kf0 for i in range(nb)]
        pts = [wins[i % 8].pts0 for i in range(nb)]
        orbhip.inertial_ba_solve_batch(ctx, structs, kfs, pts, p)
        t0 = time.perf_counter()
        orbhip.inertial_ba_solve_batch(ctx, structs, kfs, pts, p)
        tb = time.perf_counter() - t0
        print("   batch of %d windows: %.1f ms = %.0f windows/s" % (nb, tb * 1e3, nb / tb), flush=True)
        rb = orbhip.IbaBatch(ctx, structs, kfs, pts)         # the same batch resident: packed + uploaded once
        rb.solve(p)
        t0 = time.perf_counter()
        rb.solve(p)
        tr = time.perf_counter() - t0
        rb.close()
        r1 = orbhip.IbaBatch(ctx, [s], [win.kf0], [win.pts0])
        r1.solve(p)
        t0 = time.perf_counter()
        for _ in range(5):
            r1.solve(p)
        t1r = (time.perf_counter() - t0) / 5
        r1.close()
        print("   resident batch: %d windows %.1f ms = %.0f windows/s; single window %.2f ms" % (nb, tr * 1e3, nb / tr, t1r * 1e3), flush=True)
    ctx.close()
    if os.environ.get("ORBHIP_PROBE_MAPS"):              # address map of the process, to resolve a crash report of the exit handlers afterwards
        with open(os.environ["ORBHIP_PROBE_MAPS"], "w") as f:
            f.write(open("/proc/self/maps").read())


if __name__ == "__main__":
    main()
