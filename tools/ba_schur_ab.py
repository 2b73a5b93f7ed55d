"""A/B of the two pair-list Schur kernels on the bench batch: k_ba_schur_big<16> (ORBHIP_BA_SCHUR_ROWS=0) against the row-owner
k_ba_schur_rows (default).  Same batch, both forms: the off-diagonal blocks agree bit for bit, the diagonal ones to rounding (the solves: same LM decisions, estimates to 1e-9); prints the hipEvent time per Schur launch.
usage (GPU box): python tools/ba_schur_ab.py [n_graphs=256]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
import numpy as np
import orbhip
import synth_ba


def run(ctx, glist, rows):
    os.environ["ORBHIP_BA_SCHUR_ROWS"] = "1" if rows else "0"
    bb = orbhip.BaBatch(ctx, glist)
    bb.solve()
    t0 = time.perf_counter()
    for _ in range(2):
        bb.solve()
    dt = (time.perf_counter() - t0) / 2
    res = bb.download()
    bb.set_profiling(True)
    bb.solve()
    ms, n, _ = bb.gemm_profile()
    bb.set_profiling(False)
    bb.close()
    return res, dt, ms / max(n, 1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ctx = orbhip.Context(0)
    graphs = [synth_ba.make_graph(seed=1000 + i) for i in range(min(n, 8))]
    glist = [graphs[i % len(graphs)] for i in range(n)]
    a, dta, msa = run(ctx, glist, False)
    b, dtb, msb = run(ctx, glist, True)
    same = all(np.array_equal(x, y) for x, y in zip(a[0], b[0])) and all(np.array_equal(x, y) for x, y in zip(a[1], b[1])) and \
        all(np.array_equal(x, y) for x, y in zip(a[2], b[2])) and a[3] == b[3]
    dmax = max(max(float(np.max(np.abs(x - y))) for x, y in zip(a[0], b[0])), max(float(np.max(np.abs(x - y))) for x, y in zip(a[1], b[1])))
    decisions = all(x["lm_trials"] == y["lm_trials"] and x["iterations_run"] == y["iterations_run"] for x, y in zip(a[3], b[3])) and \
        all(np.array_equal(x, y) for x, y in zip(a[2], b[2]))
    print("graphs %d: k_ba_schur_big %.3f ms/launch, solve %.1f ms | k_ba_schur_rows %.3f ms/launch, solve %.1f ms | bit-identical: %s, max |diff| %.3g, same LM decisions and outliers: %s"
          % (n, msa, dta * 1e3, msb, dtb * 1e3, same, dmax, decisions), flush=True)
    ctx.close()
    return 0 if decisions and dmax < 1e-9 else 1


if __name__ == "__main__":
    sys.exit(main())
