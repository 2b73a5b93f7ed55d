#!/usr/bin/env python3
"""Summaries of rocprofv3 output that can be reconciled with bench.py WITHOUT arithmetic across batch sizes.

A bench.py run launches every ORB kernel at several batch sizes (the timed 1024-frame steps, the 4-frame host re-check,
the 256-pair stereo leg ...), so rocprofv3's own --stats table averages unlike launches.  Here every figure is kept per
(kernel, grid size): the full-batch dispatches are the rows with the largest grid of a kernel.

  profile_summary.py trace <kernel_trace.csv> <out.csv>
      per (kernel, grid): dispatches, avg / min / max / total duration (us), VGPRs, LDS bytes
  profile_summary.py traffic <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> <out.json>
      per kernel, full-batch dispatches only (its largest grid + every grid launched as often: k_resize has one per level):
      HBM bytes per step (one launch of each of those grids) and per launch.  FETCH_SIZE is doubled (tools/pmc_calibrate.hip; MI355X_MICROARCH.md "HBM"), both
      counters are in units of 1024 B.
  profile_summary.py valu <SQ counter_collection.csv> <out.json>
      per kernel, full-batch dispatches only: SQ_INSTS_VALU per step, busy fraction of the chip's vector issue slots
      (SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)), other SQ counters per launch when present
"""
import collections
import csv
import json
import sys


def kname(s):
    s = s.split("(")[0]
    return s.replace("void ", "").strip()


def trace(path, out):
    rows = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(path)):
        k = kname(r["Kernel_Name"])
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        rows[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        meta[(k, g)] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size_X"])
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_threads", "workgroup", "dispatches", "avg_us", "min_us", "max_us", "total_us", "vgpr", "agpr", "sgpr", "lds_bytes"])
        for (k, g) in sorted(rows, key=lambda kg: (kg[0], -kg[1])):
            if not k.startswith("k_"):
                continue
            d = rows[(k, g)]
            m = meta[(k, g)]
            w.writerow([k, g, m[4], len(d), "%.2f" % (sum(d) / len(d)), "%.2f" % min(d), "%.2f" % max(d), "%.1f" % sum(d), m[0], m[1], m[2], m[3]])


def load_counters(path):
    """-> {kernel: {grid: {"n": dispatches, counter: sum}}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = kname(r["Kernel_Name"])
        g = int(r["Grid_Size"])
        acc[k][g][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[(k, g)].add(r["Dispatch_Id"])
    for (k, g), s in seen.items():
        acc[k][g]["n"] = len(s)
    return acc


def full_batch(grids):
    """The full-batch dispatches of one kernel: its largest grid plus every other grid launched as often (k_resize has one
    grid per pyramid level).  -> (summed counters over those grids, dispatches of the largest grid, number of grids)"""
    gmax = max(grids)
    n = grids[gmax]["n"]
    tot = collections.defaultdict(float)
    ng = 0
    for g, c in grids.items():
        if c["n"] == n:
            ng += 1
            for name, v in c.items():
                if name != "n":
                    tot[name] += v
    return tot, n, ng


def traffic(fetch_csv, write_csv, out):
    F, Wr = load_counters(fetch_csv), load_counters(write_csv)
    res = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, per kernel for its LARGEST grid only "
                     "(= the full-batch dispatches); FETCH_SIZE doubled (gfx950 reports 1/2 of a coalesced stream: "
                     "tools/pmc_calibrate.hip, MI355X_MICROARCH.md HBM section), units of 1024 B", "kernels": {}}
    for k in sorted(F):
        if not k.startswith("k_") or k not in Wr:
            continue
        tf, nf, ngf = full_batch(F[k])
        tw, nw, ngw = full_batch(Wr[k])
        if ngf != ngw:
            continue
        f = tf["FETCH_SIZE"] / nf * 1024 * 2            # per step = one launch of every full-batch grid
        w = tw["WRITE_SIZE"] / nw * 1024
        res["kernels"][k] = {"largest_grid_threads": max(F[k]), "steps_profiled": int(nf), "launches_per_step": ngf,
                             "hbm_read_bytes_per_step": int(f), "hbm_write_bytes_per_step": int(w), "hbm_bytes_per_step": int(f + w),
                             "hbm_bytes_per_launch": int((f + w) / ngf)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_step"] for k, v in res["kernels"].items()}))


def valu(sq_csv, out):
    A = load_counters(sq_csv)
    res = {"method": "rocprofv3 --pmc SQ_INSTS_VALU ... GRBM_GUI_ACTIVE, per kernel for its LARGEST grid only; busy = SQ_INSTS_VALU x 4 "
                     "cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); one wave64 vector instruction = one 4-cycle issue slot of its SIMD",
           "file": out.split("/")[-1], "simds": 1024, "kernels": {}}
    for k in sorted(A):
        if not k.startswith("k_"):
            continue
        c, n, ng = full_batch(A[k])
        if not c.get("GRBM_GUI_ACTIVE"):
            continue
        e = {"largest_grid_threads": max(A[k]), "steps_profiled": int(n), "launches_per_step": ng}
        for name, v in sorted(c.items()):
            e[name + "_per_step"] = v / n
        e["valu_issue_busy_frac"] = round(c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0), 4)
        res["kernels"][k] = e
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v["valu_issue_busy_frac"] for k, v in res["kernels"].items()}))


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "trace":
        trace(sys.argv[2], sys.argv[3])
    elif mode == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    elif mode == "valu":
        valu(sys.argv[2], sys.argv[3])
    else:
        raise SystemExit(__doc__)
