"""Summary of tools/pmc_iba.sh's four counter passes: k_iba_solve launches in dispatch order, 16 per probe case (6 one-window one-shot calls,
2 batch calls, 2 resident batch solves, 6 resident one-window solves), averaged per (case, kind).  FETCH_SIZE x 2 KiB (tools/pmc_calibrate.hip),
WRITE_SIZE x 1 KiB.   usage: python3 tools/pmc_iba_summary.py gpurun_out/pmc_iba"""
import collections
import csv
import glob
import json
import sys

CASES = ["10 KF + 40 fixed, 1500 points (56 k edges)", "10 KF + 20 fixed, 600 points (13 k edges)", "25 KF + 60 fixed, 2000 points, bLarge (126 k edges)"]
KINDS = [("one_window", range(0, 6)), ("batch", range(6, 10)), ("one_window", range(10, 16))]


def main(out):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    grids = {}
    for d in ("fetch", "write", "sq", "lds"):
        for path in glob.glob("%s/%s/*counter_collection.csv" % (out, d)):
            rows = [r for r in csv.DictReader(open(path)) if "k_iba_solve" in r["Kernel_Name"]]
            ids = sorted({int(r["Dispatch_Id"]) for r in rows})
            order = {i: k for k, i in enumerate(ids)}
            for r in rows:
                k = order[int(r["Dispatch_Id"])]
                case, slot = divmod(k, 16)
                if case >= len(CASES):
                    continue
                kind = next(name for name, rg in KINDS if slot in rg)
                key = (case, kind)
                grids[key] = int(r.get("Grid_Size") or r.get("Grid_Size_X")) // 1024
                acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[key][r["Counter_Name"]] += 1
    res = {"method": "rocprofv3 --pmc, four separate passes over tools/iba_probe.py <n_batch>; per-launch averages", "cases": []}
    for (case, kind) in sorted(acc):
        a = {c: acc[(case, kind)][c] / cnt[(case, kind)][c] for c in acc[(case, kind)]}
        e = {"window": CASES[case], "launch": kind, "workgroups": grids[(case, kind)]}
        if "FETCH_SIZE" in a: e["hbm_read_bytes"] = int(a.pop("FETCH_SIZE") * 2048)
        if "WRITE_SIZE" in a: e["hbm_write_bytes"] = int(a.pop("WRITE_SIZE") * 1024)
        if a.get("SQ_ACTIVE_INST_LDS"): e["lds_conflict_cycles_per_lds_active_cycle"] = round(a.get("SQ_LDS_BANK_CONFLICT", 0) / a["SQ_ACTIVE_INST_LDS"], 3)
        if a.get("SQ_WAVE_CYCLES"): e["wait_frac_of_wave_cycles"] = round(a.get("SQ_WAIT_INST_ANY", 0) / a["SQ_WAVE_CYCLES"], 3)
        if a.get("GRBM_GUI_ACTIVE"): e["valu_insts_per_gpu_cycle"] = round(a.get("SQ_INSTS_VALU", 0) / a["GRBM_GUI_ACTIVE"], 3)
        e["counters"] = {c: int(v) for c, v in sorted(a.items())}
        res["cases"].append(e)
    json.dump(res, open("%s/iba_pmc.json" % out, "w"), indent=1)
    for e in res["cases"]:
        print({k: v for k, v in e.items() if k != "counters"})


if __name__ == "__main__":
    main(sys.argv[1])
