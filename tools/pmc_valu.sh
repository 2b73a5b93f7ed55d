#!/bin/bash
# GPU-box helper: VALU issue occupancy of the ORB kernels: SQ_INSTS_VALU (one 4-cycle issue slot each, summed over all SIMDs) against
# GRBM_GUI_ACTIVE (cycles the kernel kept the GPU busy) -> fraction of the chip's VALU issue slots used, independent of the clock.
set -o pipefail
root=$PWD; out=$root/gpurun_out/pmc_valu; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES -d $out/a -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 > $out/a.log 2>&1 || exit 1
python3 - <<'P'
import csv, json, collections, glob
f = glob.glob("gpurun_out/pmc_valu/a/*counter_collection.csv")[0]
tot = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
out = {"method": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE; busy = SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)",
       "workload": "bench.py default ORB leg (1024 VGA frames)", "simds": 1024, "kernels": {}}
for k in sorted(tot):
    if not k.startswith("k_") or not tot[k]["GRBM_GUI_ACTIVE"]: continue
    out["kernels"][k] = {"dispatches": n[k], "SQ_INSTS_VALU": tot[k]["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": tot[k]["SQ_ACTIVE_INST_VALU"],
                         "GRBM_GUI_ACTIVE": tot[k]["GRBM_GUI_ACTIVE"],
                         "valu_issue_busy_frac": round(tot[k]["SQ_INSTS_VALU"] * 4.0 / (1024.0 * tot[k]["GRBM_GUI_ACTIVE"] / 8.0), 4)}
json.dump(out, open("gpurun_out/pmc_valu/valu.json", "w"), indent=1)
print({k: v["valu_issue_busy_frac"] for k, v in out["kernels"].items()})
P
