#!/bin/bash
# GPU-box helper: kernel trace of the ORB front-end step alone (serial steps), per-kernel average of the full-batch dispatches + start offsets
# inside a step.   usage: [ENV=...] tools/trace_frontend.sh <tag>
set -o pipefail
tag=${1:-fe}; root=$PWD; out=$root/gpurun_out/trace_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o run --output-format csv -- python3 bench.py --steps 6 --warmup 2 --pipelines 1 --no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --no-tracking --no-latency --no-hd-leg > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - $out/run_kernel_trace.csv <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"]); r["k"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
# full-batch step = from a k_fast_cells dispatch with the largest grid; take the last such and print the timeline of kernels around it
fc = [r for r in rows if r["k"].startswith("k_fast_cells")]
gmax = max(int(r.get("Grid_Size") or r.get("Grid_Size_X")) for r in fc)
fcs = [r for r in fc if int(r.get("Grid_Size") or r.get("Grid_Size_X")) == gmax]
ref = fcs[-2]
t0 = ref["s"] - 600000
print("timeline around one step (us relative to k_fast_cells start):")
for r in sorted(rows, key=lambda r: r["s"]):
    if t0 <= r["s"] <= ref["s"] + 3600000 and (r["e"] - r["s"]) > 15000:
        print("  %-28s start %8.1f  end %8.1f  dur %7.1f" % (r["k"][:28], (r["s"] - ref["s"]) / 1e3, (r["e"] - ref["s"]) / 1e3, (r["e"] - r["s"]) / 1e3))
P
