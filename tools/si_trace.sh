#!/bin/bash
# GPU-box helper: per-kernel durations of SearchForInitialization's two forms (rocprofv3 kernel trace of tools/si_sweep.py)
set -o pipefail
root=$PWD; out=$root/gpurun_out/si_trace; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o run --output-format csv -- python3 tools/si_sweep.py > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - $out/run_kernel_trace.csv <<'P'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith(("k_si_", "k_search_init", "void k_si_", "void k_search_init"))]
agg = collections.defaultdict(list)
for r in rows:
    g = r.get("Grid_Size") or r.get("Grid_Size_X")
    agg[(r["Kernel_Name"].split("(")[0], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(agg, key=lambda k: (k[0], int(k[1]))):
    v = agg[k]; print("%-22s grid %-9s n=%-3d min %.1f us  median %.1f us" % (k[0], k[1], len(v), min(v), sorted(v)[len(v) // 2]))
P
