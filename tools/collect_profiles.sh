#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + stats, and the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ/GRBM) of the default
# bench workload, each in its own rocprofv3 run (counters never share a run with a trace), then the per-(kernel, grid) summaries.
# (--pipelines 1 --serial-matchers: every kernel of a step alone on the chip apart from the blur beside FAST, so that a dispatch duration is that kernel's own)
# usage: tools/collect_profiles.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/
set -o pipefail
tag=${1:-rXX}; shift
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
ORB="--no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --pipelines 1 --serial-matchers --no-tracking --no-latency --no-hd-leg --no-4k-leg"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out/trace -o run --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --stereo-pairs 0 --pipelines 1 --serial-matchers --no-tracking --no-latency "$@" > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
echo trace done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 $ORB "$@" > $out/fetch.log 2>&1 || { tail -5 $out/fetch.log; exit 1; }
echo fetch done
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $out/write -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 $ORB "$@" > $out/write.log 2>&1 || { tail -5 $out/write.log; exit 1; }
echo write done
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $out/sq -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 $ORB "$@" > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 1; }
echo sq done
python3 tools/profile_summary.py trace $(find $out/trace -name "*kernel_trace.csv") $out/${tag}_kernel_trace_by_dispatch.csv
cp $(find $out/trace -name "*kernel_stats.csv") $out/${tag}_kernel_stats.csv
python3 tools/profile_summary.py traffic $(find $out/fetch -name "*counter_collection.csv") $(find $out/write -name "*counter_collection.csv") $out/${tag}_pmc_traffic.json
python3 tools/profile_summary.py valu $(find $out/sq -name "*counter_collection.csv") $out/${tag}_pmc_valu_issue.json
grep "^{\"metric\"" $out/trace.log | tail -1 > $out/${tag}_bench_under_trace.json
# the inertial local BA in its own trace
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/iba -o run --output-format csv -- python3 tools/iba_probe.py 64 > $out/iba.log 2>&1
python3 - $out/iba/run_kernel_trace.csv $out/${tag}_iba_kernel_trace.csv <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_iba" in r["Kernel_Name"]]
w = csv.writer(open(sys.argv[2], "w")); w.writerow(["kernel", "grid_threads", "workgroup", "duration_us"])
for r in rows: w.writerow(["k_iba_solve", r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 2)])
P
echo done
