#!/bin/bash
# GPU-box helper: per-dispatch kernel durations of the ORB-only bench (serial steps), printed per (kernel, grid)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/trace_orb; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace -d $out -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --pipelines 1 "$@" > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 tools/profile_summary.py trace $(find $out -name "*kernel_trace.csv") $out/by_dispatch.csv
cat $out/by_dispatch.csv
