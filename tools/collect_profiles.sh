#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + the two PMC passes of the default bench workload.
# (the stereo leg is left out: it runs the same ORB kernels at another batch size and would blur the per-launch averages)
# usage: tools/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/{trace,fetch,write}
set -o pipefail
tag=${1:-rXX}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out/trace -o run --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --stereo-pairs 0 > $out/trace.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 > $out/fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $out/write -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 > $out/write.log 2>&1 || exit 1
echo done
