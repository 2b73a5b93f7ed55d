#!/bin/bash
# usage: tools/ab_so_kernels.sh "A B ...": per-stage times (roofline.all_kernels, stages timed one after the other) of bench.py's default ORB leg per library build
ORB="--no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --no-tracking --no-latency --no-hd-leg --no-4k-leg"
for v in $1; do
  cp tools/ab/liborbhip_$v.so orb-slam3-mac_amd/lib/liborbhip.so || exit 1
  echo -n "$v: "; timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 $ORB | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], {k: v['ms'] for k, v in d['roofline']['all_kernels'].items()})" || exit 1
done
