#!/bin/bash
# usage: tools/ab_so.sh reps "A B ..." [bench args]: the default ORB leg of bench.py with tools/ab/liborbhip_<name>.so in turn as the library (same box)
reps=$1; names=$2; shift 2
ORB="--no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --no-tracking --no-latency --no-hd-leg --no-4k-leg"
for r in $(seq $reps); do for v in $names; do
  cp tools/ab/liborbhip_$v.so orb-slam3-mac_amd/lib/liborbhip.so || exit 1
  echo -n "$v: "; timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 $ORB "$@" | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" || exit 1
done; done
