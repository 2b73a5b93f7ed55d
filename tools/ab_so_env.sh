#!/bin/bash
# usage: tools/ab_so_env.sh reps "libA libB" VAR "v1 v2 ...": default ORB leg per (library build, value of VAR) on one box
reps=$1; names=$2; var=$3; vals=$4
ORB="--no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --no-tracking --no-latency --no-hd-leg --no-4k-leg"
for r in $(seq $reps); do for n in $names; do
  cp tools/ab/liborbhip_$n.so orb-slam3-mac_amd/lib/liborbhip.so || exit 1
  for v in $vals; do
    echo -n "$n $var=$v: "; env $var=$v timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 $ORB | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" || exit 1
  done; done; done
