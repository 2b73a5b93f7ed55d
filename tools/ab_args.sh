#!/bin/bash
# usage: tools/ab_args.sh reps "args A" "args B" ...: the default ORB leg of bench.py with each argument string in turn (same box)
reps=$1; shift
ORB="--no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --no-tracking --no-latency --no-hd-leg --no-4k-leg"
for r in $(seq $reps); do for a in "$@"; do
  echo -n "[$a]: "; timeout -k 10 300 python3 bench.py --warmup 3 $ORB $a | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" || exit 1
done; done
