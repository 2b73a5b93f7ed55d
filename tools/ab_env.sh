#!/bin/bash
# usage: tools/ab_env.sh VAR "v1 v2 ..." reps [bench args]: the default ORB leg of bench.py with VAR set to each value in turn, reps rounds (one box: A/B is only valid inside one call)
var=$1; vals=$2; reps=$3; shift 3
ORB="--no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --no-tracking --no-latency --no-hd-leg --no-4k-leg"
for r in $(seq $reps); do for v in $vals; do
  echo -n "$var=$v: "; env $var=$v timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 $ORB "$@" | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d.get('kernels_ms',{}))" || exit 1
done; done
