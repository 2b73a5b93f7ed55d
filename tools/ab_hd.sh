#!/bin/bash
# usage: tools/ab_hd.sh VAR "v1 v2" reps [workload]: bench.py --workload hd|4k extraction + matching line per value of VAR (same box)
var=$1; vals=$2; reps=$3; wl=${4:-hd}
ORB="--no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --no-tracking --no-latency"
for r in $(seq $reps); do for v in $vals; do
  echo -n "$wl $var=$v: "; env $var=$v timeout -k 10 400 python3 bench.py --workload $wl --steps 5 --warmup 2 $ORB | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['config'].get('stage_ms'))" || exit 1
done; done
