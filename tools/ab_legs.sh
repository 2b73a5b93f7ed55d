#!/bin/bash
# usage: tools/ab_legs.sh VAR "v1 v2" reps: VGA + hd + 4k extraction legs of bench.py per value of VAR (same box)
var=$1; vals=$2; reps=$3
ORB="--no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --no-tracking --no-latency"
for r in $(seq $reps); do for v in $vals; do
  echo -n "$var=$v: "; env $var=$v timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 $ORB | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], 'hd', (d.get('hd') or {}).get('value'), '4k', (d.get('uhd') or d.get('4k') or {}).get('value'))" || exit 1
done; done
