#!/bin/bash
# GPU-box helper: share of the batch whose blur runs beside k_fast_cells (the rest runs beside k_octree)
for sp in 100 80 70 60 50 40; do
ORBHIP_TUNE_BLUR_SPLIT=$sp timeout -k 10 300 python bench.py --steps 16 --no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 > gpurun_out/b_sw.log 2>&1; echo -n "blur_split=$sp "; tail -1 gpurun_out/b_sw.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
