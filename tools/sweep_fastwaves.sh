#!/bin/bash
# GPU-box helper: two-pipeline throughput against the LDS share k_fast_cells takes per CU (ORBHIP_TUNE_FAST_WAVES)
for fw in 0 14 12 10 8; do for p in 2 3; do
ORBHIP_TUNE_FAST_WAVES=$fw timeout -k 10 300 python bench.py --pipelines $p --steps 16 --no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 > gpurun_out/b_sw.log 2>&1; echo -n "fast_waves=$fw pipelines=$p "; tail -1 gpurun_out/b_sw.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
