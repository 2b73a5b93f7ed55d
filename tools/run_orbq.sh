timeout -k 10 600 python -m pytest tests/test_gpu_orb.py -x -q > gpurun_out/t_orb.log 2>&1; tail -2 gpurun_out/t_orb.log
for p in 1 2; do timeout -k 10 300 python bench.py --pipelines $p --no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 > gpurun_out/b_rs.log 2>&1; tail -1 gpurun_out/b_rs.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms'])"; done
