export ORBHIP_BENCH_BACKEND=gloo ORBHIP_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 2 --warmup 1 --batch 64 --ba-graphs 8 --ba-steps 1 --pose-frames 64 --stereo-pairs 0 --ba-sharded-graphs 2 > gpurun_out/reh.json 2> gpurun_out/reh.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/reh.json").read().strip().splitlines()[-1]); print(d["n_gpus"], d["value"], d.get("ba_sharded"))
P
