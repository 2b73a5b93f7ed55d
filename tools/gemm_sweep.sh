# GPU-box ablation of the Schur GEMM (rebuilds liborbhip.so per configuration; run through gpurun).  Results of EXP != 0 are wrong on purpose.
set -e
for e in 1 2 3; do
  touch orb-slam3-mac_amd/csrc/ba_kernels.hip
  make -s -C orb-slam3-mac_amd lib/liborbhip.so EXTRA="-DGEMM_EXP=$e" > gpurun_out/mk.log 2>&1
  timeout -k 10 300 python bench.py --pose-frames 0 --stereo-pairs 0 --steps 2 --warmup 1 --batch 64 > gpurun_out/b_$e.json 2> gpurun_out/b_$e.err
  python -c "
import json;d=json.loads(open('gpurun_out/b_$e.json').read().strip().splitlines()[-1]);b=d['ba'];print('EXP=$e',b['value'],b['ms_per_batch'],b['roofline']['avg_launch_ms'])"
done
