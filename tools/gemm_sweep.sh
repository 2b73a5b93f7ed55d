# GPU-box sweep of the Schur-GEMM tiling knobs (rebuilds liborbhip.so per configuration; run through gpurun)
set -e
for cfg in "3 8 1 0" "3 8 1 1" "3 8 1 2"; do
  set -- $cfg
  touch orb-slam3-mac_amd/csrc/ba_kernels.hip
  make -s -C orb-slam3-mac_amd lib/liborbhip.so EXTRA="-DGEMM_C=$1 -DGEMM_NCH=$2 -DGEMM_TILE_TEST=$3 -DGEMM_EXP=$4" > gpurun_out/mk.log 2>&1
  timeout -k 10 300 python bench.py --pose-frames 0 --stereo-pairs 0 --steps 2 --warmup 1 > gpurun_out/b_$4.json 2> gpurun_out/b_$4.err
  python -c "
import json;d=json.loads(open('gpurun_out/b_$4.json').read().strip().splitlines()[-1]);b=d['ba'];print('C=$1 NCH=$2 T=$3 EXP=$4',b['value'],b['ms_per_batch'],b['roofline']['avg_launch_ms'],b['roofline']['frac'],b['roofline']['mfma_flops_issued_per_launch']/1e9)"
done
