#!/bin/bash
# LDS counters of the ORB kernels (bank conflicts, LDS busy) -- one PMC pass, no trace (GPU box, via gpurun)
set -o pipefail
root=$PWD; out=$root/gpurun_out/lds_$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/pmc -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --no-tracking --no-latency --no-hd-leg > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
python3 tools/pmc_summary.py $out/pmc k_fast_cells k_orient_desc k_octree k_blur_rows
