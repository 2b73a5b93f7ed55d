#!/bin/bash
# GPU-box helper: SQ counters of the BA leg (two PMC passes), summarised per kernel.  usage: tools/pmc_ba.sh <tag>
set -o pipefail
tag=${1:-ba}; root=$PWD; out=$root/gpurun_out/pmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/a -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 16 --ba-steps 1 --pose-frames 0 --stereo-pairs 0 > $out/a.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU -d $out/b -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 16 --ba-steps 1 --pose-frames 0 --stereo-pairs 0 > $out/b.log 2>&1 || exit 1
python3 tools/pmc_summary.py $out/a k_ba_ > $out/summary.txt 2>&1; python3 tools/pmc_summary.py $out/b k_ba_ >> $out/summary.txt 2>&1
echo done
