#!/bin/bash
# GPU-box helper: counters of the inertial local-BA kernel (k_iba_solve), each group in its own rocprofv3 --pmc pass (never with a trace):
# HBM traffic (FETCH_SIZE x2 KiB per tools/pmc_calibrate.hip, WRITE_SIZE KiB), issue / wait / LDS counters.  The probe solves one window
# (a team of 16 workgroups), then a batch of <n> windows, for three window sizes.   usage: tools/pmc_iba.sh [n_batch=64]
set -o pipefail
nb=${1:-64}
root=$PWD; out=$root/gpurun_out/pmc_iba; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
export ORBHIP_PROBE_NO_CPU=1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o run --output-format csv -- python3 tools/iba_probe.py $nb > $out/fetch.log 2>&1 || { tail -5 $out/fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/write -o run --output-format csv -- python3 tools/iba_probe.py $nb > $out/write.log 2>&1 || { tail -5 $out/write.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/sq -o run --output-format csv -- python3 tools/iba_probe.py $nb > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT -d $out/lds -o run --output-format csv -- python3 tools/iba_probe.py $nb > $out/lds.log 2>&1 || { tail -5 $out/lds.log; exit 1; }
python3 tools/pmc_iba_summary.py $out
