#!/bin/bash
# GPU-box helper: memory-path counters of chosen kernels, one rocprofv3 --pmc pass per counter group (never with a trace).
# PMC_GROUPS="A B C;D E" replaces the default counter groups (one pass per ';'-separated group).
# usage: tools/pmc_kernels.sh <out-tag> <kernel-regex> -- <program> [args]     (the program directly, e.g. python3 tools/ba_schur_ab.py 256)
set -o pipefail
tag=$1; rx=$2; shift 3
root=$PWD; out=$root/gpurun_out/pmck_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
i=0
if [ -n "$PMC_GROUPS" ]; then IFS=';' read -ra GROUPS_ <<< "$PMC_GROUPS"; else GROUPS_=(); fi
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp -d $out/p$i -o run --output-format csv -- "$@" > $out/p$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -3 $out/p$i.log; }
done
[ -n "$PMC_GROUPS" ] || for grp in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr TCC_TAG_STALL_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp -d $out/p$i -o run --output-format csv -- "$@" > $out/p$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -3 $out/p$i.log; }
done
python3 - $out "$rx" <<'P'
import csv, glob, json, re, sys, collections
out, rx = sys.argv[1], re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for path in glob.glob(out + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not rx.search(k): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
res = {k: {c: acc[k][c] / cnt[k][c] for c in sorted(acc[k])} for k in sorted(acc)}
for k in res: res[k]["launches"] = max(cnt[k].values())
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
P
