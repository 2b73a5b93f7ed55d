set -o pipefail
root=$PWD; out=$root/gpurun_out/sq_$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $out/sq -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 1; }
python3 tools/profile_summary.py valu $(find $out/sq -name "*counter_collection.csv") $out/valu.json
python3 - <<P
import json
d=json.load(open("$out/valu.json"))
for k,v in d["kernels"].items():
    if k in ("k_fast_cells","k_blur","k_resize","k_orient_desc","k_blur_score"):
        print(k, {a:round(b/1e6,1) if isinstance(b,float) else b for a,b in v.items()})
P
