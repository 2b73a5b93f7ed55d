# GPU-box helper: SQ counters of the 2-NN matcher kernel (two passes; counters never share a run with a trace)
set -o pipefail
root=$PWD; out=$root/gpurun_out/sq_bf2nn; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
ARGS="--steps 2 --warmup 1 --pipelines 1 --no-cpu-baseline --ba-graphs 0 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $out/a -o run --output-format csv -- python3 bench.py $ARGS > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $out/b -o run --output-format csv -- python3 bench.py $ARGS > $out/b.log 2>&1 || { tail -5 $out/b.log; exit 1; }
python3 - <<P
import csv, glob, collections
for d in ("a", "b"):
    f = glob.glob("$out/%s/*counter_collection.csv" % d)[0]
    acc = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "bf2nn" not in k: continue
        if int(r["Grid_Size"]) < 100000: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(k, {c: round(v / n[(k, c)] / 1e6, 3) for c, v in acc[k].items()}, "M per launch")
P
