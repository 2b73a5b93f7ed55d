set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/kt; mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 64 --pose-frames 0 --stereo-pairs 0 --no-tracking --no-latency --no-hd-leg --no-4k-leg --inertial-windows 0 > $out/log.txt 2>&1
python3 - <<'P'
import csv,glob
f=glob.glob("gpurun_out/kt/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]: print(r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
P
