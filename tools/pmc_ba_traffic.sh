#!/bin/bash
# GPU-box helper: HBM traffic of the BA kernels (FETCH_SIZE / WRITE_SIZE in separate passes), default BA workload (256 graphs).
set -o pipefail
root=$PWD; out=$root/gpurun_out/pmc_ba_traffic; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 16 --ba-steps 1 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --no-tracking --no-latency --no-hd-leg --no-4k-leg > $out/fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $out/write -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 16 --ba-steps 1 --pose-frames 0 --stereo-pairs 0 --inertial-windows 0 --no-tracking --no-latency --no-hd-leg --no-4k-leg > $out/write.log 2>&1 || exit 1
python3 - <<'P'
import csv, json, collections, glob
def load(path, counter):
    tot = collections.Counter(); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", ""); tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n
f, nf = load(glob.glob("gpurun_out/pmc_ba_traffic/fetch/*counter_collection.csv")[0], "FETCH_SIZE")
w, nw = load(glob.glob("gpurun_out/pmc_ba_traffic/write/*counter_collection.csv")[0], "WRITE_SIZE")
out = {"workload": "256 graphs x (50 KF, 2000 points, 10 obs), bench.py BA leg", "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 (tools/pmc_calibrate.hip)", "kernels": {}}
for k in sorted(f):
    if not k.startswith("k_ba_"): continue
    out["kernels"][k] = {"launches": nf[k], "hbm_read_bytes_per_launch": int(f[k] * 2048 / nf[k]), "hbm_write_bytes_per_launch": int(w[k] * 1024 / max(nw[k], 1)),
                         "hbm_bytes_per_launch": int(f[k] * 2048 / nf[k] + w[k] * 1024 / max(nw[k], 1))}
json.dump(out, open("gpurun_out/pmc_ba_traffic/ba_traffic.json", "w"), indent=1)
print({k: v["hbm_bytes_per_launch"] for k, v in out["kernels"].items()})
P
