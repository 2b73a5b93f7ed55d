#!/bin/bash
# GPU-box helper: kernel durations of single-frame extraction calls (batch 1, eager), per kernel
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/trace_lat; rm -rf $out; mkdir -p $out
cat > $out/one.py <<'P'
import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "orb-slam3-mac_amd", "python"))
import torch, orbhip
ctx = orbhip.Context(0); ext = orbhip.Extractor(ctx, 1000, 1.2, 8, 20, 7)
d = torch.from_numpy(orbhip.synth_frames(640, 480, 1, seed=9)).cuda()
for _ in range(30):
    ext.extract_device(d.data_ptr(), 640, 480, 640, 640 * 480, 1, (0, 0)); ctx.synchronize()
P
timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o run --output-format csv -- python3 $out/one.py > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/trace_lat/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = collections.defaultdict(list)
for r in rows: d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = v[len(v) // 2:]                                     # steady state
    print("%-40s calls/frame %.1f  avg %.1f us  per frame %.1f us" % (k[:40], len(v) / 15.0, sum(v) / len(v), sum(v) / 15.0)); tot += sum(v) / 15.0
print("sum of kernel time per frame: %.1f us" % tot)
ts = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)[-12:]
print("last frame: first start -> last end %.1f us" % ((ts[-1][1] - ts[0][0]) / 1e3))
P
