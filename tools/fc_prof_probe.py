#!/usr/bin/env python3
"""Phase split of k_fast_cells on the bench's default workload (1024 VGA frames), from the cycle counters of a DEBUG build:
     make -C orb-slam3-mac_amd EXTRA=-DFC_PROF -B build/orb_kernels.o lib/liborbhip.so && python tools/fc_prof_probe.py
Prints wave-cycles per phase (summed over the waves) as shares of the kernel's wave time, and cycles per cell."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
import numpy as np
import torch
import orbhip
import bench

B, W, H = int(os.environ.get("FC_B", 1024)), 640, 480
imgs = bench.synth_frames_parallel(orbhip, W, H, B, 20241004, 0)
d = torch.from_numpy(imgs).cuda()
ctx = orbhip.Context(0)
ext = orbhip.Extractor(ctx, 1000, 1.2, 8, 20, 7)
ext.reserve(W, H, B)
fn = orbhip.lib.orbhip_debug_fc_prof
fn.argtypes = [C.c_void_p, C.c_int]
for _ in range(2):
    ext.extract_device(d.data_ptr(), W, H, W, W * H, B, (0, 0))
ctx.synchronize()
out = (C.c_ulonglong * 12)()
assert fn(out, 1) == 0
N = 3
for _ in range(N):
    ext.extract_device(d.data_ptr(), W, H, W, W * H, B, (0, 0))
ctx.synchronize()
assert fn(out, 0) == 0
v = [int(x) for x in out]
names = ["stage rest", "sync", "necessary test", "arc scores", "nms+emit", "tail"]
sub = {"stage: wait for the loads": 11, "stage: lds writes": 8, "stage: next geometry": 9, "stage: issue loads": 10}
tot = v[7]
res = {"frames": B, "launches": N, "cells": v[6] // N, "wave_cycles_per_launch": tot // N,
       "share": {n: round(v[i] / tot, 4) for i, n in enumerate(names)}, "cycles_per_cell": {n: round(v[i] / max(v[6], 1), 1) for i, n in enumerate(names)},
       "stage_parts_share": {n: round(v[i] / tot, 4) for n, i in sub.items()}, "stage_parts_cycles_per_cell": {n: round(v[i] / max(v[6], 1), 1) for n, i in sub.items()},
       "unaccounted_share": round(1 - (sum(v[:6]) + sum(v[8:12])) / tot, 4)}
print(json.dumps(res))
