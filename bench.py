#!/usr/bin/env python3
"""bench.py -- ORB extract+match throughput on synthetic image batches (BASELINE.json configs[1], [2]) + local BA.

One "step" = one pass of the hot path over one batch already resident in HBM:
  ORBextractor::operator() on `batch` frames (pyramid, FAST+NMS, octree, blur, IC-angle, rBRIEF, lapping
  assembly) + Hamming 2-NN match of every frame against its successor + SearchForInitialization.
Workloads (--workload): vga = configs[1] (640x480 x1024, 1000 feats; the default and the metric's config),
  hd = configs[2] (1920x1080, 2000 feats, 512 frames per GPU = 4096 over 8 GPUs), 4k (3840x2160 x64).
Prints ONE JSON line (rank 0).  N>1: one process per GPU, frames sharded, no data-path collective
-> "scaling": "weak".  `python bench.py --gpus N` without a torch.distributed environment starts the N ranks itself
(child processes, spawned before this process touches the GPU).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

# /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0          # HBM3E 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0      # 6.29 TB/s measured (float4 copy)
L2_PEAK_GBS = 34500.0          # aggregate L2 rate, MI355X_MICROARCH.md "L2 (per XCD)": 8 x 4 MiB, ~34.5 TB/s
MAX_CLOCK_GHZ = 2.4
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
VALU_LANES_PER_CLK = 16384     # 1024 SIMDs x 16 lanes (one wave64 instruction = one 4-cycle issue slot)
# v_mfma_f64_16x16x4_f64: 2*16*16*4 = 2048 flop per instruction, one per 64 cycles per SIMD (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA = 64)
MFMA_F64_PEAK_TFLOPS = 2048.0 / 64.0 * N_SIMD * MAX_CLOCK_GHZ * 1e9 / 1e12      # 78.6

WORKLOADS = {      # name -> (width, height, nfeatures, frames per GPU, stereo pairs)
    "vga": (640, 480, 1000, 1024, 256),
    "hd": (1920, 1080, 2000, 512, 64),
    "4k": (3840, 2160, 2000, 64, 0),
}
PROFILE_TAG = "r04"


def level_dims(w, h, nlevels=8, scale=1.2):
    import numpy as np
    sf = np.float32(1.0)
    dims = []
    for l in range(nlevels):
        inv = np.float32(1.0) / sf
        dims.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        sf = np.float32(sf * np.float32(scale))
    return dims


def algorithmic_bytes(w, h, n_kp, n_cand, nlevels=8):
    """per-frame algorithmic bytes of each stage.  SURVEY's B_extract has no term for the octree and the
    assembly (list work); for them the bytes of the lists they must touch are used so that every stage can be priced."""
    dims = level_dims(w, h, nlevels)
    S = sum(a * b for a, b in dims)
    S_lo = S - dims[-1][0] * dims[-1][1]
    S_hi = S - dims[0][0] * dims[0][1]
    return {"pyramid": S_lo + S_hi,                      # pyramid reads + writes
            "blur": 2 * S,                               # Gaussian blur: read S, write S
            "fast_cells": S,                             # FAST: every pyramid pixel read once (scores, the per-cell NMS and the
                                                         # two-threshold rule stay on chip; the candidate list is n_cand * 4, in "octree")
            "octree": n_cand * 4 * 2 + n_kp * 4,         # candidate keys in, node ids, selected keys out
            "desc": n_kp * (749 + 512) + n_kp * 36,      # orientation patch + BRIEF samples + angle/descriptor out
            "assemble": n_kp * 60 * 2,
            "total": S_lo + S_hi + S + 2 * S + n_kp * (749 + 512) + n_kp * 60, "S": S}


def cpu_baseline(w, h, nfeat, seconds_budget=20.0):
    """Oracle (CPU restatement of the reference path) timed on this host's cores: kind 'port'."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    import orbhip
    import oracle_bind as ob
    import oracle_match_bind as om
    cores = min(os.cpu_count() or 1, 16)
    imgs = orbhip.synth_frames(w, h, 3, seed=4242)
    e = ob.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    t0 = time.time()
    res = [e.extract(imgs[i], (0, 0)) for i in range(3)]

    def match(a, b):
        om.bf2nn(a[1], b[1], 0.7)
        om.search_for_initialization(a[0], a[1], b[0], b[1], (0.0, 0.0, float(w), float(h)),
                                     np.stack([a[0]["x"], a[0]["y"]], 1), 100, 0.9, True)
    for i in range(2):
        match(res[i], res[i + 1])
    t1 = (time.time() - t0) / 2.0
    per_thread = max(2, min(48, int(seconds_budget / max(t1, 1e-3))))

    def work(tid):
        ee = ob.OracleExtractor(nfeat, 1.2, 8, 20, 7)
        fr = orbhip.synth_frames(w, h, per_thread + 1, seed=777, first=tid * 64)
        prev = ee.extract(fr[0], (0, 0))
        for i in range(1, per_thread + 1):
            cur = ee.extract(fr[i], (0, 0))
            match(prev, cur)
            prev = cur
        return per_thread

    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(cores)))
    dt = time.time() - t0
    return {"value": round(done / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port", "single_thread_ms": round(t1 * 1e3, 3),
            "sample": "%d threads x %d frames %dx%d, %d feats, extract + BF 2-NN + SearchForInitialization vs successor; "
                      "single-thread %.2f frames/s" % (cores, per_thread, w, h, nfeat, 1.0 / t1)}


def pose_cpu_baseline(probs, seconds_budget=6.0):
    """PoseOptimization oracle on this host's cores: kind 'port'."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle_ba_bind as obb
    cores = min(os.cpu_count() or 1, 16)
    p = probs[0]
    t0 = time.time()
    obb.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["cam"], p["pose0"])
    t1 = time.time() - t0
    per_thread = max(4, min(400, int(seconds_budget / max(t1, 1e-4))))

    def work(tid):
        for i in range(per_thread):
            q = probs[(tid + i) % len(probs)]
            obb.pose_optimization(q["Xw"], q["obs"], q["inv_sigma2"], q["cam"], q["pose0"])
        return per_thread
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(cores)))
    dt = time.time() - t0
    return {"value": round(done / dt, 1), "unit": "frames/s", "cores": cores, "kind": "port", "single_thread_ms": round(t1 * 1e3, 3),
            "sample": "%d threads x %d frames of 1000 unary edges; single-thread %.1f frames/s" % (cores, per_thread, 1.0 / t1)}


def ba_cpu_baseline(graphs, seconds_budget=12.0):
    """BA oracle (CPU restatement of g2o LM+Schur) on this host's cores: kind 'port'."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle_ba_bind as obb
    cores = min(os.cpu_count() or 1, 16)
    t0 = time.time()
    obb.solve(graphs[0])
    t1 = time.time() - t0
    per_thread = max(1, min(8, int(seconds_budget / max(t1, 1e-3))))

    def work(tid):
        for i in range(per_thread):
            obb.solve(graphs[(tid + i) % len(graphs)])
        return per_thread
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(cores)))
    dt = time.time() - t0
    return {"value": round(done / dt, 2), "unit": "solves/s", "cores": cores, "kind": "port", "single_thread_ms": round(t1 * 1e3, 3),
            "sample": "%d threads x %d solves of the 50KFx2000ptx10obs graph; single-thread %.2f solves/s"
                      % (cores, per_thread, 1.0 / t1)}


def iba_cpu_baseline(wins, seconds_budget=6.0):
    """Oracle LocalInertialBA (oracle/iba_oracle.c) on the host cores, one window per thread at a time: kind 'port'."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle_iba_bind as oib
    cores = min(os.cpu_count() or 1, 16)
    t0 = time.time()
    oib.solve(wins[0])
    t1 = time.time() - t0
    per_thread = max(1, min(16, int(seconds_budget / max(t1, 1e-3))))

    def work(tid):
        for i in range(per_thread):
            oib.solve(wins[(tid + i) % len(wins)])
        return per_thread
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(cores)))
    dt = time.time() - t0
    return {"value": round(done / dt, 2), "unit": "windows/s", "cores": cores, "kind": "port", "single_thread_ms": round(t1 * 1e3, 3),
            "sample": "%d threads x %d solves of the bench windows; single-thread %.1f ms per window" % (cores, per_thread, t1 * 1e3)}


def tracking_cpu_baseline(kp_h, desc_h, cnt_h, q, bounds, seconds_budget=5.0):
    """Oracle ORBmatcher::SearchByProjection (last frame and local map) on the pairs of the tracking leg: kind 'port'."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    import oracle_match_bind as om
    cores = min(os.cpu_count() or 1, 16)
    B = len(cnt_h)

    def pair(i, which):
        a, b = i % (B - 1), i % (B - 1) + 1
        na, nb = int(cnt_h[a]), int(cnt_h[b])
        tm = np.full(nb, -1, np.int32)
        if which == 0:
            return om.search_by_projection(q[a, :na], desc_h[a, :na], kp_h[b, :nb], desc_h[b, :nb], None, bounds, tm, 100, True)[0]
        return om.search_by_projection_map(q[a, :na], desc_h[a, :na], kp_h[b, :nb], desc_h[b, :nb], None, bounds, tm, 100, 0.8)[0]
    out = {}
    for which, name in ((0, "search_by_projection_last_frame"), (1, "search_by_projection_local_map")):
        pair(0, which)
        t0 = time.time()
        for i in range(3):
            pair(i, which)
        t1 = (time.time() - t0) / 3
        per_thread = max(4, min(200, int(seconds_budget / 2 / max(t1, 1e-4))))

        def work(tid):
            for i in range(per_thread):
                pair(tid * per_thread + i, which)
            return per_thread
        t0 = time.time()
        with ThreadPoolExecutor(cores) as ex:
            done = sum(ex.map(work, range(cores)))
        dt = time.time() - t0
        out[name] = {"value": round(done / dt, 1), "unit": "frame pairs/s", "cores": cores, "kind": "port", "single_thread_ms": round(t1 * 1e3, 3),
                     "sample": "%d threads x %d pairs of the leg's own queries / keypoints" % (cores, per_thread)}
    return out


def stereo_cpu_baseline(lefts, rights, nfeat, mb, mbf, seconds_budget=8.0):
    """Oracle stereo front-end (two extractions + Frame::ComputeStereoMatches) on pairs of the stereo leg: kind 'port'."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle_bind as ob
    cores = min(os.cpu_count() or 1, 16)

    def one(eL, eR, j):
        kl, dl, _ = eL.extract(lefts[j], (0, 0))
        kr, dr, _ = eR.extract(rights[j], (0, 0))
        return ob.compute_stereo_matches(eL, eR, kl, dl, kr, dr, mb, mbf)[0]
    eL, eR = ob.OracleExtractor(nfeat, 1.2, 8, 20, 7), ob.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    t0 = time.time()
    one(eL, eR, 0)
    t1 = time.time() - t0
    per_thread = max(1, min(24, int(seconds_budget / max(t1, 1e-3))))

    def work(tid):
        a, b = ob.OracleExtractor(nfeat, 1.2, 8, 20, 7), ob.OracleExtractor(nfeat, 1.2, 8, 20, 7)
        for i in range(per_thread):
            one(a, b, (tid * per_thread + i) % len(lefts))
        return per_thread
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(cores)))
    dt = time.time() - t0
    return {"value": round(done / dt, 2), "unit": "pairs/s", "cores": cores, "kind": "port", "single_thread_ms": round(t1 * 1e3, 3),
            "sample": "%d threads x %d pairs of the leg's own images: 2 x extract + ComputeStereoMatches" % (cores, per_thread)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="vga",
                    help="vga = BASELINE configs[1] (default, the metric's config); hd = configs[2] (512 frames per GPU); 4k")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU (default: the workload's)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--nfeatures", type=int, default=None)
    ap.add_argument("--ba-graphs", type=int, default=256, help="local-BA graphs solved concurrently per GPU (0 = skip BA leg)")
    ap.add_argument("--ba-steps", type=int, default=3)
    ap.add_argument("--ba-sharded-graphs", type=int, default=0, help="N > 1 only, opt-in: graphs solved cooperatively with the points sharded over the ranks and the Schur block all-gathered every LM trial (SURVEY 8e optional mode)")
    ap.add_argument("--pose-frames", type=int, default=1024, help="frames of pose-only BA solved per launch (0 = skip)")
    ap.add_argument("--pipelines", type=int, default=1,
                    help="1 = strictly serial steps (default).  > 1: the K steps alternate between this many independent pipelines (own "
                         "context, extractor, HIP streams and match buffers; every step is still one full pass over one resident batch); "
                         "paid +7 %% before the blur became an LDS-free kernel that fills k_fast_cells' idle slots, nothing since")
    ap.add_argument("--serial-matchers", action="store_true",
                    help="run the two matchers of a step one after the other on the extraction's stream (the default puts "
                         "SearchForInitialization on a second context of the same GPU, ordered after the extraction by orbhip_ctx_wait_for: "
                         "one latency-bound wave per pair beside the matrix-core 2-NN kernel -- 3.53 -> 3.38 ms per step)")
    ap.add_argument("--no-tracking", dest="tracking", action="store_false", help="skip the SearchByProjection legs")
    ap.add_argument("--inertial-windows", type=int, default=32, help="LocalInertialBA windows solved per call (0 = skip)")
    ap.add_argument("--stereo-pairs", type=int, default=None, help="rectified stereo pairs for the ComputeStereoMatches leg (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", dest="latency", action="store_false", help="skip the batch-1 latency object")
    ap.add_argument("--no-hd-leg", dest="hd_leg", action="store_false", help="skip the 1920x1080 leg of the default (vga) run")
    ap.add_argument("--hd-frames", type=int, default=512, help="frames of the hd leg (512 = BASELINE config #3's per-GPU shard)")
    ap.add_argument("--no-4k-leg", dest="uhd_leg", action="store_false", help="skip the 3840x2160 leg of the default (vga) run")
    ap.add_argument("--4k-frames", dest="uhd_frames", type=int, default=128, help="frames of the 4k leg")
    args = ap.parse_args(argv)
    w, h, nf, b, sp = WORKLOADS[args.workload]
    args.width = args.width or w
    args.height = args.height or h
    args.nfeatures = args.nfeatures or nf
    args.batch = args.batch or b
    if args.stereo_pairs is None:
        args.stereo_pairs = sp
    return args


def spawn_ranks(args):
    """`bench.py --gpus N` outside a torch.distributed environment: start N ranks (one per GPU) as child processes of
    `python -m torch.distributed.run` and return its exit code.  Nothing in THIS process has touched the GPU yet (no torch, no
    orbhip import), and it never execs: it waits for the child and passes its output and return code through."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["ORBHIP_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def synth_frames_parallel(orbhip, w, h, n, seed, first):
    """orbhip.synth_frames on the host cores (the C generator releases the GIL); frame ids first .. first+n-1."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    if n * w * h < (64 << 20):
        return orbhip.synth_frames(w, h, n, seed=seed, first=first)
    out = np.empty((n, h, w), np.uint8)
    chunks = [(i, min(i + 8, n)) for i in range(0, n, 8)]      # 8 = the generator's sequence length: chunks start on a sequence

    def gen(c):
        out[c[0]:c[1]] = orbhip.synth_frames(w, h, c[1] - c[0], seed=seed, first=first + c[0])
    with ThreadPoolExecutor(min(os.cpu_count() or 1, 16)) as ex:
        list(ex.map(gen, chunks))
    return out


def load_profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(spawn_ranks(args))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU "
                         "(python bench.py --gpus N starts them itself)" % (args.gpus, env_world))

    import numpy as np
    import torch
    import torch.distributed as dist
    import orbhip

    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # rehearsal hooks (single-GPU boxes): ORBHIP_BENCH_BACKEND=gloo keeps the collectives on the CPU and
    # ORBHIP_BENCH_DEVICE pins every rank to one card; the driver's runs use neither (RCCL, one rank per GPU).
    backend = os.environ.get("ORBHIP_BENCH_BACKEND", "nccl")
    if "ORBHIP_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["ORBHIP_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    ranks_seen = [0]
    # ORBHIP_BENCH_FORCE_DIST=1: build the process group also for one rank (a 1-GPU box then runs the RCCL code path end to end)
    distributed = world > 1 or os.environ.get("ORBHIP_BENCH_FORCE_DIST") == "1"
    if distributed and "MASTER_ADDR" not in os.environ:      # forced one-rank group outside a launcher
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
    if distributed:
        import datetime
        tmo = datetime.timedelta(seconds=600)      # a rank that died turns into an error on the others, not a hang
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        # who is there: every rank's id through the collective backend itself (RCCL with backend nccl)
        mine = torch.tensor([rank], dtype=torch.int32, device=coll_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ranks_seen = sorted(int(t.item()) for t in allr)
        assert ranks_seen == list(range(world)), ranks_seen

    def barrier():
        if distributed:
            dist.barrier()

    def max_over_ranks(*vals):
        if not distributed:
            return [float(v) for v in vals]
        t = torch.tensor(list(vals), dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    B, W, H = args.batch, args.width, args.height
    # synthetic frames: generated on the host, then resident in HBM before the timed region
    imgs = synth_frames_parallel(orbhip, W, H, B, 20241004, rank * B)
    d_imgs = torch.from_numpy(imgs).cuda()
    ctx = orbhip.Context(local_rank)
    ext = orbhip.Extractor(ctx, args.nfeatures, 1.2, 8, 20, 7)
    ext.reserve(W, H, B)
    max_kp = ext.max_keypoints
    d_idx2 = torch.empty((B, max_kp, 2), dtype=torch.int32, device="cuda")
    d_dist2 = torch.empty((B, max_kp, 2), dtype=torch.int32, device="cuda")
    d_acc = torch.zeros((B, max_kp), dtype=torch.uint8, device="cuda")
    d_prev = torch.zeros((B, max_kp, 2), dtype=torch.float32, device="cuda")
    d_m12 = torch.empty((B, max_kp), dtype=torch.int32, device="cuda")
    d_nm = torch.zeros((B,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    kp_p, desc_p, cnt_p, mono_p = ext.results_device()
    dstride = max_kp * 32
    lib_stream = torch.cuda.ExternalStream(ctx.stream)      # the library's stream, for hipEvents around the matcher launches

    def extract():
        # lap (0,0): keypoints come out in level order (the stereo constructors' lapping, Frame.cc:109-110)
        ext.extract_device(d_imgs.data_ptr(), W, H, W, W * H, B, (0, 0))

    def bf_match():
        # frame i vs frame i+1 (B-1 pairs) + wrap-around pair (B-1 vs 0): every frame matched once
        if B > 1:
            orbhip.match_bf2nn_device(ctx, desc_p, cnt_p, dstride, desc_p + dstride, cnt_p + 4, dstride, B - 1, max_kp,
                                      0.7, d_idx2.data_ptr(), d_dist2.data_ptr(), d_acc.data_ptr())
        orbhip.match_bf2nn_device(ctx, desc_p + (B - 1) * dstride, cnt_p + 4 * (B - 1), dstride, desc_p, cnt_p, dstride, 1,
                                  max_kp, 0.7, d_idx2.data_ptr() + (B - 1) * max_kp * 8,
                                  d_dist2.data_ptr() + (B - 1) * max_kp * 8, d_acc.data_ptr() + (B - 1) * max_kp)

    # The two matchers of a step read the same extraction and are independent of each other: the windowed one (one latency-bound wave per
    # frame pair) runs on a second context of the same GPU beside the matrix-core 2-NN kernel.  orbhip_ctx_wait_for orders the streams:
    # the windowed matcher after this step's extraction, the next extraction after the windowed matcher (it overwrites what that reads).
    ctx2 = None if args.serial_matchers else orbhip.Context(local_rank)

    def windowed(c=None):
        # ORBmatcher::SearchForInitialization(frame i, frame i+1) (Tracking.cc:1506-1507: ORBmatcher(0.9,true),
        # windowSize 100) with vbPrevMatched = frame i's keypoint positions (Tracking.cc:1497-1499); B-1 pairs.
        c = c or ctx
        if B > 1:
            orbhip.prev_matched_init_device(c, kp_p, max_kp, B - 1, max_kp, d_prev.data_ptr())
            orbhip.search_for_initialization_device(c, kp_p, desc_p, cnt_p, kp_p + max_kp * 28, desc_p + dstride, cnt_p + 4,
                                                    B - 1, max_kp, max_kp, (0.0, 0.0, float(W), float(H)), 100, 0.9, True,
                                                    d_prev.data_ptr(), d_m12.data_ptr(), d_nm.data_ptr())

    def step():
        if ctx2 is not None:
            ctx.wait_for(ctx2)
            extract()
            ctx2.wait_for(ctx)
            windowed(ctx2)
            bf_match()
            return
        extract()
        bf_match()
        windowed()

    class Pipe:
        """One more independent pipeline of the same step (throughput mode): same resident input batch, own context (= own HIP
        streams), own extractor and match buffers."""

        def __init__(self):
            self.ctx = orbhip.Context(local_rank)
            self.ext = orbhip.Extractor(self.ctx, args.nfeatures, 1.2, 8, 20, 7)
            self.ext.reserve(W, H, B)
            self.idx2 = torch.empty((B, max_kp, 2), dtype=torch.int32, device="cuda")
            self.dist2 = torch.empty((B, max_kp, 2), dtype=torch.int32, device="cuda")
            self.acc = torch.zeros((B, max_kp), dtype=torch.uint8, device="cuda")
            self.prev = torch.zeros((B, max_kp, 2), dtype=torch.float32, device="cuda")
            self.m12 = torch.empty((B, max_kp), dtype=torch.int32, device="cuda")
            self.nm = torch.zeros((B,), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            self.kp, self.desc, self.cnt, _ = self.ext.results_device()

        def step(self):
            c = self.ctx
            self.ext.extract_device(d_imgs.data_ptr(), W, H, W, W * H, B, (0, 0))
            if B > 1:
                orbhip.match_bf2nn_device(c, self.desc, self.cnt, dstride, self.desc + dstride, self.cnt + 4, dstride, B - 1, max_kp,
                                          0.7, self.idx2.data_ptr(), self.dist2.data_ptr(), self.acc.data_ptr())
            orbhip.match_bf2nn_device(c, self.desc + (B - 1) * dstride, self.cnt + 4 * (B - 1), dstride, self.desc, self.cnt, dstride, 1,
                                      max_kp, 0.7, self.idx2.data_ptr() + (B - 1) * max_kp * 8,
                                      self.dist2.data_ptr() + (B - 1) * max_kp * 8, self.acc.data_ptr() + (B - 1) * max_kp)
            if B > 1:
                orbhip.prev_matched_init_device(c, self.kp, max_kp, B - 1, max_kp, self.prev.data_ptr())
                orbhip.search_for_initialization_device(c, self.kp, self.desc, self.cnt, self.kp + max_kp * 28, self.desc + dstride,
                                                        self.cnt + 4, B - 1, max_kp, max_kp, (0.0, 0.0, float(W), float(H)), 100, 0.9,
                                                        True, self.prev.data_ptr(), self.m12.data_ptr(), self.nm.data_ptr())

    extra = [Pipe() for _ in range(max(args.pipelines, 1) - 1)]
    steppers = [step] + [q.step for q in extra]

    def sync():
        ctx.synchronize()
        for q in extra:
            q.ctx.synchronize()
        torch.cuda.synchronize()

    # ---- the timed region: W warm-up steps, then exactly K steps, stage profiling OFF -------------------------------
    for i in range(max(args.warmup, len(steppers))):
        steppers[i % len(steppers)]()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        steppers[i % len(steppers)]()
    sync()
    barrier()
    dt = time.perf_counter() - t0
    (dt,) = max_over_ranks(dt)
    for q in extra:
        q.ctx.check_status()
        assert int(device_view(torch, q.cnt, (B,), "<i4").min().item()) > 0, "a pipeline produced an empty frame"
    for q in extra:                                      # the other legs run on pipeline 0 only
        q.ext.close(); q.ctx.close()
    extra.clear()

    # ---- separate profiled pass (not part of `value`): hipEvents on the library's stream around every kernel's launches
    ext.set_profiling(True)
    n_prof = max(2, min(args.steps, 8))
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_prof)]
    t0 = time.perf_counter()
    for i in range(n_prof):
        ev[i][0].record(lib_stream)
        extract()
        ev[i][1].record(lib_stream)
        bf_match()
        ev[i][2].record(lib_stream)
        windowed()
        ev[i][3].record(lib_stream)
    sync()
    dt_prof = time.perf_counter() - t0
    stage = ext.stage_ms()
    ext.set_profiling(False)
    stage["match_bf2nn"] = sum(e[1].elapsed_time(e[2]) for e in ev) / n_prof
    stage["search_init"] = sum(e[2].elapsed_time(e[3]) for e in ev) / n_prof
    extract_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / n_prof

    ctx.check_status()                                   # loud failure on any device-side capacity overflow
    if ctx2 is not None:
        ctx2.check_status()
    res_chk = ext.extract_host(imgs[:min(B, 4)], (0, 0))  # host entry point re-check (raises on capacity errors)
    n_kp_avg = float(np.mean([len(r[0]) for r in res_chk]))
    n_cand_avg = float(np.mean([sum(len(ext.fast_candidates(f, l)[0]) for l in range(8)) for f in range(min(B, 4))]))
    extract()                                            # restore the full batch's results for the checks below
    sync()
    d_cnt = device_view(torch, cnt_p, (B,), "<i4").clone()      # keypoints per frame (the extractor's device counts)
    win_matches = float(d_nm[:max(B - 1, 1)].float().mean().item()) if B > 1 else 0.0
    bf_accept = float(d_acc.float().sum().item()) / B

    records_gathered = B
    if distributed:
        # the only data exchange of the sharded ORB path: one all-gather of fixed-size per-frame records
        # (keypoint count, windowed matches), outside the timed region (a host-side Tracking consumer would D2H per GPU)
        import shard
        rec = torch.stack([d_cnt, d_nm.to(torch.int32)], 1).to(coll_dev)
        allrec = shard.allgather_records(rec)
        assert allrec.shape[0] == world and allrec.shape[1] == B
        records_gathered = int(allrec.shape[0] * allrec.shape[1])
        assert int((allrec[:, :, 0] > 0).sum().item()) == records_gathered, "a rank returned an empty frame record"

    # ---- local-BA leg: G graphs per GPU solved concurrently (replicas) ----------
    ba = None
    graphs = None
    if args.ba_graphs > 0:
        import synth_ba
        distinct = min(args.ba_graphs, 8)
        graphs = [synth_ba.make_graph(seed=1000 * rank + i) for i in range(distinct)]
        glist = [graphs[i % distinct] for i in range(args.ba_graphs)]
        bb = orbhip.BaBatch(ctx, glist)
        bb.solve()                                   # warm-up
        barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.ba_steps):
            bb.solve()
        sync()
        barrier()
        dt_ba = time.perf_counter() - t0
        (dt_ba,) = max_over_ranks(dt_ba)
        ticks = bb.ticks
        poses_def, points_def, _, stats = bb.download()
        # separate profiled solve (hipEvents around every Schur launch), not part of `value`
        bb.set_profiling(True)
        bb.solve()
        sync()
        schur_ms, schur_n, _ = bb.gemm_profile()
        bb.set_profiling(False)
        bb.close()
        # the same batch through the FP64-MFMA panel GEMM (the round-1 design, selectable): reported beside the default
        orbhip.ba_set_schur_mode(ctx, 2)
        bg = orbhip.BaBatch(ctx, glist)
        orbhip.ba_set_schur_mode(ctx, 0)
        bg.solve()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.ba_steps):
            bg.solve()
        sync()
        dt_gemm = time.perf_counter() - t0
        bg.set_profiling(True)
        bg.solve()
        sync()
        gemm_ms, gemm_n, gemm_fl = bg.gemm_profile()
        gemm_dense = bg.gemm_dense_flops()
        gemm_issued = bg.gemm_issued_flops()
        bg.set_profiling(False)
        # the MFMA form must reproduce the default form's solve: same LM decisions on every graph, estimates to rounding
        poses_g, points_g, _, stats_g = bg.download()
        bg.close()
        trials_equal = all(a["lm_trials"] == b["lm_trials"] and a["iterations_run"] == b["iterations_run"] for a, b in zip(stats, stats_g))
        pose_rmse = max(float(np.sqrt(np.mean((a - b) ** 2))) for a, b in zip(poses_def, poses_g))
        point_rmse = max(float(np.sqrt(np.mean((a - b) ** 2))) for a, b in zip(points_def, points_g))
        assert trials_equal and pose_rmse <= 1e-9 and point_rmse <= 1e-9, ("MFMA Schur variant diverges from the default", trials_equal, pose_rmse, point_rmse)
        peak64 = orbhip.mfma_f64_peak_tflops(ctx)
        tfl = gemm_fl * gemm_n / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        # sparse-exact flops of the Schur complement (what g2o's per-block products do): per point with k free observers
        # k(k+1)/2 products 6x3 . 3x6 (2*6*3*6 flop) + k products 6x3 . 3x3 (2*6*3*3); algorithmic bytes of the pair kernel:
        # every Hpl block (144 B) and C^-1 (48 B per point) read once, the reduced matrix written once
        useful, blocks, pts_n, s_bytes, pair_entries = 0.0, 0.0, 0.0, 0.0, 0.0
        for g in glist:
            free = 1 - np.asarray(g["pose_fixed"], np.int64)
            kfree = np.bincount(np.asarray(g["edge_point"]), weights=free[np.asarray(g["edge_pose"])], minlength=g["n_points"])
            useful += float(np.sum(kfree * (kfree + 1) / 2 * 216 + kfree * 108))
            pair_entries += float(np.sum(kfree * (kfree + 1) / 2))
            blocks += float(kfree.sum()); pts_n += g["n_points"]; s_bytes += 8.0 * (6 * int(free.sum())) ** 2
        alg_bytes = blocks * 144 + pts_n * 48 + s_bytes
        launch_ms = schur_ms / max(schur_n, 1)
        l2_bytes = (pair_entries - blocks) * (144 + 8) + blocks * (144 + 48 + 24 + 12)
        ach = alg_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        ba_traffic, ba_traffic_src = None, None
        pmc_ba = load_profile_json(PROFILE_TAG + "_pmc_traffic_ba.json")
        big = [v for k, v in (pmc_ba or {}).get("kernels", {}).items() if k == "k_ba_schur_rows"]
        if big and args.ba_graphs == 256:
            ba_traffic = big[0]["hbm_bytes_per_launch"]
            ba_traffic_src = "profiles/%s_pmc_traffic_ba.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch)" % PROFILE_TAG
        ba = {"metric": "local-BA solves/sec", "value": round(world * args.ba_graphs * args.ba_steps / dt_ba, 2),
              "unit": "solves/s", "graphs_per_gpu": args.ba_graphs, "ms_per_batch": round(dt_ba / args.ba_steps * 1e3, 2),
              "lm_ticks": ticks, "workload": "50 KF (2 fixed) x 2000 points x 10 obs, 5+10 LM iterations, Huber, Schur",
              "lm_trials_graph0": stats[0]["lm_trials"], "dtype": "f64",
              # the fraction printed is against what bounds the kernel.  Its counters (profiles/*_pmc_traffic_ba.json)
              # show HBM traffic at 1.4x the algorithmic bytes and ~0.12 of the HBM rate: it is bound by the gather stream its pair
              # lists pull out of the XCD's L2 (every Hpl block is read once per pair it belongs to), so `achieved` = those bytes per
              # launch / the launch time against the aggregate L2 rate of MI355X_MICROARCH.md (34.5 TB/s); the HBM figures stay beside it
              "roofline": {"bound": "l2", "kernel": "k_ba_schur_rows", "achieved": round(l2_bytes / (launch_ms * 1e-3) / 1e9, 1) if launch_ms > 0 else 0.0,
                           "peak": L2_PEAK_GBS, "unit": "GB/s",
                           "frac": round(l2_bytes / (launch_ms * 1e-3) / 1e9 / L2_PEAK_GBS, 4) if launch_ms > 0 else 0.0,
                           "peak_source": "MI355X_MICROARCH.md, L2 (per XCD): 4 MiB x 8, ~34.5 TB/s aggregate",
                           "hbm": {"achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                   "peak_measured": HBM_MEASURED_GBS, "frac_of_measured_peak": round(ach / HBM_MEASURED_GBS, 4)},
                           "traffic": ba_traffic, "traffic_source": ba_traffic_src,
                           "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(launch_ms, 4),
                           "sparse_exact_flops_per_launch": useful,
                           "achieved_tflops_of_useful_flops": round(useful / (launch_ms * 1e-3) / 1e12, 2) if launch_ms > 0 else None,
                           "l2_gather_bytes_per_launch": int(l2_bytes),
                           "note": "the Schur complement from per-block-pair lists, one 384-thread workgroup per ROW of the block matrix: pose i's "
                                   "W D^-1 blocks are computed once into LDS (with the diagonal block and W D^-1 b on the way), every (i, j > i, shared "
                                   "point) entry then gathers ONE 144-byte Hpl block + an 8-byte list entry, one 16-lane row per pair, a graph's blocks "
                                   "pinned to one XCD's L2; bound by the L2 gather stream (l2_gather_bytes_per_launch), not by HBM.  Round 2's "
                                   "k_ba_schur_big (both blocks + C^-1 gathered per entry: 1.61 ms per 256 windows) remains for rows of more than "
                                   "1024 blocks and batches of fewer than 8 windows"},
              "mfma_gemm_variant": {"value": round(world * args.ba_graphs * args.ba_steps / dt_gemm, 2), "unit": "solves/s",
                                    "ms_per_batch": round(dt_gemm / args.ba_steps * 1e3, 2),
                                    "roofline": {"bound": "mfma", "kernel": "k_ba_schur_gemm", "achieved": round(tfl, 2),
                                                 "peak": round(MFMA_F64_PEAK_TFLOPS, 1), "unit": "TFLOP/s", "frac": round(tfl / MFMA_F64_PEAK_TFLOPS, 4),
                                                 "peak_source": "v_mfma_f64_16x16x4_f64: 2048 flop / 64 cycles x 1024 SIMDs x 2.4 GHz max clock (fixed)",
                                                 "peak_measured_on_device": round(peak64, 2),
                                                 "frac_of_measured_peak": round(tfl / peak64, 4) if peak64 else None,
                                                 "useful_flop_frac": round(useful / (gemm_fl * 1.0), 4) if gemm_fl else None,
                                                 "avg_launch_ms": round(gemm_ms / max(gemm_n, 1), 4),
                                                 "mfma_flops_of_data_tiles_per_launch": gemm_fl, "mfma_flops_issued_per_launch": gemm_issued,
                                                 "flops_per_launch_without_sparsity_skipping": gemm_dense},
                                    "parity_vs_default": {"lm_trials_and_iterations_equal_on_all_graphs": bool(trials_equal), "max_pose_rmse": pose_rmse,
                                                          "max_point_rmse": point_rmse, "asserted": "<= 1e-9; both forms are oracle-checked at this size in "
                                                          "tests/test_gpu_ba.py::test_ba_full_size_mono[pairs|mfma]"},
                                    "note": "orbhip_ctx_set_ba_schur_mode(ctx, 2): S = Z^T Z on v_mfma_f64_16x16x4_f64 with block-sparsity skipping (the round-1 "
                                            "design); 16x16 tiles of 6-row blocks are mostly zeros, so it issues ~9x the useful flops"}}

    # ---- landmark-sharded single-graph mode (optional): the SAME graphs solved by all ranks together, the shared Schur
    # block all-gathered every LM trial (RCCL over xGMI with backend nccl).  Latency-bound by design; reported, not hidden.
    ba_sh = None
    if world > 1 and args.ba_graphs > 0 and args.ba_sharded_graphs > 0:
        import shard
        import synth_ba

        def all_ok(ok):
            """agree on success across ranks before the next collective phase: nobody waits for a rank that failed"""
            t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return int(t.item()) == 0
        sb, err = None, None
        try:
            gs = [synth_ba.make_graph(seed=9000 + i) for i in range(min(args.ba_sharded_graphs, 4))]
            gl = [gs[i % len(gs)] for i in range(args.ba_sharded_graphs)]
            sb = orbhip.BaBatch(ctx, gl, rank=rank, world=world)
            stride = sb.exchange_doubles
            xbuf = torch.zeros(world * stride, dtype=torch.float64, device="cuda")
            sb.set_exchange_buffer(xbuf.data_ptr(), world * stride)
            n_x = [0]
            gather = shard.make_ba_exchange(xbuf, stride)

            def xch(stage, count):
                n_x[0] += 1
                gather(stage, count)
        except Exception as e:
            err = repr(e)[:300]
        if not all_ok(err is None):
            ba_sh = {"error": err or "another rank failed to set up the sharded batch"}
        else:
            # from here on a failing rank would leave the others inside an all-gather: the process group's timeout turns that
            # into an error; the main line has been computed already and is printed regardless
            try:
                sb.solve_sharded(xch)                        # warm-up
                n_x[0] = 0
                barrier(); sync()
                t0 = time.perf_counter()
                sb.solve_sharded(xch)
                sync(); barrier()
                (dt_sh,) = max_over_ranks(time.perf_counter() - t0)
                st_sh = sb.download()[3]
                ba_sh = {"metric": "local-BA solves/sec, points sharded over all GPUs (one all-gather of the Schur block per LM trial)",
                         "value": round(args.ba_sharded_graphs / dt_sh, 2), "unit": "solves/s", "graphs": args.ba_sharded_graphs,
                         "ranks": world, "all_gathers": n_x[0], "doubles_per_rank_and_gather": int(stride),
                         "lm_trials_graph0": st_sh[0]["lm_trials"], "backend": backend}
            except Exception as e:
                ba_sh = {"error": repr(e)[:300]}
        if sb is not None:
            sb.close()

    pose = None
    pose_probs = None
    if args.pose_frames > 0:
        import synth_ba
        pose_probs = [synth_ba.make_pose_problem(7000 + 16 * rank + k, n=1000, stereo_frac=0.25 * (k % 4), outlier_frac=0.1)
                      for k in range(16)]
        F, M = args.pose_frames, 1000
        hx = np.stack([pose_probs[k % 16]["Xw"] for k in range(F)]); ho = np.stack([pose_probs[k % 16]["obs"] for k in range(F)])
        hw = np.stack([pose_probs[k % 16]["inv_sigma2"] for k in range(F)]); hp = np.stack([pose_probs[k % 16]["pose0"] for k in range(F)])
        dx, do_, dw = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (hx, ho, hw))
        dn = torch.full((F,), M, dtype=torch.int32, device="cuda")
        dps = [torch.from_numpy(hp).cuda() for _ in range(args.ba_steps + 1)]     # initial poses (in/out), one set per launch
        dout = torch.zeros((F, M), dtype=torch.uint8, device="cuda"); dni = torch.zeros((F,), dtype=torch.int32, device="cuda")

        def pose_step(dp):
            orbhip.pose_optimization_device(ctx, dx.data_ptr(), do_.data_ptr(), dw.data_ptr(), dn.data_ptr(), F, M,
                                            pose_probs[0]["cam"], dp.data_ptr(), dout.data_ptr(), dni.data_ptr())
        sync(); pose_step(dps[-1]); sync()
        barrier()
        t0 = time.perf_counter()
        for i in range(args.ba_steps):
            pose_step(dps[i])
        sync()
        barrier()
        (dt_po,) = max_over_ranks(time.perf_counter() - t0)
        pose = {"metric": "pose-only BA frames/sec", "value": round(world * F * args.ba_steps / dt_po, 1), "unit": "frames/s",
                "frames_per_gpu": F, "ms_per_batch": round(dt_po / args.ba_steps * 1e3, 3), "dtype": "f64",
                "workload": "Optimizer::PoseOptimization: 1000 unary edges/frame (0-75 % stereo), 10 % gross outliers, 4 rounds x 10 LM its",
                "mean_inliers": round(float(dni.float().mean().item()), 1)}

    inertial = None
    iba_wins = None
    if args.inertial_windows > 0:
        import synth_iba
        iba_wins = [synth_iba.make_window(9100 + 8 * rank + k, n_opt=10, n_fixed_vis=20, n_points=600) for k in range(8)]
        NW = args.inertial_windows
        structs = [iba_wins[i % 8].struct(orbhip.IbaWindow) for i in range(NW)]
        kfs0 = [iba_wins[i % 8].kf0 for i in range(NW)]
        pts0 = [iba_wins[i % 8].pts0 for i in range(NW)]
        ip = orbhip.iba_default_params(False)
        orbhip.inertial_ba_solve_batch(ctx, structs, kfs0, pts0, ip)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.ba_steps):
            iba_res = orbhip.inertial_ba_solve_batch(ctx, structs, kfs0, pts0, ip)
        barrier()
        (dt_ib,) = max_over_ranks(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(args.ba_steps):
            orbhip.inertial_ba_solve_batch(ctx, structs[:1], kfs0[:1], pts0[:1], ip)
        dt_one = (time.perf_counter() - t0) / args.ba_steps
        # the same windows as a resident batch (orbhip_iba_batch_*): packed and uploaded once, a solve uploads the states and launches
        ibatch = orbhip.IbaBatch(ctx, structs, kfs0, pts0)
        ibatch.solve(ip)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.ba_steps):
            ibatch.solve(ip)
        barrier()
        (dt_res,) = max_over_ranks(time.perf_counter() - t0)
        res_dl = ibatch.download()
        assert all(np.array_equal(a, b) for a, b in zip(res_dl[0], iba_res[0])), "resident inertial batch differs from the one-shot call"
        ibatch.close()
        ione = orbhip.IbaBatch(ctx, structs[:1], kfs0[:1], pts0[:1])
        ione.solve(ip)
        t0 = time.perf_counter()
        for _ in range(args.ba_steps):
            ione.solve(ip)
        dt_one_res = (time.perf_counter() - t0) / args.ba_steps
        ione.close()
        inertial = {"metric": "inertial local-BA windows/sec", "value": round(world * NW * args.ba_steps / dt_ib, 1), "unit": "windows/s",
                    "windows_per_gpu": NW, "ms_per_batch": round(dt_ib / args.ba_steps * 1e3, 3), "single_window_ms": round(dt_one * 1e3, 3),
                    "resident": {"value": round(world * NW * args.ba_steps / dt_res, 1), "unit": "windows/s", "ms_per_batch": round(dt_res / args.ba_steps * 1e3, 3),
                                 "single_window_ms": round(dt_one_res * 1e3, 3),
                                 "what": "orbhip_iba_batch_*: windows packed and uploaded once; a solve = state upload + kernel (no download)"},
                    "dtype": "f64", "lm_trials_window0": iba_res[3][0]["lm_trials"],
                    "workload": "Optimizer::LocalInertialBA: 10 IMU keyframes (15 unknowns each) + 1 fixed IMU keyframe + 20 fixed visual "
                                "keyframes, 600 landmarks (%d visual edges), optimize(10), host arrays in / out (packing + H2D + kernel + D2H)"
                                % iba_wins[0].n_edges}

    # ---- the tracking matchers on the bench batch (ORBmatcher::SearchByProjection, last frame and local map): every frame's keypoints
    # as projection queries against its successor; timed on their own, not part of `value`
    tracking = None
    if args.tracking and B > 1:
        import ctypes as C
        hipl = C.CDLL("libamdhip64.so"); hipl.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        extract(); sync()
        kp_h = np.zeros((B, max_kp), orbhip.KP_DTYPE)
        assert hipl.hipMemcpy(kp_h.ctypes.data, kp_p, kp_h.nbytes, 2) == 0
        desc_h = np.zeros((B, max_kp, 32), np.uint8)
        assert hipl.hipMemcpy(desc_h.ctypes.data, desc_p, desc_h.nbytes, 2) == 0
        cnt_h = d_cnt.cpu().numpy()
        sf = ext.table(0)
        q = np.zeros((B, max_kp), orbhip.PROJ_QUERY_DTYPE)
        q["u"] = kp_h["x"]; q["v"] = kp_h["y"]; q["angle"] = kp_h["angle"]; q["radius"] = np.float32(15.0) * sf[np.clip(kp_h["octave"], 0, 7)]
        q["min_level"] = kp_h["octave"] - 1; q["max_level"] = kp_h["octave"] + 1; q["has_obs"] = 1; q["ur"] = -1
        d_q = torch.from_numpy(q.view(np.uint8)).cuda()
        d_tm = torch.full((B, max_kp), -1, dtype=torch.int32, device="cuda"); d_tn = torch.zeros((B,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        bnds = (0.0, 0.0, float(W), float(H))
        legs = {"search_by_projection_last_frame": lambda: orbhip.search_by_projection_device(
                    ctx, d_q.data_ptr(), desc_p, cnt_p, max_kp, kp_p + max_kp * 28, desc_p + dstride, None, cnt_p + 4, max_kp, max_kp, B - 1, bnds,
                    100, True, d_tm.data_ptr(), d_tn.data_ptr()),
                "search_by_projection_local_map": lambda: orbhip.search_local_map_device(
                    ctx, d_q.data_ptr(), desc_p, cnt_p, max_kp, kp_p + max_kp * 28, desc_p + dstride, None, cnt_p + 4, max_kp, max_kp, B - 1, bnds,
                    100, 0.8, d_tm.data_ptr(), d_tn.data_ptr())}
        tracking = {"workload": "%d consecutive frame pairs of the bench batch, every keypoint of frame i projected into frame i+1 (radius 15 x scale, "
                                "levels octave-1..octave+1), ORBmatcher.cc:1965 / :48" % (B - 1)}
        for name, fn in legs.items():
            dt_t = 0.0
            for it in range(5):                                  # d_tm is in/out (claimed train keypoints are skipped): reset before every call
                d_tm.fill_(-1); torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn(); sync()
                if it:
                    dt_t += (time.perf_counter() - t0) / 4
            (dt_t,) = max_over_ranks(dt_t)
            tracking[name] = {"ms_per_batch": round(dt_t * 1e3, 3), "frame_pairs_per_s": round(world * (B - 1) / dt_t, 1),
                              "matches_per_pair": round(float(d_tn[:B - 1].float().mean().item()), 1)}
        # the same two calls with ONE frame pair, each waited for: what Tracking::TrackWithMotionModel / SearchLocalPoints feel
        one_pair = {"search_by_projection_last_frame": lambda: orbhip.search_by_projection_device(
                        ctx, d_q.data_ptr(), desc_p, cnt_p, max_kp, kp_p + max_kp * 28, desc_p + dstride, None, cnt_p + 4, max_kp, max_kp, 1, bnds,
                        100, True, d_tm.data_ptr(), d_tn.data_ptr()),
                    "search_by_projection_local_map": lambda: orbhip.search_local_map_device(
                        ctx, d_q.data_ptr(), desc_p, cnt_p, max_kp, kp_p + max_kp * 28, desc_p + dstride, None, cnt_p + 4, max_kp, max_kp, 1, bnds,
                        100, 0.8, d_tm.data_ptr(), d_tn.data_ptr())}
        for name, fn in one_pair.items():
            dt_t = 0.0
            for it in range(21):
                d_tm[0].fill_(-1); torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn(); ctx.synchronize()
                if it:
                    dt_t += (time.perf_counter() - t0) / 20
            tracking[name]["one_pair_ms"] = round(dt_t * 1e3, 4)

    stereo = None
    stereo_host = None
    if args.stereo_pairs > 0:
        S = args.stereo_pairs
        # a different set of pairs every step (frames change between calls, as in a replay): the right extractor's next
        # extraction must not overwrite what the previous step's stereo kernels still read
        nset = 2
        sets = []
        for k in range(nset):
            big = synth_frames_parallel(orbhip, W + 64, H, S, 777 + 100000 * rank + 31 * k, 0)
            disp = [4 + (7 * j + 3 * k) % 40 for j in range(S)]
            lefts = np.ascontiguousarray(big[:, :, 0:W])
            rights = np.stack([big[j, :, disp[j]:disp[j] + W] for j in range(S)])
            sets.append((torch.from_numpy(lefts).cuda(), torch.from_numpy(np.ascontiguousarray(rights)).cuda()))
            if k == 0:
                stereo_host = (lefts[:min(S, 48)].copy(), np.ascontiguousarray(rights[:min(S, 48)]))
        ctx_r = orbhip.Context(local_rank)
        ext_l = orbhip.Extractor(ctx, args.nfeatures, 1.2, 8, 20, 7); ext_r = orbhip.Extractor(ctx_r, args.nfeatures, 1.2, 8, 20, 7)
        ext_l.reserve(W, H, S); ext_r.reserve(W, H, S)
        mk = ext_l.max_keypoints
        d_ur = torch.empty((S, mk), dtype=torch.float32, device="cuda"); d_dp = torch.empty((S, mk), dtype=torch.float32, device="cuda")
        d_nk = torch.zeros((S,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()

        def stereo_step(k):
            # the two extractors run on their own streams (the stereo constructor's two threads, Frame.cc:109-110)
            d_l, d_r = sets[k % nset]
            ext_l.extract_device(d_l.data_ptr(), W, H, W, W * H, S, (0, 0))
            ext_r.extract_device(d_r.data_ptr(), W, H, W, W * H, S, (0, 0))
            orbhip.compute_stereo_matches_device(ext_l, ext_r, 40.0 / 458.0, 40.0, d_ur.data_ptr(), d_dp.data_ptr(), d_nk.data_ptr())
        stereo_step(0); ctx_r.synchronize(); sync()
        barrier()
        t0 = time.perf_counter()
        for k in range(args.ba_steps):
            stereo_step(k + 1)
        ctx_r.synchronize(); sync()
        t_all = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(args.ba_steps):
            orbhip.compute_stereo_matches_device(ext_l, ext_r, 40.0 / 458.0, 40.0, d_ur.data_ptr(), d_dp.data_ptr(), d_nk.data_ptr())
        ctx_r.synchronize(); sync()
        t_match = time.perf_counter() - t0
        barrier()
        t_all, t_match = max_over_ranks(t_all, t_match)
        stereo = {"metric": "stereo frame pairs/sec (2 x ORB extract + ComputeStereoMatches)", "value": round(world * S * args.ba_steps / t_all, 1),
                  "unit": "pairs/s", "pairs_per_gpu": S, "ms_per_batch": round(t_all / args.ba_steps * 1e3, 3),
                  "compute_stereo_matches_ms_per_batch": round(t_match / args.ba_steps * 1e3, 3),
                  "mean_stereo_matches_per_pair": round(float(d_nk.float().mean().item()), 1),
                  "workload": "synthetic rectified %dx%d pairs, disparity 4..43 px, %d feats, %d alternating sets" % (W, H, args.nfeatures, nset)}
        ext_l.close(); ext_r.close(); ctx_r.close()

    # ---- per-call latencies at batch 1 (what a drop-in caller feels): each call waited for, the oracle timed single-threaded beside it
    latency = None
    if args.latency and world == 1:
        latency = {"note": "one call of each entry point with ONE unit of work, waited for (hipStreamSynchronize) -- not part of `value`; "
                           "cpu_oracle_ms_1thread = the CPU oracle on one host thread (filled in when the CPU baselines run)"}
        ext1 = orbhip.Extractor(ctx, args.nfeatures, 1.2, 8, 20, 7)
        ext1.reserve(W, H, 1)
        for _ in range(20):
            ext1.extract_device(d_imgs.data_ptr(), W, H, W, W * H, 1, (0, 0))
        ctx.synchronize()
        nrep = 200
        t0 = time.perf_counter()
        for _ in range(nrep):
            ext1.extract_device(d_imgs.data_ptr(), W, H, W, W * H, 1, (0, 0))
            ctx.synchronize()
        latency["extract_one_frame"] = {"gpu_ms": round((time.perf_counter() - t0) / nrep * 1e3, 4), "what": "ORBextractor::operator() on one %dx%d frame resident in HBM" % (W, H)}
        ext1.close()
        if B > 1:
            # one frame pair of SearchForInitialization (what Tracking::MonocularInitialization calls), on the bench batch's frames 0 / 1
            def si_one():
                orbhip.prev_matched_init_device(ctx, kp_p, max_kp, 1, max_kp, d_prev.data_ptr())
                orbhip.search_for_initialization_device(ctx, kp_p, desc_p, cnt_p, kp_p + max_kp * 28, desc_p + dstride, cnt_p + 4, 1, max_kp, max_kp,
                                                        (0.0, 0.0, float(W), float(H)), 100, 0.9, True, d_prev.data_ptr(), d_m12.data_ptr(), d_nm.data_ptr())
            extract(); sync()
            for _ in range(5):
                si_one()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                si_one()
                ctx.synchronize()
            latency["search_for_initialization_one_pair"] = {"gpu_ms": round((time.perf_counter() - t0) / 50 * 1e3, 4),
                                                              "what": "vbPrevMatched reset + ORBmatcher::SearchForInitialization of frames 0 / 1 of the batch (window 100, ratio 0.9)"}
        if tracking is not None:
            for name in ("search_by_projection_last_frame", "search_by_projection_local_map"):
                latency[name + "_one_pair"] = {"gpu_ms": tracking[name]["one_pair_ms"], "what": "~%d queries against one frame" % int(cnt_h[0])}
        if pose_probs is not None:
            dp1 = torch.from_numpy(hp[:1].copy()).cuda()
            tl = 0.0
            for it in range(21):
                dp1.copy_(torch.from_numpy(hp[:1])); torch.cuda.synchronize()
                t0 = time.perf_counter()
                orbhip.pose_optimization_device(ctx, dx.data_ptr(), do_.data_ptr(), dw.data_ptr(), dn.data_ptr(), 1, M, pose_probs[0]["cam"],
                                                dp1.data_ptr(), dout.data_ptr(), dni.data_ptr())
                ctx.synchronize()
                if it:
                    tl += (time.perf_counter() - t0) / 20
            latency["pose_optimization_one_frame"] = {"gpu_ms": round(tl * 1e3, 4), "what": "1000 unary edges, 4 x 10 LM iterations"}
        if graphs is not None:
            b1 = orbhip.BaBatch(ctx, graphs[:1])
            b1.solve(); sync()
            t0 = time.perf_counter()
            for _ in range(5):
                b1.solve(); sync()
            latency["local_ba_one_window"] = {"gpu_ms": round((time.perf_counter() - t0) / 5 * 1e3, 3), "what": "50 KF x 2000 points x 10 obs, graph resident (orbhip_ba_batch_solve)"}
            b1.close()
        if inertial is not None:
            latency["inertial_ba_one_window"] = {"gpu_ms": inertial["single_window_ms"], "what": "10 IMU keyframes, 600 landmarks, host arrays in / out"}
        # ---- the signature-preserving C++ classes themselves: wall time per call with host cv::Mat / std::vector in and
        # results out -- packing, the PCIe copies, the kernels, unpacking -- measured by lib/host_smoke (host/host_latency.cc) in its own process
        # while this one is idle; `device_resident_twin_ms` = the C-ABI call on data already in HBM from the entries above (same shapes, not
        # the same bytes), `ratio` = host class / twin
        torch.cuda.synchronize()
        exe = os.path.join(ROOT, "orb-slam3-mac_amd", "lib", "host_smoke")
        try:
            hc = subprocess.run([exe, "latency", "40"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            host_classes = json.loads(hc.stdout) if hc.returncode == 0 else {"error": "host_smoke latency: rc %d: %s" % (hc.returncode, hc.stderr[-400:])}
        except Exception as e:               # the line must still print
            host_classes = {"error": repr(e)}
        twins = {"extract_one_frame": "extract_one_frame", "search_by_projection_last_frame": "search_by_projection_last_frame_one_pair",
                 "search_by_projection_local_map": "search_by_projection_local_map_one_pair", "pose_optimization_one_frame": "pose_optimization_one_frame",
                 "local_ba_one_window": "local_ba_one_window"}
        for k, t in twins.items():
            if k in host_classes and t in latency:
                host_classes[k]["device_resident_twin_ms"] = latency[t]["gpu_ms"]
                host_classes[k]["ratio"] = round(host_classes[k]["host_class_ms"] / latency[t]["gpu_ms"], 3) if latency[t]["gpu_ms"] else None
        host_classes["note"] = ("median wall time per call of ORBextractor::operator(), ORBmatcher::SearchByProjection (both overloads), Optimizer::PoseOptimization(Frame*) "
                                "and Optimizer::LocalBundleAdjustment(KeyFrame*, ...) through orb-slam3-mac_amd/host/ (one page-locked blob in, one out; the frame just "
                                "extracted stays resident for the matchers; mvImagePyramid materialised on first read)")
        latency["host_classes"] = host_classes

    # ---- BASELINE config #3's per-GPU shard beside the default line: 1920x1080, 2000 features, 512 frames, same step
    # ---- the other image sizes north_star names, beside the default line: the same step (extract + 2-NN + SearchForInitialization) on
    # 1920x1080 (BASELINE config #3's per-GPU shard) and on 3840x2160 frames; sampled frames are checked against the oracle bit for bit
    # when the CPU baselines run
    def big_leg(tag, HW, HH, HF, HB, HS, seed, what):
        import hashlib
        h_imgs = synth_frames_parallel(orbhip, HW, HH, HB, seed, 0)
        dh = torch.from_numpy(h_imgs).cuda()
        exh = orbhip.Extractor(ctx, HF, 1.2, 8, 20, 7)
        exh.reserve(HW, HH, HB)
        mk = exh.max_keypoints
        h_idx2 = torch.empty((HB, mk, 2), dtype=torch.int32, device="cuda"); h_dist2 = torch.empty((HB, mk, 2), dtype=torch.int32, device="cuda")
        h_acc = torch.zeros((HB, mk), dtype=torch.uint8, device="cuda"); h_prev = torch.zeros((HB, mk, 2), dtype=torch.float32, device="cuda")
        h_m12 = torch.empty((HB, mk), dtype=torch.int32, device="cuda"); h_nm = torch.zeros((HB,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        hk, hd_, hc, _ = exh.results_device()
        hs = mk * 32

        cw = ctx2 or ctx                                          # the windowed matcher's context (as in the main line's step)

        def leg_step():
            if ctx2 is not None:
                ctx.wait_for(ctx2)
            exh.extract_device(dh.data_ptr(), HW, HH, HW, HW * HH, HB, (0, 0))
            if ctx2 is not None:
                ctx2.wait_for(ctx)
            orbhip.match_bf2nn_device(ctx, hd_, hc, hs, hd_ + hs, hc + 4, hs, HB - 1, mk, 0.7, h_idx2.data_ptr(), h_dist2.data_ptr(), h_acc.data_ptr())
            orbhip.match_bf2nn_device(ctx, hd_ + (HB - 1) * hs, hc + 4 * (HB - 1), hs, hd_, hc, hs, 1, mk, 0.7, h_idx2.data_ptr() + (HB - 1) * mk * 8,
                                      h_dist2.data_ptr() + (HB - 1) * mk * 8, h_acc.data_ptr() + (HB - 1) * mk)
            orbhip.prev_matched_init_device(cw, hk, mk, HB - 1, mk, h_prev.data_ptr())
            orbhip.search_for_initialization_device(cw, hk, hd_, hc, hk + mk * 28, hd_ + hs, hc + 4, HB - 1, mk, mk, (0.0, 0.0, float(HW), float(HH)), 100,
                                                    0.9, True, h_prev.data_ptr(), h_m12.data_ptr(), h_nm.data_ptr())
        leg_step(); sync()
        t0 = time.perf_counter()
        for _ in range(HS):
            leg_step()
        sync()
        dt_leg = time.perf_counter() - t0
        ctx.check_status()
        if ctx2 is not None:
            ctx2.check_status()
        # digest of sampled frames (keypoint records + descriptors as the C ABI returns them) and the oracle's digest of the same frames
        sample = sorted({0, HB // 2, HB - 1})
        got = exh.extract_host(h_imgs[sample], (0, 0))
        dig = hashlib.sha256()
        for gk, gd, gm in got:
            dig.update(gk.tobytes()); dig.update(gd.tobytes())
        leg = {"metric": "ORB extract+match frames/sec", "value": round(HB * HS / dt_leg, 1), "unit": "frames/s", "steps": HS, "ms_per_step": round(dt_leg / HS * 1e3, 3),
               "config": {"workload": what % HB}, "nfeatures": HF,
               "keypoints_sampled_frames": [int(len(g[0])) for g in got], "sampled_frames": sample,
               "sha256_keypoints_and_descriptors_of_sampled_frames": dig.hexdigest(),
               "algorithmic_GBps": round(algorithmic_bytes(HW, HH, float(np.mean([len(g[0]) for g in got])), 0)["total"] * HB * HS / dt_leg / 1e9, 1)}
        sample_imgs = h_imgs[sample].copy()
        exh.close()
        del dh, h_imgs
        torch.cuda.empty_cache()
        return leg, sample_imgs

    hd, uhd, hd_sample_imgs, uhd_sample_imgs = None, None, None, None
    if args.hd_leg and (W, H) == (640, 480) and world == 1:
        hd, hd_sample_imgs = big_leg("hd", 1920, 1080, 2000, args.hd_frames, 3, 20241004,
                                     "hd: synthetic 1920x1080 batch=%d per GPU (BASELINE config #3's shard of 4096 frames over 8 GPUs), 8-level pyramid, 2000 feats/frame, "
                                     "same step as the main line")
    if args.uhd_leg and (W, H) == (640, 480) and world == 1:
        uhd, uhd_sample_imgs = big_leg("4k", 3840, 2160, 2000, args.uhd_frames, 3, 20241005,
                                       "4k: synthetic 3840x2160 batch=%d per GPU, 8-level pyramid, 2000 feats/frame, same step as the main line (north_star: VGA / HD / 4K batches)")

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        fps = world * B * args.steps / dt
        ab = algorithmic_bytes(W, H, n_kp_avg, n_cand_avg)
        # the dominant kernel of the step: largest device time among ALL of its kernels (separate profiled pass).
        # one stage == one kernel (SURVEY's "FAST read S" = k_fast_cells, "blur read+write 2S" = k_blur)
        kern = {"pyramid": "k_resize_rows", "blur": ext.blur_kernel(B), "fast_cells": "k_fast_cells", "octree": "k_octree",
                "desc": "k_orient_desc", "assemble": "k_assemble", "match_bf2nn": "k_bf2nn_mfma", "search_init": "k_search_init"}

        def profile_entry(table, name):                     # template instances are listed as "k_fast_cells<3, 24, 2>"
            for k, v in (table or {}).items():
                if k == name or k.startswith(name + "<"):
                    return v
            return None
        cand = [k for k in kern if k in ab]                  # kernels priced in bytes (the matchers are lane-op work: listed in stage_ms)
        dom = max(cand, key=lambda k: stage.get(k, 0.0))
        launches = {"pyramid": 7}.get(dom, 1)
        dom_bytes = ab[dom] * B
        achieved = dom_bytes / (stage[dom] * 1e-3) / 1e9 if stage[dom] > 0 else 0.0
        # HBM traffic / vector-issue occupancy of the dominant kernel: PMC counters are collected in separate rocprofv3 passes
        # (they cannot be read from inside this process); the committed pass is reported when the workload matches.
        valu = None
        pv = load_profile_json(PROFILE_TAG + "_pmc_valu_issue.json") or load_profile_json("r01_pmc_valu_issue.json")
        if pv and (B, W, H) == (1024, 640, 480) and profile_entry(pv.get("kernels"), kern[dom]):
            kv = profile_entry(pv.get("kernels"), kern[dom])
            insts = kv.get("SQ_INSTS_VALU_per_step")
            valu = {"valu_issue_busy_frac": kv.get("valu_issue_busy_frac"), "source": "profiles/%s (rocprofv3 --pmc: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs))" % pv.get("file", "*_pmc_valu_issue.json")}
            if insts:
                lane_ops_px = insts * 64.0 / (ab["S"] * B)
                floor_ms = insts * 64.0 / VALU_LANES_PER_CLK / (MAX_CLOCK_GHZ * 1e9) * 1e3
                valu.update({"lane_ops_per_pixel": round(lane_ops_px, 1), "issue_slot_floor_ms_at_2.4GHz": round(floor_ms, 4),
                             "frac_of_issue_slot_roofline": round(floor_ms / stage[dom], 4) if stage[dom] > 0 else None})
        traffic, traffic_src = None, None
        pmc = load_profile_json(PROFILE_TAG + "_pmc_traffic.json") or load_profile_json("r01_pmc_traffic.json")
        if pmc and (W, H, B, args.nfeatures) == (640, 480, 1024, 1000) and profile_entry(pmc.get("kernels"), kern[dom]):
            traffic = profile_entry(pmc.get("kernels"), kern[dom])["hbm_bytes_per_step"]
            traffic_src = "profiles/*_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per step)"
        all_k = {}
        for kk in cand:
            if stage.get(kk, 0.0) <= 0:
                continue
            ent = {"kernel": kern[kk], "ms": round(stage[kk], 4), "algorithmic_gbs": round(ab[kk] * B / (stage[kk] * 1e-3) / 1e9, 1),
                   "frac_of_8tbs": round(ab[kk] * B / (stage[kk] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            pe = profile_entry(pmc.get("kernels"), kern[kk]) if (pmc and (W, H, B, args.nfeatures) == (640, 480, 1024, 1000)) else None
            if pe:
                ent["traffic_bytes_per_step"] = pe["hbm_bytes_per_step"]
                ent["traffic_gbs"] = round(pe["hbm_bytes_per_step"] / (stage[kk] * 1e-3) / 1e9, 1)
                ent["traffic_frac_of_copy_rate"] = round(pe["hbm_bytes_per_step"] / (stage[kk] * 1e-3) / 1e9 / 4900.0, 4)
            all_k[kk] = ent
        out = {
            "metric": "ORB extract+match frames/sec", "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%s: synthetic %dx%d batch=%d per GPU, 8-level pyramid, %d feats/frame, "
                                   "ORB extract + Hamming 2-NN match (Frame.cc:1146) + SearchForInitialization "
                                   "(ORBmatcher.cc:710) vs successor frame" % (args.workload, W, H, B, args.nfeatures),
                       "pipelines": len(steppers),
                       "pipelines_note": "1 = strictly serial steps; inside a step the blur runs beside k_fast_cells on a second HIP stream"
                                         + ("" if args.serial_matchers else " and SearchForInitialization beside the 2-NN matcher on a second context "
                                            "(ordered after the extraction, the next extraction ordered after it)")
                                         + "; stage_ms and the roofline come from the separate profiled pass, one kernel at a time",
                       "matchers": "serial" if args.serial_matchers else "side by side (two contexts)",
                       "frames_total": world * B, "ranks_seen": ranks_seen, "collective_backend": backend if distributed else None,
                       "records_gathered": records_gathered,
                       "keypoints_per_frame": round(n_kp_avg, 1), "fast_candidates_per_frame": round(n_cand_avg, 1),
                       "bf_ratio_matches_per_frame": round(bf_accept, 1),
                       "windowed_matches_per_pair": round(win_matches, 1),
                       "stage_ms": {k: round(v, 4) for k, v in stage.items()},
                       "stage_ms_source": "separate pass of %d steps with hipEvents on the library's stream (%.3f ms per step, extract %.3f ms); "
                                          "the timed region runs with stage profiling off" % (n_prof, dt_prof / n_prof * 1e3, extract_ms),
                       "end_to_end_algorithmic_GBps": round(ab["total"] * world * B * args.steps / dt / 1e9, 2)},
            "roofline": {"bound": "hbm", "kernel": kern[dom], "stage": dom, "launches_per_step": launches,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "peak_measured": HBM_MEASURED_GBS, "frac_of_measured_peak": round(achieved / HBM_MEASURED_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_step": int(dom_bytes), "avg_launch_ms": round(stage[dom] / launches, 4),
                         # what actually limits this kernel: its vector-instruction issue slots (committed PMC pass, VGA/1024 workload)
                         "valu": valu,
                         # every kernel priced in bytes, not only the dominant one: algorithmic rate from the live stage times, counter traffic
                         # (committed PMC pass, matching workload only) against the measured ceilings of tools/copy_probe.hip
                         "all_kernels": all_k,
                         "hbm_measured_gbs": {"read": 6370.0, "write": 4660.0, "copy": 4900.0, "source": "tools/copy_probe.hip, 2 GB spans"}},
        }
        if ba is not None:
            out["ba"] = ba
        if ba_sh is not None:
            out["ba_sharded"] = ba_sh
        if pose is not None:
            out["pose_opt"] = pose
        if stereo is not None:
            out["stereo"] = stereo
        if tracking is not None:
            out["tracking"] = tracking
        if inertial is not None:
            out["inertial_ba"] = inertial
        if hd is not None:
            out["hd"] = hd
        if uhd is not None:
            out["4k"] = uhd
        if latency is not None:
            out["latency"] = latency
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H, args.nfeatures)
            if graphs is not None:
                out["ba"]["cpu_baseline"] = ba_cpu_baseline(graphs)
            if pose_probs is not None:
                out["pose_opt"]["cpu_baseline"] = pose_cpu_baseline(pose_probs)
            if iba_wins is not None:
                out["inertial_ba"]["cpu_baseline"] = iba_cpu_baseline(iba_wins)
            if tracking is not None:
                tb = tracking_cpu_baseline(kp_h, desc_h, cnt_h, q, bnds)
                for name, v in tb.items():
                    out["tracking"][name]["cpu_baseline"] = v
            if stereo is not None and stereo_host is not None:
                out["stereo"]["cpu_baseline"] = stereo_cpu_baseline(stereo_host[0], stereo_host[1], args.nfeatures, 40.0 / 458.0, 40.0)
            for key, leg, simgs in (("hd", hd, hd_sample_imgs), ("4k", uhd, uhd_sample_imgs)):
                if leg is None:
                    continue
                import hashlib
                import oracle_bind as ob
                oe = ob.OracleExtractor(leg["nfeatures"], 1.2, 8, 20, 7)
                dig = hashlib.sha256()
                t0 = time.time()
                for im in simgs:
                    ok_, od_, _ = oe.extract(im, (0, 0))
                    dig.update(ok_.tobytes()); dig.update(od_.tobytes())
                t_or = (time.time() - t0) / len(simgs)
                out[key]["oracle_sha256_of_the_same_frames"] = dig.hexdigest()
                out[key]["bit_exact_vs_oracle_on_sampled_frames"] = dig.hexdigest() == leg["sha256_keypoints_and_descriptors_of_sampled_frames"]
                out[key]["cpu_oracle_extract_ms_1thread"] = round(t_or * 1e3, 1)
                assert out[key]["bit_exact_vs_oracle_on_sampled_frames"], "%s leg: sampled frames differ from the oracle" % key
            if latency is not None:
                import oracle_bind as ob
                oe = ob.OracleExtractor(args.nfeatures, 1.2, 8, 20, 7)
                oe.extract(imgs[0], (0, 0))
                t0 = time.time()
                for i in range(3):
                    oe.extract(imgs[i % B], (0, 0))
                latency["extract_one_frame"]["cpu_oracle_ms_1thread"] = round((time.time() - t0) / 3 * 1e3, 3)
                if "search_for_initialization_one_pair" in latency:
                    import oracle_match_bind as om
                    ka, da, _ = oe.extract(imgs[0], (0, 0)); kb, db, _ = oe.extract(imgs[1], (0, 0))
                    pv = np.stack([ka["x"], ka["y"]], 1)
                    t0 = time.time()
                    for i in range(20):
                        om.search_for_initialization(ka, da, kb, db, (0.0, 0.0, float(W), float(H)), pv, 100, 0.9, True)
                    latency["search_for_initialization_one_pair"]["cpu_oracle_ms_1thread"] = round((time.time() - t0) / 20 * 1e3, 4)
                for name in ("search_by_projection_last_frame", "search_by_projection_local_map"):
                    if name + "_one_pair" in latency:
                        latency[name + "_one_pair"]["cpu_oracle_ms_1thread"] = out["tracking"][name]["cpu_baseline"]["single_thread_ms"]
                for key, src in (("pose_optimization_one_frame", "pose_opt"), ("local_ba_one_window", "ba"), ("inertial_ba_one_window", "inertial_ba")):
                    if key in latency and src in out and "cpu_baseline" in out[src]:
                        latency[key]["cpu_oracle_ms_1thread"] = out[src]["cpu_baseline"]["single_thread_ms"]
                hcl = latency.get("host_classes", {})
                for k, t in (("extract_one_frame", "extract_one_frame"), ("search_by_projection_last_frame", "search_by_projection_last_frame_one_pair"),
                             ("search_by_projection_local_map", "search_by_projection_local_map_one_pair"), ("pose_optimization_one_frame", "pose_optimization_one_frame"),
                             ("local_ba_one_window", "local_ba_one_window")):
                    if k in hcl and t in latency and "cpu_oracle_ms_1thread" in latency[t]:
                        hcl[k]["cpu_oracle_ms_1thread"] = latency[t]["cpu_oracle_ms_1thread"]
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def device_view(torch, ptr, shape, typestr):
    """torch view of a raw device address owned by the library (no copy), through __cuda_array_interface__"""
    class _Raw:
        pass
    r = _Raw()
    r.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(r, device="cuda")


if __name__ == "__main__":
    main()
