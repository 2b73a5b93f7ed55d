#!/usr/bin/env python3
"""bench.py -- ORB extract+match throughput on synthetic VGA batches (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch already resident in HBM:
  ORBextractor::operator() on `batch` frames (pyramid, FAST+NMS, octree, blur, IC-angle,
  rBRIEF, lapping assembly) + Hamming 2-NN match of every frame against its successor.
Prints ONE JSON line (rank 0).  N>1: one process per GPU, frames sharded, no data-path
collective -> "scaling": "weak".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec" (6.29 TB/s measured copy)


def level_dims(w, h, nlevels=8, scale=1.2):
    import numpy as np
    sf = np.float32(1.0)
    dims = []
    for l in range(nlevels):
        inv = np.float32(1.0) / sf
        dims.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        sf = np.float32(sf * np.float32(scale))
    return dims


def algorithmic_bytes(w, h, n_kp, nlevels=8):
    """per-frame algorithmic bytes of each stage."""
    dims = level_dims(w, h, nlevels)
    S = sum(a * b for a, b in dims)
    S_lo = S - dims[-1][0] * dims[-1][1]
    S_hi = S - dims[0][0] * dims[0][1]
    return {"pyramid": S_lo + S_hi, "blur_score": 3 * S, "fast_cells": 0, "desc": n_kp * (749 + 512) + n_kp * 36,
            "octree": 0, "assemble": n_kp * 60 * 2, "total": S_lo + S_hi + S + 2 * S + n_kp * (749 + 512) + n_kp * 60}


def cpu_baseline(w, h, nfeat, seconds_budget=20.0):
    """Oracle (CPU restatement of the reference path) timed on this host's cores: kind 'port'."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    import orbhip
    import oracle_bind as ob
    import oracle_match_bind as om
    cores = min(os.cpu_count() or 1, 16)
    # single-thread probe: 4 frames
    imgs = orbhip.synth_frames(w, h, 5, seed=4242)
    e = ob.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    t0 = time.time()
    res = [e.extract(imgs[i], (0, 0)) for i in range(5)]
    def match(a, b):
        om.bf2nn(a[1], b[1], 0.7)
        om.search_for_initialization(a[0], a[1], b[0], b[1], (0.0, 0.0, float(w), float(h)),
                                     np.stack([a[0]["x"], a[0]["y"]], 1), 100, 0.9, True)
    for i in range(4):
        match(res[i], res[i + 1])
    t1 = (time.time() - t0) / 4.0
    per_thread = max(4, int(seconds_budget / max(t1, 1e-3) / 1.0 / 1) // cores)
    per_thread = min(per_thread, 48)

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
    return {"value": round(done / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
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
    return {"value": round(done / dt, 1), "unit": "frames/s", "cores": cores, "kind": "port",
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
    return {"value": round(done / dt, 2), "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": "%d threads x %d solves of the 50KFx2000ptx10obs graph; single-thread %.2f solves/s"
                      % (cores, per_thread, 1.0 / t1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--ba-graphs", type=int, default=256, help="local-BA graphs solved concurrently per GPU (0 = skip BA leg)")
    ap.add_argument("--ba-steps", type=int, default=3)
    ap.add_argument("--ba-sharded-graphs", type=int, default=0, help="N > 1 only, opt-in: graphs solved cooperatively with the points sharded over the ranks and the Schur block all-gathered every LM trial (SURVEY 8e optional mode)")
    ap.add_argument("--pose-frames", type=int, default=1024, help="frames of pose-only BA solved per launch (0 = skip)")
    ap.add_argument("--stereo-pairs", type=int, default=256, help="rectified stereo pairs for the ComputeStereoMatches leg (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import orbhip

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    B, W, H = args.batch, args.width, args.height
    # synthetic frames: generated on the host, then resident in HBM before the timed region
    imgs = orbhip.synth_frames(W, H, B, seed=20241004, first=rank * B)
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
    d_nm = torch.empty((B,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    kp_p, desc_p, cnt_p, mono_p = ext.results_device()
    dstride = max_kp * 32

    def step():
        # lap (0,0): keypoints come out in level order (the stereo constructors' lapping, Frame.cc:109-110)
        ext.extract_device(d_imgs.data_ptr(), W, H, W, W * H, B, (0, 0))
        # frame i vs frame i+1 (B-1 pairs) + wrap-around pair (B-1 vs 0): every frame matched once
        if B > 1:
            orbhip.match_bf2nn_device(ctx, desc_p, cnt_p, dstride, desc_p + dstride, cnt_p + 4, dstride, B - 1, max_kp,
                                      0.7, d_idx2.data_ptr(), d_dist2.data_ptr(), d_acc.data_ptr())
        orbhip.match_bf2nn_device(ctx, desc_p + (B - 1) * dstride, cnt_p + 4 * (B - 1), dstride, desc_p, cnt_p, dstride, 1,
                                  max_kp, 0.7, d_idx2.data_ptr() + (B - 1) * max_kp * 8,
                                  d_dist2.data_ptr() + (B - 1) * max_kp * 8, d_acc.data_ptr() + (B - 1) * max_kp)
        if B > 1:
            windowed()

    def windowed():
        # ORBmatcher::SearchForInitialization(frame i, frame i+1) (Tracking.cc:1506-1507: ORBmatcher(0.9,true),
        # windowSize 100) with vbPrevMatched = frame i's keypoint positions (Tracking.cc:1497-1499); B-1 pairs.
        orbhip.prev_matched_init_device(ctx, kp_p, max_kp, B - 1, max_kp, d_prev.data_ptr())
        orbhip.search_for_initialization_device(ctx, kp_p, desc_p, cnt_p, kp_p + max_kp * 28, desc_p + dstride, cnt_p + 4,
                                                B - 1, max_kp, max_kp, (0.0, 0.0, float(W), float(H)), 100, 0.9, True,
                                                d_prev.data_ptr(), d_m12.data_ptr(), d_nm.data_ptr())

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    ext.set_profiling(True)
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    stage = ext.stage_ms()
    ext.set_profiling(False)

    ctx.check_status()                                   # loud failure on any device-side capacity overflow
    res_chk = ext.extract_host(imgs[:min(B, 4)], (0, 0))  # host entry point re-check (raises on capacity errors)
    n_kp_avg = float(np.mean([len(r[0]) for r in res_chk]))
    win_matches = float(d_nm[:max(B - 1, 1)].float().mean().item()) if B > 1 else 0.0
    bf_accept = float(d_acc.float().sum().item()) / B

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the only data exchange of the sharded ORB path: one all-gather of fixed-size per-frame
        # records, outside the timed region (a host-side Tracking consumer would D2H per GPU instead)
        import shard
        rec = torch.stack([d_nm.to(torch.int32), d_nm.to(torch.int32)], 1).to(coll_dev)
        allrec = shard.allgather_records(rec)
        assert allrec.shape[0] == world

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
        bb.set_profiling(True)
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.ba_steps):
            bb.solve()
        sync()
        if world > 1:
            dist.barrier()
        dt_ba = time.perf_counter() - t0
        gemm_ms, gemm_n, gemm_fl = bb.gemm_profile()
        gemm_dense = bb.gemm_dense_flops()
        gemm_issued = bb.gemm_issued_flops()
        ticks = bb.ticks
        _, _, _, stats = bb.download()
        bb.close()
        if world > 1:
            t = torch.tensor([dt_ba], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ba = float(t.item())
        peak64 = orbhip.mfma_f64_peak_tflops(ctx)
        ba_traffic, ba_traffic_src = None, None
        try:     # HBM bytes per GEMM launch from the committed PMC passes (256-graph workload only)
            if args.ba_graphs == 256:
                pmc_ba = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_ba.json")))
                ba_traffic = pmc_ba["kernels"]["k_ba_schur_gemm"]["hbm_bytes_per_launch"]
                ba_traffic_src = "profiles/r01_pmc_traffic_ba.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch)"
        except Exception:
            pass
        tfl = gemm_fl * gemm_n / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        ba = {"metric": "local-BA solves/sec", "value": round(world * args.ba_graphs * args.ba_steps / dt_ba, 2),
              "unit": "solves/s", "graphs_per_gpu": args.ba_graphs, "ms_per_batch": round(dt_ba / args.ba_steps * 1e3, 2),
              "lm_ticks": ticks, "workload": "50 KF (2 fixed) x 2000 points x 10 obs, 5+10 LM iterations, Huber, Schur",
              "lm_trials_graph0": stats[0]["lm_trials"], "dtype": "f64",
              "roofline": {"bound": "mfma", "kernel": "k_ba_schur_gemm", "achieved": round(tfl, 2),
                           "peak": round(peak64, 2), "unit": "TFLOP/s", "frac": round(tfl / peak64, 4) if peak64 else None,
                           "traffic": ba_traffic, "traffic_source": ba_traffic_src,
                           "peak_source": "measured v_mfma_f64_16x16x4_f64 micro-benchmark on this device",
                           "avg_launch_ms": round(gemm_ms / max(gemm_n, 1), 4),
                           "mfma_flops_of_data_tiles_per_launch": gemm_fl, "mfma_flops_issued_per_launch": gemm_issued,
                           "flops_per_launch_without_sparsity_skipping": gemm_dense,
                           "note": "achieved = flops of the 16x16x4 MFMAs whose tiles hold data / hipEvent time; only tiles that hold data are issued (per-stage ballot hit maps)"}}

    # ---- landmark-sharded single-graph mode (optional): the SAME graphs solved by all ranks together, the shared Schur
    # block all-gathered every LM trial (RCCL over xGMI with backend nccl).  Latency-bound by design; reported, not hidden.
    ba_sh = None
    if world > 1 and args.ba_graphs > 0 and args.ba_sharded_graphs > 0:
        try:
            import shard
            import synth_ba
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
            sb.solve_sharded(xch)                        # warm-up
            n_x[0] = 0
            dist.barrier(); sync()
            t0 = time.perf_counter()
            sb.solve_sharded(xch)
            sync(); dist.barrier()
            dt_sh = time.perf_counter() - t0
            t = torch.tensor([dt_sh], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            st_sh = sb.download()[3]
            ba_sh = {"metric": "local-BA solves/sec, points sharded over all GPUs (one all-gather of the Schur block per LM trial)",
                     "value": round(args.ba_sharded_graphs / float(t.item()), 2), "unit": "solves/s", "graphs": args.ba_sharded_graphs,
                     "ranks": world, "all_gathers": n_x[0], "doubles_per_rank_and_gather": int(stride),
                     "lm_trials_graph0": st_sh[0]["lm_trials"], "backend": backend}
            sb.close()
        except Exception as e:                           # never lose the main line to the optional leg
            ba_sh = {"error": repr(e)[:300]}

    pose = None
    pose_probs = None
    if args.pose_frames > 0:
        import numpy as np
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
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(args.ba_steps):
            pose_step(dps[i])
        sync()
        if world > 1:
            dist.barrier()
        dt_po = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_po], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_po = float(t.item())
        pose = {"metric": "pose-only BA frames/sec", "value": round(world * F * args.ba_steps / dt_po, 1), "unit": "frames/s",
                "frames_per_gpu": F, "ms_per_batch": round(dt_po / args.ba_steps * 1e3, 3), "dtype": "f64",
                "workload": "Optimizer::PoseOptimization: 1000 unary edges/frame (0-75 % stereo), 10 % gross outliers, 4 rounds x 10 LM its",
                "mean_inliers": round(float(dni.float().mean().item()), 1)}

    stereo = None
    if args.stereo_pairs > 0:
        import numpy as np
        S = args.stereo_pairs
        big = orbhip.synth_frames(W + 64, H, S, seed=777 + 100000 * rank)
        disp = [4 + (7 * k) % 40 for k in range(S)]
        lefts = np.ascontiguousarray(big[:, :, 0:W])
        rights = np.stack([big[k, :, disp[k]:disp[k] + W] for k in range(S)])
        ctx_r = orbhip.Context(local_rank)
        ext_l = orbhip.Extractor(ctx, args.nfeatures, 1.2, 8, 20, 7); ext_r = orbhip.Extractor(ctx_r, args.nfeatures, 1.2, 8, 20, 7)
        ext_l.reserve(W, H, S); ext_r.reserve(W, H, S)
        d_l = torch.from_numpy(lefts).cuda(); d_r = torch.from_numpy(np.ascontiguousarray(rights)).cuda()
        mk = ext_l.max_keypoints
        d_ur = torch.empty((S, mk), dtype=torch.float32, device="cuda"); d_dp = torch.empty((S, mk), dtype=torch.float32, device="cuda")
        d_nk = torch.zeros((S,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()

        def stereo_step():
            # the two extractors run on their own streams (the stereo constructor's two threads, Frame.cc:109-110)
            ext_l.extract_device(d_l.data_ptr(), W, H, W, W * H, S, (0, 0))
            ext_r.extract_device(d_r.data_ptr(), W, H, W, W * H, S, (0, 0))
            orbhip.compute_stereo_matches_device(ext_l, ext_r, 40.0 / 458.0, 40.0, d_ur.data_ptr(), d_dp.data_ptr(), d_nk.data_ptr())
        stereo_step(); ctx_r.synchronize(); sync()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.ba_steps):
            stereo_step()
        ctx_r.synchronize(); sync()
        t_all = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(args.ba_steps):
            orbhip.compute_stereo_matches_device(ext_l, ext_r, 40.0 / 458.0, 40.0, d_ur.data_ptr(), d_dp.data_ptr(), d_nk.data_ptr())
        sync()
        t_match = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
            t = torch.tensor([t_all, t_match], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_all, t_match = float(t[0].item()), float(t[1].item())
        stereo = {"metric": "stereo frame pairs/sec (2 x ORB extract + ComputeStereoMatches)", "value": round(world * S * args.ba_steps / t_all, 1),
                  "unit": "pairs/s", "pairs_per_gpu": S, "ms_per_batch": round(t_all / args.ba_steps * 1e3, 3),
                  "compute_stereo_matches_ms_per_batch": round(t_match / args.ba_steps * 1e3, 3),
                  "mean_stereo_matches_per_pair": round(float(d_nk.float().mean().item()), 1),
                  "workload": "synthetic rectified %dx%d pairs, disparity 4..43 px, %d feats" % (W, H, args.nfeatures)}
        ext_l.close(); ext_r.close(); ctx_r.close()

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        fps = world * B * args.steps / dt
        ab = algorithmic_bytes(W, H, n_kp_avg)
        # dominant kernel among those with an algorithmic byte count: one stage == one kernel
        # (k_blur_score = SURVEY's "FAST read S" + "blur read+write 2S" done from one staged tile)
        kern = {"pyramid": "k_resize", "blur_score": "k_blur_score", "desc": "k_orient_desc"}
        dom = max(kern, key=lambda k: stage[k])
        launches = {"pyramid": 7, "blur_score": 1, "desc": 1}[dom]
        dom_bytes = ab[dom] * B
        achieved = dom_bytes / (stage[dom] * 1e-3) / 1e9 if stage[dom] > 0 else 0.0
        # HBM traffic of the dominant kernel: PMC counters are collected in separate rocprofv3 passes (they cannot be
        # read from inside this process); the per-step sums of the committed pass are reported when the workload matches.
        valu_busy, valu_src = None, None
        try:
            if (B, W, H) == (1024, 640, 480):
                pv = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_valu_issue.json")))
                valu_busy = pv["kernels"][kern[dom]]["valu_issue_busy_frac"]
                valu_src = "profiles/r01_pmc_valu_issue.json (rocprofv3 --pmc: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs))"
        except Exception:
            pass
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if (W, H, B, args.nfeatures) == (640, 480, 1024, 1000) and kern[dom] in pmc["kernels"]:
                traffic = pmc["kernels"][kern[dom]]["hbm_bytes_per_step"]
                traffic_src = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per step)"
        except Exception:
            pass
        out = {
            "metric": "ORB extract+match frames/sec", "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "synthetic %dx%d batch=%d per GPU, 8-level pyramid, %d feats/frame, "
                                   "ORB extract + Hamming 2-NN match (Frame.cc:1146) + SearchForInitialization "
                                   "(ORBmatcher.cc:710) vs successor frame" % (W, H, B, args.nfeatures),
                       "keypoints_per_frame": round(n_kp_avg, 1),
                       "bf_ratio_matches_per_frame": round(bf_accept, 1),
                       "windowed_matches_per_pair": round(win_matches, 1),
                       "stage_ms": {k: round(v, 4) for k, v in stage.items()},
                       "end_to_end_algorithmic_GBps": round(ab["total"] * B * args.steps / dt / 1e9, 2)},
            "roofline": {"bound": "hbm", "kernel": kern[dom], "stage": dom, "launches_per_step": launches,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_step": int(dom_bytes),
                         # what actually limits this kernel: its vector-instruction issue slots (committed PMC pass, VGA/1024 workload)
                         "valu_issue_busy_frac": valu_busy, "valu_issue_source": valu_src},
        }
        if ba is not None:
            out["ba"] = ba
        if ba_sh is not None:
            out["ba_sharded"] = ba_sh
        if pose is not None:
            out["pose_opt"] = pose
        if stereo is not None:
            out["stereo"] = stereo
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H, args.nfeatures)
            if graphs is not None:
                out["ba"]["cpu_baseline"] = ba_cpu_baseline(graphs)
            if pose_probs is not None:
                out["pose_opt"]["cpu_baseline"] = pose_cpu_baseline(pose_probs)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
