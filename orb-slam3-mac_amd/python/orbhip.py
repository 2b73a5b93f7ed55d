"""ctypes binding of the orbhip C ABI (include/orbhip.h) for tests and bench.py.

Plumbing only: every call goes straight to liborbhip.so (HIP kernels).  There is no
Python or CPU implementation behind these wrappers; if the shared library is missing the
import fails loudly.

torch is imported BEFORE the library is loaded so that one HIP runtime (torch's bundled
libamdhip64.so.7) serves both torch tensors and our kernels in the same process.
"""
import ctypes as C
import os
import numpy as np

try:  # one HIP runtime per process: let torch load its copy first when torch is present
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(os.path.dirname(_HERE), "lib")
LIB_PATH = os.path.join(LIB_DIR, "liborbhip.so")
SYNTH_PATH = os.path.join(LIB_DIR, "libsynth.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "liborbhip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(there is no CPU fallback for the HIP path)")

lib = C.CDLL(LIB_PATH)

OK, E_BADARG, E_NODEVICE, E_HIP, E_CAPACITY, E_ABORTED, E_NOTSPD, E_EMPTY = 0, -1, -2, -3, -4, -5, -6, -7
STAGES = ["pyramid", "fast_cells", "blur", "octree", "desc", "assemble"]

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

vp, ci, cf, cd, sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
lib.orbhip_version.restype = C.c_char_p
lib.orbhip_last_error.restype = C.c_char_p
lib.orbhip_ctx_create.argtypes = [ci, vp, C.POINTER(vp)]
lib.orbhip_ctx_destroy.argtypes = [vp]
lib.orbhip_ctx_synchronize.argtypes = [vp]
lib.orbhip_ctx_stream.argtypes = [vp]
lib.orbhip_ctx_stream.restype = vp
lib.orbhip_extractor_create.argtypes = [vp, ci, cf, ci, ci, ci, C.POINTER(vp)]
lib.orbhip_extractor_destroy.argtypes = [vp]
lib.orbhip_extractor_levels.argtypes = [vp]
lib.orbhip_extractor_table.argtypes = [vp, ci, vp]
lib.orbhip_extractor_features_per_level.argtypes = [vp, vp]
lib.orbhip_extractor_umax.argtypes = [vp, vp]
lib.orbhip_extractor_reserve.argtypes = [vp, ci, ci, ci]
lib.orbhip_extractor_max_keypoints.argtypes = [vp]
lib.orbhip_extract_batch_device.argtypes = [vp, vp, ci, ci, sz, sz, ci, ci, ci]
lib.orbhip_extractor_results.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
lib.orbhip_extract_batch_host.argtypes = [vp, vp, ci, ci, sz, sz, ci, ci, ci, vp, vp, ci, vp, vp]
lib.orbhip_extractor_level_dims.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci)]
lib.orbhip_extractor_get_pyramid_level.argtypes = [vp, ci, ci, ci, vp, sz]
lib.orbhip_extractor_get_blurred_level.argtypes = [vp, ci, ci, vp, sz]
lib.orbhip_extractor_get_pyramid_padded.argtypes = [vp, ci, vp, vp]
lib.orbhip_extractor_get_fast_candidates.argtypes = [vp, ci, ci, vp, vp, vp, ci, C.POINTER(C.c_int32)]
lib.orbhip_extractor_get_level_keypoints.argtypes = [vp, ci, ci, vp, ci, C.POINTER(C.c_int32)]
lib.orbhip_extractor_set_profiling.argtypes = [vp, ci]
lib.orbhip_extractor_set_graph_mode.argtypes = [vp, ci]
lib.orbhip_extractor_stage_ms.argtypes = [vp, vp]
lib.orbhip_descriptor_distance.argtypes = [vp, vp]
lib.orbhip_match_bf2nn_device.argtypes = [vp, vp, vp, sz, vp, vp, sz, ci, ci, cd, vp, vp, vp]
lib.orbhip_search_for_initialization_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, sz, cf, cf, cf, cf, ci, cf, ci,
                                                        vp, vp, vp]
lib.orbhip_ctx_check_status.argtypes = [vp]
lib.orbhip_ctx_wait_for.argtypes = [vp, vp]
lib.orbhip_extractor_blur_kernel.argtypes = [vp, ci]
lib.orbhip_extractor_blur_kernel.restype = ci
lib.orbhip_prev_matched_init_device.argtypes = [vp, vp, sz, ci, ci, vp]
lib.orbhip_search_by_projection_device.argtypes = [vp, vp, vp, vp, ci, vp, vp, vp, vp, ci, sz, ci, cf, cf, cf, cf, ci, ci, vp, vp]
lib.orbhip_search_local_map_device.argtypes = [vp, vp, vp, vp, ci, vp, vp, vp, vp, ci, sz, ci, cf, cf, cf, cf, ci, cf, vp, vp]


class OrbHipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__("%s failed: %d (%s)" % (where, code, lib.orbhip_last_error().decode()))


def _chk(rc, where):
    if rc != OK:
        raise OrbHipError(rc, where)


class Context:
    def __init__(self, device=0, stream=None):
        h = vp()
        _chk(lib.orbhip_ctx_create(device, stream, C.byref(h)), "orbhip_ctx_create")
        self.h = h

    def synchronize(self):
        _chk(lib.orbhip_ctx_synchronize(self.h), "orbhip_ctx_synchronize")

    def check_status(self):
        _chk(lib.orbhip_ctx_check_status(self.h), "orbhip_ctx_check_status")

    def wait_for(self, other):
        """work submitted to this context from now on starts after everything submitted to `other` so far (no host synchronisation)"""
        _chk(lib.orbhip_ctx_wait_for(self.h, other.h), "orbhip_ctx_wait_for")

    @property
    def stream(self):
        return lib.orbhip_ctx_stream(self.h)

    def close(self):
        if self.h:
            lib.orbhip_ctx_destroy(self.h)
            self.h = None


class Extractor:
    """Mirror of ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:49-81)."""

    def __init__(self, ctx, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        h = vp()
        _chk(lib.orbhip_extractor_create(ctx.h, nfeatures, scale_factor, nlevels, ini_th, min_th, C.byref(h)),
             "orbhip_extractor_create")
        self.h, self.ctx, self.nlevels = h, ctx, nlevels

    def close(self):
        if self.h:
            lib.orbhip_extractor_destroy(self.h)
            self.h = None

    # -- tables (GetScaleFactors etc.)
    def table(self, which):
        out = np.zeros(self.nlevels, np.float32)
        _chk(lib.orbhip_extractor_table(self.h, which, out.ctypes.data), "orbhip_extractor_table")
        return out

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        _chk(lib.orbhip_extractor_features_per_level(self.h, out.ctypes.data), "features_per_level")
        return out

    def umax(self):
        out = np.zeros(16, np.int32)
        _chk(lib.orbhip_extractor_umax(self.h, out.ctypes.data), "umax")
        return out

    def reserve(self, w, h, batch):
        _chk(lib.orbhip_extractor_reserve(self.h, w, h, batch), "orbhip_extractor_reserve")

    @property
    def max_keypoints(self):
        return lib.orbhip_extractor_max_keypoints(self.h)

    def blur_kernel(self, batch):
        """the kernel that blurs a batch of that many frames: 'k_blur' (LDS tiles), 'k_blur_rows' or 'k_blur_mfma'"""
        return ("k_blur", "k_blur_rows", "k_blur_mfma")[lib.orbhip_extractor_blur_kernel(self.h, int(batch))]

    def set_graph_mode(self, on):
        _chk(lib.orbhip_extractor_set_graph_mode(self.h, 1 if on else 0), "orbhip_extractor_set_graph_mode")

    def set_profiling(self, on):
        _chk(lib.orbhip_extractor_set_profiling(self.h, 1 if on else 0), "set_profiling")

    def stage_ms(self):
        out = np.zeros(len(STAGES), np.float32)
        _chk(lib.orbhip_extractor_stage_ms(self.h, out.ctypes.data), "stage_ms")
        return dict(zip(STAGES, out.tolist()))

    # -- operator()
    def extract_host(self, images, lap=(0, 1000)):
        """images: uint8 [B,H,W] (or [H,W]).  Returns list of (kp structured array, desc [n,32], mono_index)."""
        img = np.ascontiguousarray(images, np.uint8)
        if img.ndim == 2:
            img = img[None]
        B, H, W = img.shape
        self.reserve(W, H, B)
        cap = self.max_keypoints
        kp = np.zeros((B, cap), KP_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        cnt = np.zeros(B, np.int32)
        mono = np.zeros(B, np.int32)
        _chk(lib.orbhip_extract_batch_host(self.h, img.ctypes.data, W, H, W, W * H, B, lap[0], lap[1],
                                           kp.ctypes.data, desc.ctypes.data, cap, cnt.ctypes.data, mono.ctypes.data),
             "orbhip_extract_batch_host")
        return [(kp[f, :cnt[f]].copy(), desc[f, :cnt[f]].copy(), int(mono[f])) for f in range(B)]

    def extract_device(self, d_ptr, w, h, row_stride, frame_stride, batch, lap=(0, 1000)):
        _chk(lib.orbhip_extract_batch_device(self.h, d_ptr, w, h, row_stride, frame_stride, batch, lap[0], lap[1]),
             "orbhip_extract_batch_device")

    def results_device(self):
        a, b, c, d = vp(), vp(), vp(), vp()
        _chk(lib.orbhip_extractor_results(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "results")
        return a.value, b.value, c.value, d.value

    # -- taps
    def level_dims(self, level):
        w, h = ci(), ci()
        _chk(lib.orbhip_extractor_level_dims(self.h, level, C.byref(w), C.byref(h)), "level_dims")
        return w.value, h.value

    def pyramid_level(self, frame, level, padded=False):
        w, h = self.level_dims(level)
        if padded:
            w, h = w + 38, h + 38
        out = np.zeros((h, w), np.uint8)
        _chk(lib.orbhip_extractor_get_pyramid_level(self.h, frame, level, 1 if padded else 0, out.ctypes.data, w),
             "get_pyramid_level")
        return out

    def pyramid_padded(self, frame):
        """All levels as (w+38) x (h+38) reflect-101 padded parents (what mvImagePyramid's ROI views live in)."""
        outs = [np.zeros((h + 38, w + 38), np.uint8) for w, h in (self.level_dims(l) for l in range(self.nlevels))]
        ptrs = (vp * self.nlevels)(*[o.ctypes.data for o in outs])
        strides = (C.c_size_t * self.nlevels)(*[o.shape[1] for o in outs])
        _chk(lib.orbhip_extractor_get_pyramid_padded(self.h, frame, ptrs, strides), "get_pyramid_padded")
        return outs

    def blurred_level(self, frame, level):
        w, h = self.level_dims(level)
        out = np.zeros((h, w), np.uint8)
        _chk(lib.orbhip_extractor_get_blurred_level(self.h, frame, level, out.ctypes.data, w), "get_blurred_level")
        return out

    def fast_candidates(self, frame, level, cap=1 << 17):
        xs, ys, ss = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int32()
        _chk(lib.orbhip_extractor_get_fast_candidates(self.h, frame, level, xs.ctypes.data, ys.ctypes.data,
                                                      ss.ctypes.data, cap, C.byref(n)), "get_fast_candidates")
        return xs[:n.value].copy(), ys[:n.value].copy(), ss[:n.value].copy()

    def level_keypoints(self, frame, level, cap=8192):
        out = np.zeros(cap, KP_DTYPE)
        n = C.c_int32()
        _chk(lib.orbhip_extractor_get_level_keypoints(self.h, frame, level, out.ctypes.data, cap, C.byref(n)),
             "get_level_keypoints")
        return out[:n.value].copy()


def match_bf2nn_device(ctx, d_descA, d_nA, strideA, d_descB, d_nB, strideB, pairs, max_n, ratio, d_idx2, d_dist2,
                       d_accept):
    """Frame.cc:1146-1153 semantics, batched; all pointers are device addresses (ints)."""
    _chk(lib.orbhip_match_bf2nn_device(ctx.h, d_descA, d_nA, strideA, d_descB, d_nB, strideB, pairs, max_n, ratio,
                                       d_idx2, d_dist2, d_accept), "orbhip_match_bf2nn_device")


def search_for_initialization_device(ctx, d_kpA, d_descA, d_nA, d_kpB, d_descB, d_nB, pairs, max_n, kp_stride, bounds,
                                     window, nn_ratio, check_ori, d_prev, d_m12, d_nmatches):
    """ORBmatcher::SearchForInitialization, batched; all pointers are device addresses (ints)."""
    _chk(lib.orbhip_search_for_initialization_device(ctx.h, d_kpA, d_descA, d_nA, d_kpB, d_descB, d_nB, pairs, max_n,
                                                     kp_stride, bounds[0], bounds[1], bounds[2], bounds[3], window,
                                                     nn_ratio, 1 if check_ori else 0, d_prev, d_m12, d_nmatches),
         "orbhip_search_for_initialization_device")


# one projected map point of ORBmatcher::SearchByProjection (include/orbhip.h: orbhip_proj_query)
PROJ_QUERY_DTYPE = np.dtype([("u", np.float32), ("v", np.float32), ("radius", np.float32), ("ur", np.float32),
                             ("angle", np.float32), ("min_level", np.int32), ("max_level", np.int32), ("has_obs", np.int32)])


def search_by_projection_device(ctx, d_q, d_desc_q, d_nq, max_q, d_kp, d_desc, d_u_right, d_n, max_n, kp_stride, pairs,
                                bounds, th_high, check_ori, d_train_match, d_nmatches):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), batched; device addresses (ints)."""
    _chk(lib.orbhip_search_by_projection_device(ctx.h, d_q, d_desc_q, d_nq, max_q, d_kp, d_desc, d_u_right, d_n, max_n,
                                                kp_stride, pairs, bounds[0], bounds[1], bounds[2], bounds[3], th_high,
                                                1 if check_ori else 0, d_train_match, d_nmatches),
         "orbhip_search_by_projection_device")


def search_local_map_device(ctx, d_q, d_desc_q, d_nq, max_q, d_kp, d_desc, d_u_right, d_n, max_n, kp_stride, pairs,
                            bounds, th_high, nn_ratio, d_train_match, d_nmatches):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) (TrackLocalMap), batched; device addresses (ints)."""
    _chk(lib.orbhip_search_local_map_device(ctx.h, d_q, d_desc_q, d_nq, max_q, d_kp, d_desc, d_u_right, d_n, max_n,
                                            kp_stride, pairs, bounds[0], bounds[1], bounds[2], bounds[3], th_high,
                                            nn_ratio, d_train_match, d_nmatches), "orbhip_search_local_map_device")


def prev_matched_init_device(ctx, d_kp, kp_stride, frames, max_n, d_prev):
    _chk(lib.orbhip_prev_matched_init_device(ctx.h, d_kp, kp_stride, frames, max_n, d_prev), "orbhip_prev_matched_init_device")


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib.orbhip_descriptor_distance(a.ctypes.data, b.ctypes.data)


# ---------------------------------------------------------------- synthetic inputs (host C)
_synth = None


def synth_lib():
    global _synth
    if _synth is None:
        _synth = C.CDLL(SYNTH_PATH)
        _synth.synth_frame.argtypes = [vp, ci, ci, ci, C.c_uint64, ci]
        _synth.synth_batch.argtypes = [vp, ci, ci, ci, C.c_uint64, ci]
        _synth.synth_frame_transform.argtypes = [C.c_uint64, ci, C.POINTER(cf), C.POINTER(cf), C.POINTER(cf)]
    return _synth


def synth_frames(w, h, n, seed=20241004, first=0):
    out = np.zeros((n, h, w), np.uint8)
    synth_lib().synth_batch(out.ctypes.data, w, h, n, seed, first)
    return out


# ---------------------------------------------------------------- local BA (Optimizer::LocalBundleAdjustment core)
def rig2_fields(g):
    """Trl, fx2, fy2, cx2, cy2, camera2_model, kb2 of a graph dict (g.get('rig2') = dict(Trl, cam=(fx,fy,cx,cy), kb or None))."""
    r = g.get("rig2")
    if r is None:
        return ((cd * 7)(0, 0, 0, 0, 0, 0, 0), 0.0, 0.0, 0.0, 0.0, 0, (cd * 4)(0, 0, 0, 0))
    kb2 = r.get("kb")
    return ((cd * 7)(*r["Trl"]), *[float(c) for c in r["cam"]], 1 if kb2 is not None else 0, (cd * 4)(*(kb2 if kb2 is not None else (0, 0, 0, 0))))


class BaCamera(C.Structure):
    _fields_ = [("fx", cd), ("fy", cd), ("cx", cd), ("cy", cd), ("bf", cd), ("camera_model", C.c_int32), ("kb", cd * 4),
                ("Trl", cd * 7), ("fx2", cd), ("fy2", cd), ("cx2", cd), ("cy2", cd), ("camera2_model", C.c_int32), ("kb2", cd * 4)]


class BaGraph(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("pose_fixed", vp), ("edge_pose", vp), ("edge_point", vp), ("edge_obs", vp),
                ("edge_inv_sigma2", vp), ("edge_stereo", vp),
                ("fx", cd), ("fy", cd), ("cx", cd), ("cy", cd), ("bf", cd),
                ("camera_model", C.c_int32), ("kb", cd * 4),
                ("Trl", cd * 7), ("fx2", cd), ("fy2", cd), ("cx2", cd), ("cy2", cd), ("camera2_model", C.c_int32), ("kb2", cd * 4),
                ("n_cameras", C.c_int32), ("cameras", vp), ("pose_camera", vp)]


def camera_table(g, cls=None):
    """(n_cameras, array of camera structs or None, pose_camera array or None) of a graph dict (per-keyframe calibration, synth_ba.make_graph(cameras=...))."""
    cls = cls or BaCamera
    cams = g.get("cameras")
    if not cams:
        return 0, None, None
    arr = (cls * len(cams))()
    for i, c in enumerate(cams):
        kb = c.get("kb")
        arr[i] = cls(c["fx"], c["fy"], c["cx"], c["cy"], c["bf"], 1 if kb is not None else 0, (cd * 4)(*(kb if kb is not None else (0, 0, 0, 0))),
                     *rig2_fields(dict(rig2=c.get("rig2"))))
    return len(cams), arr, np.ascontiguousarray(g["pose_camera"], np.int32)


class BaParams(C.Structure):
    _fields_ = [("iters1", C.c_int32), ("iters2", C.c_int32), ("huber_mono2", cd), ("huber_stereo2", cd),
                ("user_lambda_init", cd), ("tau", cd), ("max_trials", C.c_int32),
                ("stage2_exclude_outliers", C.c_int32), ("stage2_drop_robust", C.c_int32), ("no_discard", C.c_int32),
                ("gate_mono2", cd), ("gate_stereo2", cd)]


class BaStats(C.Structure):
    _fields_ = [("iterations_run", C.c_int32 * 2), ("lm_trials", C.c_int32), ("n_outliers", C.c_int32),
                ("discarded", C.c_int32), ("chi2_initial", cd), ("chi2_final", cd)]

    def as_dict(self):
        return dict(iterations_run=list(self.iterations_run), lm_trials=self.lm_trials, n_outliers=self.n_outliers,
                    discarded=self.discarded, chi2_initial=self.chi2_initial, chi2_final=self.chi2_final)


lib.orbhip_ba_default_params.argtypes = [C.POINTER(BaParams)]
lib.orbhip_ba_merge_params.argtypes = [C.POINTER(BaParams)]
lib.orbhip_ba_batch_create.argtypes = [vp, vp, ci, vp, vp, C.POINTER(vp)]
lib.orbhip_ba_batch_solve.argtypes = [vp, C.POINTER(BaParams), vp]
lib.orbhip_ba_batch_download.argtypes = [vp, vp, vp, vp, vp]
lib.orbhip_ba_batch_ticks.argtypes = [vp]
lib.orbhip_ba_batch_destroy.argtypes = [vp]
lib.orbhip_ba_solve_batch.argtypes = [vp, vp, ci, C.POINTER(BaParams), vp, vp, vp, vp, vp]
lib.orbhip_ba_batch_set_profiling.argtypes = [vp, ci]
lib.orbhip_ba_batch_gemm_profile.argtypes = [vp, C.POINTER(cf), C.POINTER(ci), C.POINTER(cd)]
lib.orbhip_mfma_f64_peak_tflops.argtypes = [vp, C.POINTER(cd)]
lib.orbhip_ba_batch_gemm_dense_flops.argtypes = [vp]
lib.orbhip_ba_batch_gemm_issued_flops.argtypes = [vp]
lib.orbhip_ba_batch_gemm_issued_flops.restype = cd
lib.orbhip_ba_batch_gemm_dense_flops.restype = cd


def mfma_f64_peak_tflops(ctx):
    v = cd()
    _chk(lib.orbhip_mfma_f64_peak_tflops(ctx.h, C.byref(v)), "orbhip_mfma_f64_peak_tflops")
    return v.value


lib.orbhip_ctx_set_ba_schur_mode.argtypes = [vp, ci]


def ba_set_schur_mode(ctx, mode):
    """0 / 1 = pair lists (default), 2 = FP64-MFMA panel GEMM; a property of the context, read when a batch is created on it."""
    _chk(lib.orbhip_ctx_set_ba_schur_mode(ctx.h, mode), "orbhip_ctx_set_ba_schur_mode")


def ba_merge_params():
    """Parameters of the map-merge local BA (Optimizer.cc:6255)."""
    p = BaParams()
    lib.orbhip_ba_merge_params(C.byref(p))
    return p


def ba_global_params(iterations, robust=True):
    """Optimizer::BundleAdjustment / GlobalBundleAdjustemnt parameters (one optimize(iterations) pass)."""
    p = BaParams()
    lib.orbhip_ba_global_params(C.byref(p), iterations, 1 if robust else 0)
    return p


def ba_default_params():
    p = BaParams()
    lib.orbhip_ba_default_params(C.byref(p))
    return p


BA_EXCHANGE_FN = C.CFUNCTYPE(ci, vp, ci, C.c_size_t)
lib.orbhip_ba_batch_exchange_doubles.restype = C.c_size_t
lib.orbhip_ba_batch_exchange_doubles.argtypes = [vp]
lib.orbhip_ba_batch_set_exchange_buffer.argtypes = [vp, vp, C.c_size_t]
lib.orbhip_ba_batch_solve_sharded.argtypes = [vp, vp, vp, BA_EXCHANGE_FN, vp]
lib.orbhip_ba_batch_create_sharded.argtypes = [vp, vp, ci, vp, vp, ci, ci, vp]


class BaBatch:
    """Device-resident batch of keyframe-window graphs (dicts as produced by synth_ba.make_graph:
    n_poses, n_points, n_edges, pose_fixed, edge_pose, edge_point, edge_obs, edge_inv_sigma2,
    edge_stereo, fx, fy, cx, cy, bf, poses0, points0)."""

    def __init__(self, ctx, graphs, rank=0, world=1):
        """world > 1: landmark-sharded batch (orbhip_ba_batch_create_sharded): every rank passes the same graphs."""
        self.ctx, self.n = ctx, len(graphs)
        self.rank, self.world = rank, world
        self._keep = []
        arr = (BaGraph * self.n)()
        self.sizes = []
        for i, g in enumerate(graphs):
            k = [np.ascontiguousarray(g["pose_fixed"], np.uint8), np.ascontiguousarray(g["edge_pose"], np.int32),
                 np.ascontiguousarray(g["edge_point"], np.int32), np.ascontiguousarray(g["edge_obs"], np.float64),
                 np.ascontiguousarray(g["edge_inv_sigma2"], np.float64), np.ascontiguousarray(g["edge_stereo"], np.uint8)]
            self._keep.append(k)
            kb = g.get("kb")                      # KannalaBrandt8 k1..k4 for the monocular edges, None = Pinhole
            ncam, cam_arr, pcam = camera_table(g)
            self._keep_cam = getattr(self, "_keep_cam", []) + [(cam_arr, pcam)]
            arr[i] = BaGraph(g["n_poses"], g["n_points"], g["n_edges"], *[a.ctypes.data for a in k],
                             g["fx"], g["fy"], g["cx"], g["cy"], g["bf"], 1 if kb is not None else 0,
                             (cd * 4)(*(kb if kb is not None else (0, 0, 0, 0))), *rig2_fields(g),
                             ncam, C.cast(cam_arr, vp) if ncam else None, pcam.ctypes.data if ncam else None)
            self.sizes.append((g["n_poses"], g["n_points"], g["n_edges"]))
        self.poses = [np.ascontiguousarray(g["poses0"], np.float64).copy() for g in graphs]
        self.points = [np.ascontiguousarray(g["points0"], np.float64).copy() for g in graphs]
        pp = (vp * self.n)(*[a.ctypes.data for a in self.poses])
        pq = (vp * self.n)(*[a.ctypes.data for a in self.points])
        h = vp()
        if world == 1:
            _chk(lib.orbhip_ba_batch_create(ctx.h, C.cast(arr, vp), self.n, C.cast(pp, vp), C.cast(pq, vp), C.byref(h)),
                 "orbhip_ba_batch_create")
        else:
            _chk(lib.orbhip_ba_batch_create_sharded(ctx.h, C.cast(arr, vp), self.n, C.cast(pp, vp), C.cast(pq, vp), rank, world,
                                                    C.byref(h)), "orbhip_ba_batch_create_sharded")
        self.h = h

    @property
    def exchange_doubles(self):
        """Doubles per rank slot of the exchange buffer (sharded batches)."""
        return lib.orbhip_ba_batch_exchange_doubles(self.h)

    def set_exchange_buffer(self, d_ptr, capacity_doubles):
        _chk(lib.orbhip_ba_batch_set_exchange_buffer(self.h, d_ptr, capacity_doubles), "orbhip_ba_batch_set_exchange_buffer")

    def solve_sharded(self, exchange, params=None, abort=None):
        """exchange(stage, count) -> None: all-gather `count` doubles per rank inside the exchange buffer (slot r = rank r)."""
        p = params or ba_default_params()
        err = []

        def _cb(user, stage, count):
            try:
                exchange(int(stage), int(count))
                return 0
            except BaseException as e:          # never let an exception cross the C boundary
                err.append(e)
                return 1
        cb = BA_EXCHANGE_FN(_cb)
        rc = lib.orbhip_ba_batch_solve_sharded(self.h, C.byref(p), abort.ctypes.data if abort is not None else None, cb, None)
        if err:
            raise err[0]
        if rc not in (OK, E_ABORTED):
            raise OrbHipError(rc, "orbhip_ba_batch_solve_sharded")
        return rc

    def solve(self, params=None, abort=None):
        p = params or ba_default_params()
        rc = lib.orbhip_ba_batch_solve(self.h, C.byref(p), abort.ctypes.data if abort is not None else None)
        if rc not in (OK, E_ABORTED):
            raise OrbHipError(rc, "orbhip_ba_batch_solve")
        return rc

    @property
    def ticks(self):
        return lib.orbhip_ba_batch_ticks(self.h)

    def set_graph_mode(self, on):
        _chk(lib.orbhip_extractor_set_graph_mode(self.h, 1 if on else 0), "orbhip_extractor_set_graph_mode")

    def set_profiling(self, on):
        _chk(lib.orbhip_ba_batch_set_profiling(self.h, 1 if on else 0), "ba set_profiling")

    def gemm_profile(self):
        ms, n, fl = cf(), ci(), cd()
        _chk(lib.orbhip_ba_batch_gemm_profile(self.h, C.byref(ms), C.byref(n), C.byref(fl)), "ba gemm_profile")
        return ms.value, n.value, fl.value

    def gemm_dense_flops(self):
        return lib.orbhip_ba_batch_gemm_dense_flops(self.h)

    def gemm_issued_flops(self):
        return lib.orbhip_ba_batch_gemm_issued_flops(self.h)

    def download(self):
        poses = [a.copy() for a in self.poses]
        points = [a.copy() for a in self.points]
        outl = [np.zeros(max(s[2], 1), np.uint8) for s in self.sizes]
        stats = (BaStats * self.n)()
        pp = (vp * self.n)(*[a.ctypes.data for a in poses])
        pq = (vp * self.n)(*[a.ctypes.data for a in points])
        po = (vp * self.n)(*[a.ctypes.data for a in outl])
        _chk(lib.orbhip_ba_batch_download(self.h, C.cast(pp, vp), C.cast(pq, vp), C.cast(po, vp), C.cast(stats, vp)),
             "orbhip_ba_batch_download")
        return poses, points, [o[:s[2]] for o, s in zip(outl, self.sizes)], [st.as_dict() for st in stats]

    def close(self):
        if self.h:
            lib.orbhip_ba_batch_destroy(self.h)
            self.h = None


# ---------------------------------------------------------------------------------------------- inertial local BA
class IbaWindow(C.Structure):
    """orbhip_iba_window (include/orbhip.h): the flat description of one Optimizer::LocalInertialBA window."""
    _fields_ = [("n_kf", C.c_int32), ("kf_fixed", vp), ("kf_imu", vp), ("Rcb", cd * 9), ("tcb", cd * 3),
                ("fx", cd), ("fy", cd), ("cx", cd), ("cy", cd), ("bf", cd),
                ("camera_model", C.c_int32), ("kb", cd * 4), ("has_cam2", C.c_int32), ("Trl", cd * 12),
                ("fx2", cd), ("fy2", cd), ("cx2", cd), ("cy2", cd), ("camera2_model", C.c_int32), ("kb2", cd * 4),
                ("n_points", C.c_int32), ("n_edges", C.c_int32), ("edge_kf", vp), ("edge_point", vp), ("edge_obs", vp),
                ("edge_stereo", vp), ("edge_inv_sigma2", vp), ("edge_close", vp),
                ("n_inertial", C.c_int32), ("in_kf1", vp), ("in_kf2", vp), ("in_preint", vp), ("in_info", vp),
                ("in_info_g", vp), ("in_info_a", vp), ("in_robust", vp)]


class IbaParams(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("lambda_init", cd), ("large", C.c_int32), ("max_trials", C.c_int32)]


class IbaStats(C.Structure):
    _fields_ = [("iterations_run", C.c_int32), ("lm_trials", C.c_int32), ("n_outliers", C.c_int32), ("failed", C.c_int32),
                ("err", cd), ("err_end", cd)]

    def as_dict(self):
        return dict(iterations_run=self.iterations_run, lm_trials=self.lm_trials, n_outliers=self.n_outliers, failed=self.failed,
                    err=self.err, err_end=self.err_end)


IBA_KF, IBA_PREINT = 21, 67
lib.orbhip_iba_default_params.argtypes = [C.POINTER(IbaParams), ci]
lib.orbhip_inertial_ba_solve_batch.argtypes = [vp, vp, ci, C.POINTER(IbaParams), vp, vp, vp, vp]


def iba_default_params(large=False):
    p = IbaParams()
    lib.orbhip_iba_default_params(C.byref(p), 1 if large else 0)
    return p


def inertial_ba_solve_batch(ctx, windows, kf_states, points, params=None):
    """windows: list of IbaWindow (their arrays kept alive by the caller); kf_states[w] float64 [n_kf, 21], points[w] float64
    [n_points, 3] -> (kf_states, points, edge_outlier, stats) as new arrays (inputs untouched)."""
    n = len(windows)
    p = params or iba_default_params()
    arr = (IbaWindow * n)(*windows)
    kfs = [np.ascontiguousarray(a, np.float64).copy() for a in kf_states]
    pts = [np.ascontiguousarray(a, np.float64).copy() for a in points]
    outl = [np.zeros(max(w.n_edges, 1), np.uint8) for w in windows]
    stats = (IbaStats * n)()
    pk = (vp * n)(*[a.ctypes.data for a in kfs])
    pq = (vp * n)(*[a.ctypes.data for a in pts])
    po = (vp * n)(*[a.ctypes.data for a in outl])
    _chk(lib.orbhip_inertial_ba_solve_batch(ctx.h, C.cast(arr, vp), n, C.byref(p), C.cast(pk, vp), C.cast(pq, vp), C.cast(po, vp),
                                            C.cast(stats, vp)), "orbhip_inertial_ba_solve_batch")
    return kfs, pts, [o[:w.n_edges] for o, w in zip(outl, windows)], [st.as_dict() for st in stats]


lib.orbhip_iba_batch_create.argtypes = [vp, vp, ci, vp, vp, C.POINTER(vp)]
lib.orbhip_iba_batch_set_states.argtypes = [vp, vp, vp]
lib.orbhip_iba_batch_solve.argtypes = [vp, C.POINTER(IbaParams)]
lib.orbhip_iba_batch_download.argtypes = [vp, vp, vp, vp, vp]
lib.orbhip_iba_batch_destroy.argtypes = [vp]
lib.orbhip_iba_batch_destroy.restype = None


class IbaBatch:
    """orbhip_iba_batch: windows resident on the device; solve() re-solves them from the states last set."""

    def __init__(self, ctx, windows, kf_states, points):
        self.n = len(windows)
        self.shapes = [(w.n_kf, w.n_points, w.n_edges) for w in windows]
        arr = (IbaWindow * self.n)(*windows)
        kfs = [np.ascontiguousarray(a, np.float64) for a in kf_states]
        pts = [np.ascontiguousarray(a, np.float64) for a in points]
        pk = (vp * self.n)(*[a.ctypes.data for a in kfs])
        pq = (vp * self.n)(*[a.ctypes.data for a in pts])
        self.h = vp()
        _chk(lib.orbhip_iba_batch_create(ctx.h, C.cast(arr, vp), self.n, C.cast(pk, vp), C.cast(pq, vp), C.byref(self.h)),
             "orbhip_iba_batch_create")

    def set_states(self, kf_states, points):
        kfs = [np.ascontiguousarray(a, np.float64) for a in kf_states]
        pts = [np.ascontiguousarray(a, np.float64) for a in points]
        pk = (vp * self.n)(*[a.ctypes.data for a in kfs])
        pq = (vp * self.n)(*[a.ctypes.data for a in pts])
        _chk(lib.orbhip_iba_batch_set_states(self.h, C.cast(pk, vp), C.cast(pq, vp)), "orbhip_iba_batch_set_states")

    def solve(self, params=None):
        p = params or iba_default_params()
        _chk(lib.orbhip_iba_batch_solve(self.h, C.byref(p)), "orbhip_iba_batch_solve")

    def download(self):
        """-> (kf_states, points, edge_outlier, stats); a failed window's states come back as zeros (not written, as the one-shot call
        leaves its inputs alone)."""
        kfs = [np.zeros((s[0], IBA_KF)) for s in self.shapes]
        pts = [np.zeros((max(s[1], 1), 3)) for s in self.shapes]
        outl = [np.zeros(max(s[2], 1), np.uint8) for s in self.shapes]
        stats = (IbaStats * self.n)()
        pk = (vp * self.n)(*[a.ctypes.data for a in kfs])
        pq = (vp * self.n)(*[a.ctypes.data for a in pts])
        po = (vp * self.n)(*[a.ctypes.data for a in outl])
        _chk(lib.orbhip_iba_batch_download(self.h, C.cast(pk, vp), C.cast(pq, vp), C.cast(po, vp), C.cast(stats, vp)),
             "orbhip_iba_batch_download")
        return (kfs, [p[:s[1]] for p, s in zip(pts, self.shapes)], [o[:s[2]] for o, s in zip(outl, self.shapes)],
                [st.as_dict() for st in stats])

    def close(self):
        if self.h:
            lib.orbhip_iba_batch_destroy(self.h)
            self.h = None


def inertial_ba_last_team_size():
    """Workgroups per window of this thread's latest inertial solve (1 = no device-wide barrier)."""
    return int(lib.orbhip_inertial_ba_last_team_size())


class Camera2(C.Structure):
    _fields_ = [("Trl", cd * 7), ("fx", cd), ("fy", cd), ("cx", cd), ("cy", cd), ("camera_model", C.c_int32), ("kb", cd * 4)]


lib.orbhip_pose_optimization_device.argtypes = [vp, vp, vp, vp, vp, ci, ci, cd, cd, cd, cd, cd, vp, vp, vp, vp, vp, vp, vp]


def pose_optimization_device(ctx, d_Xw, d_obs, d_inv_sigma2, d_n_edges, frames, max_edges, cam, d_pose, d_outlier,
                             d_n_inliers, d_stats=None, kb8=None, rig2=None, d_right=None):
    """Optimizer::PoseOptimization, batched over frames; device addresses (ints); cam = (fx, fy, cx, cy, bf);
    kb8 = (k1..k4) for a KannalaBrandt8 camera (host values), None = Pinhole; rig2 = dict(Trl, cam, kb) + d_right
    (device address of per-edge flags) when some observations were made in a second, rigidly attached camera."""
    kb = (cd * 4)(*kb8) if kb8 is not None else None
    c2 = None
    if rig2 is not None:
        k2 = rig2.get("kb")
        c2 = C.byref(Camera2((cd * 7)(*rig2["Trl"]), *[float(c) for c in rig2["cam"]], 1 if k2 is not None else 0,
                             (cd * 4)(*(k2 if k2 is not None else (0, 0, 0, 0)))))
    _chk(lib.orbhip_pose_optimization_device(ctx.h, d_Xw, d_obs, d_inv_sigma2, d_n_edges, frames, max_edges,
                                             float(cam[0]), float(cam[1]), float(cam[2]), float(cam[3]), float(cam[4]),
                                             kb, c2, d_right, d_pose, d_outlier, d_n_inliers, d_stats), "orbhip_pose_optimization_device")


lib.orbhip_compute_stereo_matches_device.argtypes = [vp, vp, cf, cf, vp, vp, vp]


def compute_stereo_matches_device(ext_left, ext_right, mb, mbf, d_u_right, d_depth, d_n_matches=None):
    """Frame::ComputeStereoMatches on the latest results of two extractors; device addresses (ints)."""
    _chk(lib.orbhip_compute_stereo_matches_device(ext_left.h, ext_right.h, mb, mbf, d_u_right, d_depth, d_n_matches),
         "orbhip_compute_stereo_matches_device")


lib.orbhip_distinctive_descriptors_device.argtypes = [vp, vp, vp, ci, ci, vp, vp]


def distinctive_descriptors_device(ctx, d_desc, d_n, points, max_n, d_best_idx, d_best_desc=None):
    """MapPoint::ComputeDistinctiveDescriptors, batched over map points; device addresses (ints)."""
    _chk(lib.orbhip_distinctive_descriptors_device(ctx.h, d_desc, d_n, points, max_n, d_best_idx, d_best_desc),
         "orbhip_distinctive_descriptors_device")


lib.orbhip_bow_transform_device.argtypes = [vp, vp, vp, ci, ci, sz, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp]


def bow_transform_device(ctx, d_desc, d_n, frames, max_n, frame_stride, voc, L, levelsup, d_word_id, d_weight, d_nid):
    """DBoW2 per-feature tree descent; voc = (d_node_desc, d_child_start, d_child_ids, d_node_word, d_node_weight)."""
    _chk(lib.orbhip_bow_transform_device(ctx.h, d_desc, d_n, frames, max_n, frame_stride, voc[0], voc[1], voc[2], voc[3], voc[4],
                                         L, levelsup, d_word_id, d_weight, d_nid), "orbhip_bow_transform_device")


lib.orbhip_fuse_search_device.argtypes = [vp, vp, vp, vp, ci, vp, vp, vp, vp, ci, sz, ci, vp, ci, cf, cf, cf, cf, vp, vp]


def fuse_search_device(ctx, d_q, d_desc_q, d_nq, max_q, d_kp, d_desc, d_u_right, d_n, max_n, kp_stride, pairs, inv_level_sigma2,
                       bounds, d_best_idx, d_best_dist):
    """Search part of ORBmatcher::Fuse, batched; device addresses (ints); inv_level_sigma2 = host float array."""
    sig = np.ascontiguousarray(inv_level_sigma2, np.float32)
    _chk(lib.orbhip_fuse_search_device(ctx.h, d_q, d_desc_q, d_nq, max_q, d_kp, d_desc, d_u_right, d_n, max_n, kp_stride, pairs,
                                       sig.ctypes.data, len(sig), bounds[0], bounds[1], bounds[2], bounds[3], d_best_idx, d_best_dist),
         "orbhip_fuse_search_device")


lib.orbhip_search_by_bow_kf_device.argtypes = [vp] * 17 + [ci, ci, ci, sz, cf, ci, vp, vp]


def search_by_bow_kf_device(ctx, kf1, kf2, pairs, max_nodes, max_n, kp_stride, nn_ratio, check_ori, d_matches12, d_nmatches):
    """ORBmatcher::SearchByBoW(KeyFrame, KeyFrame), batched.  kf1 / kf2 = (d_node_ids, d_node_start, d_feat, d_nnodes, d_valid, d_kp,
    d_desc, d_n): device addresses (ints)."""
    _chk(lib.orbhip_search_by_bow_kf_device(ctx.h, *kf1, *kf2, pairs, max_nodes, max_n, kp_stride, nn_ratio, 1 if check_ori else 0,
                                            d_matches12, d_nmatches), "orbhip_search_by_bow_kf_device")


lib.orbhip_undistort_keypoints_device.argtypes = [vp, vp, vp, ci, ci, sz, cf, cf, cf, cf, vp, ci, vp]
lib.orbhip_assign_features_to_grid_device.argtypes = [vp, vp, vp, ci, ci, sz, cf, cf, cf, cf, vp, vp]


def undistort_keypoints_device(ctx, d_kp, d_n, frames, max_n, kp_stride, K, dist, d_kp_un):
    """Frame::UndistortKeyPoints, batched.  K = (fx, fy, cx, cy); dist: 4 or 5 host floats."""
    dc = np.ascontiguousarray(dist, np.float32)
    _chk(lib.orbhip_undistort_keypoints_device(ctx.h, d_kp, d_n, frames, max_n, kp_stride, K[0], K[1], K[2], K[3], dc.ctypes.data, len(dc),
                                               d_kp_un), "orbhip_undistort_keypoints_device")


def assign_features_to_grid_device(ctx, d_kp, d_n, frames, max_n, kp_stride, bounds, d_cell_start, d_items):
    """Frame::AssignFeaturesToGrid as a CSR per frame (cell ix*48+iy)."""
    _chk(lib.orbhip_assign_features_to_grid_device(ctx.h, d_kp, d_n, frames, max_n, kp_stride, bounds[0], bounds[1], bounds[2], bounds[3],
                                                   d_cell_start, d_items), "orbhip_assign_features_to_grid_device")


lib.orbhip_bow_vectors_device.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]


def bow_vectors_device(ctx, d_wid, d_w, d_nid, d_n, frames, max_n, max_nodes, d_node_ids, d_node_start, d_feat, d_nnodes, d_bow_word,
                       d_bow_value, d_nwords):
    """mFeatVec (CSR) and mBowVec (sorted, L1-normalised) from bow_transform_device's per-feature output."""
    _chk(lib.orbhip_bow_vectors_device(ctx.h, d_wid, d_w, d_nid, d_n, frames, max_n, max_nodes, d_node_ids, d_node_start, d_feat, d_nnodes,
                                       d_bow_word, d_bow_value, d_nwords), "orbhip_bow_vectors_device")


TRI_PAIR_DTYPE = np.dtype([("F12", "<f4", (9,)), ("ep_x", "<f4"), ("ep_y", "<f4"), ("only_stereo", "<i4"), ("coarse", "<i4")])
lib.orbhip_search_for_triangulation_device.argtypes = [vp] * 17 + [ci, ci, ci, sz, vp, vp, ci, ci, vp, vp]


def search_for_triangulation_device(ctx, kf1, kf2, d_pair, pairs, max_nodes, max_n, kp_stride, scale_factors, level_sigma2,
                                    check_ori, d_matches12, d_nmatches):
    """ORBmatcher::SearchForTriangulation, batched.  kf1 = (d_nid, d_has_mp, d_kp, d_desc, d_u_right|0, d_n),
    kf2 = (d_node_ids, d_node_start, d_feat, d_nnodes, d_has_mp, d_kp, d_desc, d_u_right|0, d_n): device addresses (ints);
    d_pair: [pairs] TRI_PAIR_DTYPE records; scale_factors / level_sigma2: host float32 arrays."""
    sf = np.ascontiguousarray(scale_factors, np.float32); ls = np.ascontiguousarray(level_sigma2, np.float32)
    assert len(sf) == len(ls)
    _chk(lib.orbhip_search_for_triangulation_device(ctx.h, *[a or None for a in kf1], *[a or None for a in kf2], d_pair, pairs, max_nodes,
                                                    max_n, kp_stride, sf.ctypes.data, ls.ctypes.data, len(sf), 1 if check_ori else 0,
                                                    d_matches12, d_nmatches), "orbhip_search_for_triangulation_device")


TRI_GENERAL_DTYPE = np.dtype([("R12", "<f4", (4, 9)), ("t12", "<f4", (4, 3)), ("F12", "<f4", (4, 9)), ("cam1", "<f4", (2, 8)), ("cam2", "<f4", (2, 8)),
                              ("cam1_type", "<i4", (2,)), ("cam2_type", "<i4", (2,)), ("ep_x", "<f4"), ("ep_y", "<f4"), ("nleft1", "<i4"), ("nleft2", "<i4"),
                              ("only_stereo", "<i4"), ("coarse", "<i4")])
lib.orbhip_search_for_triangulation_general_device.argtypes = [vp] * 17 + [ci, ci, ci, sz, vp, vp, vp, ci, ci, vp, vp]


def search_for_triangulation_general_device(ctx, kf1, kf2, d_pair, pairs, max_nodes, max_n, kp_stride, level_sigma2_1, scale_factors2,
                                            level_sigma2_2, check_ori, d_matches12, d_nmatches):
    """ORBmatcher::SearchForTriangulation for every camera combination (Pinhole / KannalaBrandt8 / two-camera rigs), batched.  kf1 / kf2 as
    in search_for_triangulation_device; d_pair: [pairs] TRI_GENERAL_DTYPE records."""
    a = [np.ascontiguousarray(x, np.float32) for x in (level_sigma2_1, scale_factors2, level_sigma2_2)]
    assert len(a[0]) == len(a[1]) == len(a[2])
    _chk(lib.orbhip_search_for_triangulation_general_device(ctx.h, *[x or None for x in kf1], *[x or None for x in kf2], d_pair, pairs, max_nodes,
                                                            max_n, kp_stride, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, len(a[0]),
                                                            1 if check_ori else 0, d_matches12, d_nmatches),
         "orbhip_search_for_triangulation_general_device")


TRI_POSES_DTYPE = np.dtype([("Tcw1", "<f4", (2, 12)), ("Tcw2", "<f4", (2, 12))])
lib.orbhip_match_and_triangulate_device.argtypes = [vp] * 16 + [ci, ci, ci, sz, vp, vp, ci, ci, vp, vp, vp]


def match_and_triangulate_device(ctx, kf1, kf2, d_pair, d_poses, pairs, max_nodes, max_n, kp_stride, level_sigma2_1, level_sigma2_2, check_ori,
                                 d_matches12, d_points12, d_nmatches):
    """ORBmatcher::SearchForTriangulation(..., vMatchedPoints) (ORBmatcher.cc:1212-1402), batched.  kf1 = [nid1, has_mp1, kp1, desc1, n1],
    kf2 = [node_ids2, node_start2, feat2, nnodes2, has_mp2, kp2, desc2, n2] device pointers; d_poses: [pairs] TRI_POSES_DTYPE records."""
    a = [np.ascontiguousarray(x, np.float32) for x in (level_sigma2_1, level_sigma2_2)]
    assert len(a[0]) == len(a[1]) and len(kf1) == 5 and len(kf2) == 8
    _chk(lib.orbhip_match_and_triangulate_device(ctx.h, *kf1, *kf2, d_pair, d_poses, pairs, max_nodes, max_n, kp_stride, a[0].ctypes.data,
                                                 a[1].ctypes.data, len(a[0]), 1 if check_ori else 0, d_matches12, d_points12, d_nmatches),
         "orbhip_match_and_triangulate_device")


lib.orbhip_search_by_bow_device.argtypes = [vp] * 15 + [ci, ci, ci, sz, cf, ci, vp, vp]


def search_by_bow_device(ctx, kf, f, d_nF, pairs, max_nodes, max_n, kp_stride, nn_ratio, check_ori, d_match_f, d_nmatches):
    """ORBmatcher::SearchByBoW(KeyFrame, Frame), batched.  kf = (d_node_ids, d_node_start, d_feat, d_nnodes, d_valid, d_kp, d_desc),
    f = (d_node_ids, d_node_start, d_feat, d_nnodes, d_kp, d_desc): device addresses (ints)."""
    _chk(lib.orbhip_search_by_bow_device(ctx.h, kf[0], kf[1], kf[2], kf[3], kf[4], kf[5], kf[6], f[0], f[1], f[2], f[3], f[4], f[5], d_nF,
                                         pairs, max_nodes, max_n, kp_stride, nn_ratio, 1 if check_ori else 0, d_match_f, d_nmatches),
         "orbhip_search_by_bow_device")


lib.orbhip_search_by_projection_rig_device.argtypes = [vp, ci, vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, sz, ci, cf, cf, cf, cf, ci, cf, ci, vp, vp]
lib.orbhip_search_by_bow_rig_device.argtypes = [vp] * 16 + [ci, ci, ci, sz, cf, ci, vp, vp]
lib.orbhip_assign_features_to_grid_rig_device.argtypes = [vp, vp, vp, vp, ci, ci, sz, cf, cf, cf, cf, vp, vp]


def search_by_projection_rig_device(ctx, mode, d_q, d_desc_q, d_nq, max_q, d_kp, d_desc, d_n, d_nleft, d_mirror, max_n, kp_stride, pairs, bounds,
                                    th_high, nn_ratio, check_ori, d_train_match, d_nmatches):
    """ORBmatcher::SearchByProjection on frames of a two-camera rig (Nleft != -1); mode 0: from the last frame, 1: local map points."""
    _chk(lib.orbhip_search_by_projection_rig_device(ctx.h, mode, d_q, d_desc_q, d_nq, max_q, d_kp, d_desc, d_n, d_nleft, d_mirror, max_n, kp_stride,
                                                    pairs, bounds[0], bounds[1], bounds[2], bounds[3], th_high, nn_ratio, 1 if check_ori else 0,
                                                    d_train_match, d_nmatches), "orbhip_search_by_projection_rig_device")


def search_by_bow_rig_device(ctx, kf, f, d_nF, d_nleft, pairs, max_nodes, max_n, kp_stride, nn_ratio, check_ori, d_match_f, d_nmatches):
    """ORBmatcher::SearchByBoW(KeyFrame, Frame) on rig frames; kf / f as in search_by_bow_device."""
    _chk(lib.orbhip_search_by_bow_rig_device(ctx.h, kf[0], kf[1], kf[2], kf[3], kf[4], kf[5], kf[6], f[0], f[1], f[2], f[3], f[4], f[5], d_nF, d_nleft,
                                             pairs, max_nodes, max_n, kp_stride, nn_ratio, 1 if check_ori else 0, d_match_f, d_nmatches),
         "orbhip_search_by_bow_rig_device")


def assign_features_to_grid_rig_device(ctx, d_kp, d_n, d_nleft, frames, max_n, kp_stride, bounds, d_cell_start, d_items):
    """Frame::AssignFeaturesToGrid on rig frames: mGrid | mGridRight as one CSR (2*3072+1 cell starts per frame)."""
    _chk(lib.orbhip_assign_features_to_grid_rig_device(ctx.h, d_kp, d_n, d_nleft, frames, max_n, kp_stride, bounds[0], bounds[1], bounds[2],
                                                       bounds[3], d_cell_start, d_items), "orbhip_assign_features_to_grid_rig_device")
