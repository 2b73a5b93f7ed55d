"""ctypes binding of the CPU ORACLE (oracle/liborb_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
the product package."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("ORB_ORACLE_LIB") or os.path.join(ROOT, "oracle", "liborb_oracle.so")      # (override: the sanitizer build, oracle/Makefile `asan`)
lib = C.CDLL(LIB)

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
vp, ci, cf = C.c_void_p, C.c_int, C.c_float
PI, PF = C.POINTER(ci), C.POINTER(cf)

lib.orc_extractor_create.restype = vp
lib.orc_extractor_create.argtypes = [ci, cf, ci, ci, ci]
lib.orc_extractor_destroy.argtypes = [vp]
for name, rt in [("orc_scale_factors", PF), ("orc_inv_scale_factors", PF), ("orc_level_sigma2", PF),
                 ("orc_inv_level_sigma2", PF), ("orc_features_per_level", PI), ("orc_umax", PI)]:
    getattr(lib, name).restype = rt
    getattr(lib, name).argtypes = [vp]
lib.orc_extract.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, ci, PI]
lib.orc_pyramid_level.restype = C.POINTER(C.c_uint8)
lib.orc_pyramid_level.argtypes = [vp, ci, PI, PI, PI]
lib.orc_pyramid_level_padded.restype = C.POINTER(C.c_uint8)
lib.orc_pyramid_level_padded.argtypes = [vp, ci, PI, PI, PI]
lib.orc_blurred_level.restype = C.POINTER(C.c_uint8)
lib.orc_blurred_level.argtypes = [vp, ci, PI, PI]
lib.orc_fast_candidates.argtypes = [vp, ci, vp, vp, vp, ci]
lib.orc_level_keypoints.argtypes = [vp, ci, vp, ci]
lib.orc_cell_grid.argtypes = [ci, ci, PI, PI, PI, PI]
lib.orc_resize_linear.argtypes = [vp, ci, ci, ci, vp, ci, ci, ci]
lib.orc_gaussian_blur7.argtypes = [vp, ci, ci, ci, vp, ci]
lib.orc_fast_nms.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, ci]
lib.orc_fast_arc_score.argtypes = [vp, ci]
lib.orc_fast_atan2.restype = cf
lib.orc_fast_atan2.argtypes = [cf, cf]
lib.orc_ic_angle.restype = cf
lib.orc_ic_angle.argtypes = [vp, ci, ci, ci, vp]
lib.orc_sincos_deg.argtypes = [cf, PF, PF]
lib.orc_descriptor.argtypes = [vp, ci, ci, ci, cf, vp]
lib.orc_octree.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, ci]


def _view(ptr, h, stride, w):
    arr = np.ctypeslib.as_array(ptr, shape=(h * stride,))
    return np.lib.stride_tricks.as_strided(arr, shape=(h, w), strides=(stride, 1)).copy()


class OracleExtractor:
    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7):
        self.h = lib.orc_extractor_create(nfeatures, scale, nlevels, ini_th, min_th)
        self.nlevels = nlevels
        self.nfeatures = nfeatures

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.orc_extractor_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        return dict(scale=np.array([lib.orc_scale_factors(self.h)[i] for i in range(n)], np.float32),
                    inv_scale=np.array([lib.orc_inv_scale_factors(self.h)[i] for i in range(n)], np.float32),
                    sigma2=np.array([lib.orc_level_sigma2(self.h)[i] for i in range(n)], np.float32),
                    inv_sigma2=np.array([lib.orc_inv_level_sigma2(self.h)[i] for i in range(n)], np.float32),
                    per_level=np.array([lib.orc_features_per_level(self.h)[i] for i in range(n)], np.int32),
                    umax=np.array([lib.orc_umax(self.h)[i] for i in range(16)], np.int32))

    def extract(self, img, lap=(0, 1000)):
        img = np.ascontiguousarray(img, np.uint8)
        H, W = img.shape
        # the octree's first pass splits all nIni = round(W/H) roots: up to 4 * nIni keypoints per level whatever the budget
        cap = self.nfeatures + (64 + 4 * (W // max(H - 32, 1) + 2)) * self.nlevels
        kp = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = ci()
        mono = lib.orc_extract(self.h, img.ctypes.data, W, H, W, lap[0], lap[1], kp.ctypes.data, desc.ctypes.data,
                               cap, C.byref(n))
        assert mono >= -1, mono
        return kp[:n.value].copy(), desc[:n.value].copy(), mono

    def pyramid_level(self, level, padded=False):
        w, h, s = ci(), ci(), ci()
        f = lib.orc_pyramid_level_padded if padded else lib.orc_pyramid_level
        p = f(self.h, level, C.byref(w), C.byref(h), C.byref(s))
        return _view(p, h.value, s.value, w.value)

    def blurred_level(self, level):
        w, h = ci(), ci()
        p = lib.orc_blurred_level(self.h, level, C.byref(w), C.byref(h))
        if not p:
            return None
        return _view(p, h.value, w.value, w.value)

    def fast_candidates(self, level):
        n = lib.orc_fast_candidates(self.h, level, None, None, None, 0)
        xs, ys, ss = (np.zeros(max(n, 1), np.int32) for _ in range(3))
        lib.orc_fast_candidates(self.h, level, xs.ctypes.data, ys.ctypes.data, ss.ctypes.data, n)
        return xs[:n], ys[:n], ss[:n]

    def level_keypoints(self, level):
        n = lib.orc_level_keypoints(self.h, level, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        lib.orc_level_keypoints(self.h, level, out.ctypes.data, n)
        return out[:n]


def cell_grid(w, h):
    a, b, c, d = ci(), ci(), ci(), ci()
    lib.orc_cell_grid(w, h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
    return a.value, b.value, c.value, d.value


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw = src.shape
    dst = np.zeros((dh, dw), np.uint8)
    lib.orc_resize_linear(src.ctypes.data, sw, sh, sw, dst.ctypes.data, dw, dh, dw)
    return dst


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    dst = np.zeros((h, w), np.uint8)
    lib.orc_gaussian_blur7(src.ctypes.data, w, h, w, dst.ctypes.data, w)
    return dst


def fast_nms(img, th):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = w * h
    xs, ys, ss = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib.orc_fast_nms(img.ctypes.data, w, h, w, th, xs.ctypes.data, ys.ctypes.data, ss.ctypes.data, cap)
    return xs[:n], ys[:n], ss[:n]


def fast_arc_score(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    return lib.orc_fast_arc_score(img.ctypes.data + y * w + x, w)


def sincos_deg(a):
    c, s = cf(), cf()
    lib.orc_sincos_deg(a, C.byref(c), C.byref(s))
    return c.value, s.value


def octree(xs, ys, ss, min_x, max_x, min_y, max_y, n_features):
    xs, ys, ss = (np.ascontiguousarray(a, np.int32) for a in (xs, ys, ss))
    keep = np.zeros(len(xs) + 8, np.int32)
    n = lib.orc_octree(xs.ctypes.data, ys.ctypes.data, ss.ctypes.data, len(xs), min_x, max_x, min_y, max_y,
                       n_features, keep.ctypes.data, len(keep))
    return keep[:n]


# ------------------------------------------------------------------ Frame::ComputeStereoMatches oracle
lib.orc_compute_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, ci, C.c_void_p, C.c_void_p, ci,
                                           C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
lib.orc_compute_stereo_matches.restype = ci


def compute_stereo_matches(eL, eR, kpL, dL, kpR, dR, mb, mbf):
    """eL/eR: OracleExtractor objects that have just extracted the left/right image.  Returns (kept, uRight, depth, sad)."""
    kpL = np.ascontiguousarray(kpL, KP_DTYPE); kpR = np.ascontiguousarray(kpR, KP_DTYPE)
    dL = np.ascontiguousarray(dL, np.uint8); dR = np.ascontiguousarray(dR, np.uint8)
    n = len(kpL)
    ur = np.zeros(max(n, 1), np.float32); dp = np.zeros(max(n, 1), np.float32); sad = np.zeros(max(n, 1), np.int32)
    k = lib.orc_compute_stereo_matches(eL.h, eR.h, kpL.ctypes.data, dL.ctypes.data, n, kpR.ctypes.data, dR.ctypes.data, len(kpR),
                                       mb, mbf, ur.ctypes.data, dp.ctypes.data, sad.ctypes.data)
    return k, ur[:n], dp[:n], sad[:n]
