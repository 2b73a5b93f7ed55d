"""ctypes binding of the inertial local-BA ORACLE (oracle/iba_oracle.c) + the synthetic visual-inertial windows the CPU and GPU
tests share.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import numpy as np
from oracle_bind import lib

vp, ci, cd = C.c_void_p, C.c_int, C.c_double


class Problem(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("kf_fixed", vp), ("kf_imu", vp), ("Rcb", cd * 9), ("tcb", cd * 3),
                ("fx", cd), ("fy", cd), ("cx", cd), ("cy", cd), ("bf", cd),
                ("camera_model", C.c_int32), ("kb", cd * 4), ("has_cam2", C.c_int32), ("Trl", cd * 12),
                ("fx2", cd), ("fy2", cd), ("cx2", cd), ("cy2", cd), ("camera2_model", C.c_int32), ("kb2", cd * 4),
                ("n_points", C.c_int32), ("n_edges", C.c_int32), ("edge_kf", vp), ("edge_point", vp), ("edge_obs", vp),
                ("edge_stereo", vp), ("edge_inv_sigma2", vp), ("edge_close", vp),
                ("n_inertial", C.c_int32), ("in_kf1", vp), ("in_kf2", vp), ("in_preint", vp), ("in_info", vp),
                ("in_info_g", vp), ("in_info_a", vp), ("in_robust", vp)]


class Params(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("lambda_init", cd), ("large", C.c_int32), ("max_trials", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("iterations_run", C.c_int32), ("lm_trials", C.c_int32), ("n_outliers", C.c_int32), ("failed", C.c_int32),
                ("err", cd), ("err_end", cd)]


lib.orc_iba_default_params.argtypes = [vp, ci]
lib.orc_iba_solve.argtypes = [vp, vp, vp, vp, vp, vp]
lib.orc_iba_solve.restype = ci
lib.orc_iba_kf_update.argtypes = [vp, vp, ci]
lib.orc_iba_edge_inertial.argtypes = [vp, vp, vp, vp, vp]
lib.orc_iba_edge_visual.argtypes = [vp, vp, vp, vp, ci, vp, vp, vp]
lib.orc_iba_exp_so3.argtypes = [vp, vp]
lib.orc_iba_log_so3.argtypes = [vp, vp]


def default_params(large=False):
    p = Params()
    lib.orc_iba_default_params(C.addressof(p), 1 if large else 0)
    return p


def exp_so3(w):
    w = np.ascontiguousarray(w, np.float64)
    R = np.zeros(9)
    lib.orc_iba_exp_so3(w.ctypes.data, R.ctypes.data)
    return R.reshape(3, 3)


def log_so3(R):
    R = np.ascontiguousarray(R, np.float64)
    w = np.zeros(3)
    lib.orc_iba_log_so3(R.ctypes.data, w.ctypes.data)
    return w


def kf_update(s, dx, imu=True):
    s = np.ascontiguousarray(s, np.float64).copy()
    dx = np.ascontiguousarray(dx, np.float64)
    lib.orc_iba_kf_update(s.ctypes.data, dx.ctypes.data, 1 if imu else 0)
    return s


def edge_inertial(s1, s2, preint, jac=True):
    s1 = np.ascontiguousarray(s1, np.float64)
    s2 = np.ascontiguousarray(s2, np.float64)
    preint = np.ascontiguousarray(preint, np.float64)
    err = np.zeros(9)
    J = np.zeros((9, 24))
    lib.orc_iba_edge_inertial(s1.ctypes.data, s2.ctypes.data, preint.ctypes.data, err.ctypes.data, J.ctypes.data if jac else None)
    return err, J


from synth_iba import Window, make_window, KF, PREINT      # noqa: E402,F401  (re-exported for the tests)


def edge_visual(win, s, X, obs, stereo):
    s = np.ascontiguousarray(s, np.float64)
    X = np.ascontiguousarray(X, np.float64)
    obs = np.ascontiguousarray(obs, np.float64)
    err, Jx, Jp = np.zeros(3), np.zeros((3, 3)), np.zeros((3, 6))
    p = win.struct(Problem)
    lib.orc_iba_edge_visual(C.addressof(p), s.ctypes.data, X.ctypes.data, obs.ctypes.data, int(stereo), err.ctypes.data,
                            Jx.ctypes.data, Jp.ctypes.data)
    return err, Jx, Jp


def solve(win, params=None):
    """-> (kf_state, points, edge_outlier, Stats)"""
    params = params or default_params()
    p = win.struct(Problem)
    kf, pts = win.kf0.copy(), win.pts0.copy()
    out = np.zeros(max(win.n_edges, 1), np.uint8)
    st = Stats()
    rc = lib.orc_iba_solve(C.addressof(p), C.addressof(params), kf.ctypes.data, pts.ctypes.data, out.ctypes.data, C.addressof(st))
    assert rc == 0, rc
    return kf, pts, out[:win.n_edges], st
