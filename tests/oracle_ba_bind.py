"""ctypes binding of the BA ORACLE (oracle/ba_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import numpy as np
from oracle_bind import lib

vp, ci, cd = C.c_void_p, C.c_int, C.c_double


class Camera(C.Structure):
    _fields_ = [("fx", cd), ("fy", cd), ("cx", cd), ("cy", cd), ("bf", cd), ("camera_model", C.c_int32), ("kb", cd * 4),
                ("Trl", cd * 7), ("fx2", cd), ("fy2", cd), ("cx2", cd), ("cy2", cd), ("camera2_model", C.c_int32), ("kb2", cd * 4)]


class Graph(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("pose_fixed", vp), ("edge_pose", vp), ("edge_point", vp), ("edge_obs", vp),
                ("edge_inv_sigma2", vp), ("edge_stereo", vp),
                ("fx", cd), ("fy", cd), ("cx", cd), ("cy", cd), ("bf", cd),
                ("camera_model", C.c_int32), ("kb", cd * 4),
                ("Trl", cd * 7), ("fx2", cd), ("fy2", cd), ("cx2", cd), ("cy2", cd), ("camera2_model", C.c_int32), ("kb2", cd * 4),
                ("n_cameras", C.c_int32), ("cameras", vp), ("pose_camera", vp)]


class Params(C.Structure):
    _fields_ = [("iters1", C.c_int32), ("iters2", C.c_int32), ("huber_mono2", cd), ("huber_stereo2", cd),
                ("user_lambda_init", cd), ("tau", cd), ("max_trials", C.c_int32),
                ("stage2_exclude_outliers", C.c_int32), ("stage2_drop_robust", C.c_int32), ("no_discard", C.c_int32),
                ("gate_mono2", cd), ("gate_stereo2", cd)]


class Stats(C.Structure):
    _fields_ = [("iterations_run", C.c_int32 * 2), ("lm_trials", C.c_int32), ("n_outliers", C.c_int32),
                ("discarded", C.c_int32), ("chi2_initial", cd), ("chi2_final", cd)]

    def as_dict(self):
        return dict(iterations_run=list(self.iterations_run), lm_trials=self.lm_trials, n_outliers=self.n_outliers,
                    discarded=self.discarded, chi2_initial=self.chi2_initial, chi2_final=self.chi2_final)


lib.orc_ba_default_params.argtypes = [C.POINTER(Params)]
lib.orc_ba_merge_params.argtypes = [C.POINTER(Params)]
lib.orc_ba_solve.argtypes = [C.POINTER(Graph), C.POINTER(Params), vp, vp, vp, vp, C.POINTER(Stats)]
lib.orc_ba_solve_ex.argtypes = [C.POINTER(Graph), C.POINTER(Params), vp, vp, vp, vp, C.POINTER(Stats), vp, vp]
lib.orc_se3_exp.argtypes = [vp, vp, vp]
lib.orc_se3_oplus.argtypes = [vp, vp]
lib.orc_ba_edge.argtypes = [vp, vp, vp, ci, cd, cd, cd, cd, cd, vp, vp, vp]
lib.orc_ba_edge_kb8.argtypes = [vp, vp, vp, cd, cd, cd, cd, vp, vp, vp, vp]
lib.orc_ba_edge_tobody.argtypes = [C.POINTER(Graph), vp, vp, vp, vp, vp, vp]


def default_params():
    p = Params()
    lib.orc_ba_default_params(C.byref(p))
    return p


def rig2_fields(g):
    r = g.get("rig2")
    if r is None:
        return ((cd * 7)(0, 0, 0, 0, 0, 0, 0), 0.0, 0.0, 0.0, 0.0, 0, (cd * 4)(0, 0, 0, 0))
    kb2 = r.get("kb")
    return ((cd * 7)(*r["Trl"]), *[float(c) for c in r["cam"]], 1 if kb2 is not None else 0, (cd * 4)(*(kb2 if kb2 is not None else (0, 0, 0, 0))))


def merge_params():
    p = Params()
    lib.orc_ba_merge_params(C.byref(p))
    return p


def global_params(iterations, robust=True):
    p = Params()
    lib.orc_ba_global_params(C.byref(p), iterations, 1 if robust else 0)
    return p


def make_cgraph(g, cls=Graph):
    """g: dict from synth_ba.make_graph.  Returns (struct, keepalive list)."""
    keep = [np.ascontiguousarray(g["pose_fixed"], np.uint8), np.ascontiguousarray(g["edge_pose"], np.int32),
            np.ascontiguousarray(g["edge_point"], np.int32), np.ascontiguousarray(g["edge_obs"], np.float64),
            np.ascontiguousarray(g["edge_inv_sigma2"], np.float64), np.ascontiguousarray(g["edge_stereo"], np.uint8)]
    kb = g.get("kb")
    cams = g.get("cameras")
    ncam, cam_arr, pcam = 0, None, None
    if cams:
        ncam = len(cams)
        cam_arr = (Camera * ncam)()
        for i, c in enumerate(cams):
            k2 = c.get("kb")
            cam_arr[i] = Camera(c["fx"], c["fy"], c["cx"], c["cy"], c["bf"], 1 if k2 is not None else 0, (cd * 4)(*(k2 if k2 is not None else (0, 0, 0, 0))),
                                *rig2_fields(dict(rig2=c.get("rig2"))))
        pcam = np.ascontiguousarray(g["pose_camera"], np.int32)
        keep += [cam_arr, pcam]
    s = cls(g["n_poses"], g["n_points"], g["n_edges"], *[k.ctypes.data for k in keep[:6]],
            g["fx"], g["fy"], g["cx"], g["cy"], g["bf"], 1 if kb is not None else 0, (cd * 4)(*(kb if kb is not None else (0, 0, 0, 0))),
            *rig2_fields(g), ncam, C.cast(cam_arr, vp) if ncam else None, pcam.ctypes.data if ncam else None)
    return s, keep


def solve(g, params=None, abort=None):
    p = params or default_params()
    cg, keep = make_cgraph(g)
    poses = np.ascontiguousarray(g["poses0"], np.float64).copy()
    pts = np.ascontiguousarray(g["points0"], np.float64).copy()
    out = np.zeros(max(g["n_edges"], 1), np.uint8)
    st = Stats()
    ab = abort.ctypes.data if abort is not None else None
    rc = lib.orc_ba_solve(C.byref(cg), C.byref(p), ab, poses.ctypes.data, pts.ctypes.data, out.ctypes.data, C.byref(st))
    return rc, poses, pts, out[:g["n_edges"]], st.as_dict()


def solve_with_gate_values(g, params=None):
    """solve() plus the stored chi2 and the depth of every edge at the end (what the outlier gates of Optimizer.cc:2126-2173 test)."""
    p = params or default_params()
    cg, keep = make_cgraph(g)
    poses = np.ascontiguousarray(g["poses0"], np.float64).copy()
    pts = np.ascontiguousarray(g["points0"], np.float64).copy()
    n = max(g["n_edges"], 1)
    out = np.zeros(n, np.uint8); chi2 = np.zeros(n); depth = np.zeros(n)
    st = Stats()
    rc = lib.orc_ba_solve_ex(C.byref(cg), C.byref(p), None, poses.ctypes.data, pts.ctypes.data, out.ctypes.data, C.byref(st),
                             chi2.ctypes.data, depth.ctypes.data)
    E = g["n_edges"]
    return rc, poses, pts, out[:E], st.as_dict(), chi2[:E], depth[:E]


# ------------------------------------------------------------------ PoseOptimization oracle
class PoseProblem(C.Structure):
    _fields_ = [("n_edges", C.c_int32), ("Xw", vp), ("obs", vp), ("inv_sigma2", vp),
                ("fx", cd), ("fy", cd), ("cx", cd), ("cy", cd), ("bf", cd), ("camera_model", C.c_int32), ("kb", cd * 4),
                ("right", vp), ("Trl", cd * 7), ("fx2", cd), ("fy2", cd), ("cx2", cd), ("cy2", cd), ("camera2_model", C.c_int32), ("kb2", cd * 4)]


class PoseStats(C.Structure):
    _fields_ = [("rounds", C.c_int32), ("iterations", C.c_int32 * 4), ("lm_trials", C.c_int32), ("n_bad", C.c_int32)]


lib.orc_pose_optimization.argtypes = [C.POINTER(PoseProblem), vp, vp, C.POINTER(PoseStats)]
lib.orc_pose_optimization.restype = ci


def pose_optimization(Xw, obs, inv_sigma2, cam, pose0, kb8=None, rig2=None, right=None):
    """Optimizer::PoseOptimization restated.  Returns (n_inliers, pose7, outlier[n], stats dict)."""
    Xw = np.ascontiguousarray(Xw, np.float64).reshape(-1, 3)
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 3)
    w = np.ascontiguousarray(inv_sigma2, np.float64)
    n = len(Xw)
    rt = None if right is None else np.ascontiguousarray(right, np.uint8)
    P = PoseProblem(n, Xw.ctypes.data, obs.ctypes.data, w.ctypes.data, *[float(c) for c in cam], 1 if kb8 is not None else 0,
                    (cd * 4)(*(kb8 if kb8 is not None else (0, 0, 0, 0))), None if rt is None else rt.ctypes.data,
                    *rig2_fields(dict(rig2=rig2)))
    pose = np.ascontiguousarray(pose0, np.float64).copy()
    out = np.zeros(max(n, 1), np.uint8)
    st = PoseStats()
    r = lib.orc_pose_optimization(C.byref(P), pose.ctypes.data, out.ctypes.data, C.byref(st))
    return r, pose, out[:n], dict(rounds=st.rounds, iterations=sum(st.iterations), lm_trials=st.lm_trials, n_bad=st.n_bad)
