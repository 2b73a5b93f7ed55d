#!/usr/bin/env python3
"""Generate tests/golden/*.npz: small committed known-answer vectors.

The reference has no fixtures for this path and cannot run here, so these vectors
are produced by the repo's own CPU oracle from seeded synthetic inputs.  They pin the oracle
against drift and give the GPU tests a fixture that does not need the oracle at all.
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "orb-slam3-mac_amd", "python"))
import oracle_bind as ob      # noqa: E402
import orbhip                 # noqa: E402  (only for the host-side synthetic generator)


def main():
    out = {}
    for name, (w, h, nfeat, lap, seed) in {"a": (320, 240, 300, (0, 1000), 11), "b": (384, 288, 500, (100, 250), 12)}.items():
        img = orbhip.synth_frames(w, h, 1, seed=seed)[0]
        e = ob.OracleExtractor(nfeat, 1.2, 8, 20, 7)
        kp, desc, mono = e.extract(img, lap)
        out["img_" + name] = img
        out["nfeat_" + name] = np.int32(nfeat)
        out["lap_" + name] = np.array(lap, np.int32)
        out["kp_" + name] = kp
        out["desc_" + name] = desc
        out["mono_" + name] = np.int32(mono)
        print(name, len(kp), mono)
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "orb_golden.npz"), **out)
    pose_goldens()
    match_goldens()
    iba_goldens()


def pose_goldens():
    """PoseOptimization known-answer vectors from the CPU oracle on seeded synthetic frames."""
    import oracle_ba_bind as oba
    import synth_ba
    out = {}
    cases = [dict(seed=31, n=200, stereo_frac=0.0), dict(seed=32, n=350, stereo_frac=0.5, outlier_frac=0.2),
             dict(seed=33, n=9, stereo_frac=1.0, outlier_frac=0.0)]
    for k, c in enumerate(cases):
        p = synth_ba.make_pose_problem(**c)
        r, pose, o, st = oba.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["cam"], p["pose0"])
        out.update({f"Xw{k}": p["Xw"], f"obs{k}": p["obs"], f"w{k}": p["inv_sigma2"], f"cam{k}": np.array(p["cam"]),
                    f"pose0_{k}": p["pose0"], f"r{k}": np.int32(r), f"pose{k}": pose, f"out{k}": o})
        print("pose", k, r, st)
    out["count"] = np.int32(len(cases))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pose_golden.npz"), **out)


def match_goldens():
    """Known-answer vectors of the matcher oracles (windowed claim-rule search in both modes, Fuse search, both BoW matchers,
    SearchForTriangulation, the Frame glue and the BowVector assembly) on small seeded cases."""
    import oracle_match_bind as om
    from test_oracle_match_ba import make_sbp_case, _bow_inputs, EUROC_K, EUROC_DIST
    rng = np.random.default_rng(4242)
    bounds = (0.0, 0.0, 640.0, 480.0)
    out = {"bounds": np.array(bounds, np.float32)}
    q, dq, kp, d, ur, tm = make_sbp_case(rng, 180, 150, True)
    out.update(sbp_q=q, sbp_dq=dq, sbp_kp=kp, sbp_d=d, sbp_ur=ur, sbp_tm=tm)
    n0, m0 = om.search_by_projection(q, dq, kp, d, ur, bounds, tm, 100, True)
    n1, m1 = om.search_by_projection_map(q, dq, kp, d, ur, bounds, tm, 100, 0.8)
    out.update(sbp_n=np.int32(n0), sbp_m=m0, map_n=np.int32(n1), map_m=m1)
    q2 = q.copy(); q2["min_level"] = np.maximum(q2["max_level"], 0) - 1; q2["max_level"] = q2["min_level"] + 1
    q2["radius"] = np.float32(3.0) * np.float32(1.2) ** q2["max_level"].astype(np.float32)
    sig = (np.float32(1.0) / (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2).astype(np.float32)
    bi, bd = om.fuse_search(q2, dq, kp, d, ur, sig, bounds)
    out.update(fuse_q=q2, fuse_sig=sig, fuse_bi=bi, fuse_bd=bd)
    c = om.make_bow_case(rng, 160, 170, 25); c["valid2"] = (rng.random(170) < 0.8).astype(np.uint8)
    nb, mb = om.search_by_bow(c, 0.7, True); nk, mk = om.search_by_bow_kf(c, 0.75, True)
    out.update({"bow_" + k: v for k, v in c.items()}); out.update(bow_n=np.int32(nb), bow_m=mb, bowkf_n=np.int32(nk), bowkf_m=mk)
    t = om.make_tri_case(rng, 150, 160, 25, 0.3, False, False)
    nt, mt = om.search_for_triangulation(t, True, False)
    out.update({"tri_" + k: (np.asarray(v) if not isinstance(v, (bool, tuple)) else np.array(v)) for k, v in t.items()})
    out.update(tri_n=np.int32(nt), tri_m=mt)
    un = om.undistort_keypoints(kp, EUROC_K, EUROC_DIST); cs, it = om.assign_features_to_grid(un, bounds)
    out.update(un_kp=un, grid_cs=cs, grid_it=it)
    wid, w, nid = _bow_inputs(rng, 200)
    ni, ns, ft, bw, bv = om.bow_vectors(wid, w, nid)
    out.update(bv_wid=wid, bv_w=w, bv_nid=nid, bv_ni=ni, bv_ns=ns, bv_ft=ft, bv_bw=bw, bv_bv=bv)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "match_golden.npz"), **out)
    print("match goldens:", n0, n1, int((bi >= 0).sum()), nb, nk, nt, len(bw))


IBA_KEYS = ("kf_fixed", "kf_imu", "edge_kf", "edge_point", "edge_obs", "edge_stereo", "edge_inv_sigma2", "edge_close", "in_kf1", "in_kf2",
            "in_preint", "in_info", "in_info_g", "in_info_a", "in_robust", "kf_state", "points", "cam", "Rcb", "tcb")
IBA_OPT = ("camera_model", "kb", "Trl", "cam2", "camera2_model", "kb2")


def iba_goldens():
    """Two small LocalInertialBA windows (pinhole with stereo + mono edges; two-fisheye rig) with the oracle's result."""
    import oracle_iba_bind as ib
    out = {}
    for name, kw in (("w0", dict(seed=501, n_opt=4, n_fixed_vis=3, n_points=80)), ("w1", dict(seed=502, n_opt=3, n_fixed_vis=2, n_points=60, fisheye_rig=True))):
        win = ib.make_window(**kw)
        for k in IBA_KEYS:
            out["%s_in_%s" % (name, k)] = np.asarray(win.d[k])
        for k in IBA_OPT:
            if k in win.d:
                out["%s_in_%s" % (name, k)] = np.asarray(win.d[k])
        kf, pts, outl, st = ib.solve(win)
        out[name + "_kf"] = kf; out[name + "_pts"] = pts; out[name + "_outlier"] = outl
        out[name + "_stats"] = np.array([st.iterations_run, st.lm_trials, st.n_outliers, st.failed], np.int32)
        out[name + "_err"] = np.array([st.err, st.err_end])
        print("iba golden", name, win.n_edges, st.iterations_run, st.lm_trials, st.n_outliers, st.failed)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "iba_golden.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "iba":
        iba_goldens()
    else:
        main()
