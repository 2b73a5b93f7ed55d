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


if __name__ == "__main__":
    main()
