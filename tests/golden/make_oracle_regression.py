"""Regenerates tests/golden/oracle_regression.json: outputs of the CPU oracle itself on the cfg1 pair.
NOT a pin against the reference (the reference cannot be run here) — a tripwire: any later edit of
oracle/*.c that changes what the oracle computes shows up as a failing CPU test instead of silently moving the goal posts
of the GPU parity tests.   python tests/golden/make_oracle_regression.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import conftest  # noqa: E402,F401
import oracle_py  # noqa: E402
from lidar_slam_amd import synth  # noqa: E402


def compute():
    oracle_py.build(ref=False)
    o = oracle_py.Oracle()
    scene = synth.Scene(seed=1)
    s0 = synth.gen_scan(scene, synth.pose_matrix(), seed=10)
    s1 = synth.gen_scan(scene, synth.pose_matrix(1.0, 0.05, 0.01, 0.002, -0.003, 0.02), seed=11)
    tgt = synth.to_xyzi32(s0)
    filt, _ = o.voxel_filter(synth.to_xyzi32(s1), 0.5)
    src = synth.to_xyzi32(filt)
    out = {"n_scan0": int(s0.shape[0]), "n_scan1": int(s1.shape[0]), "n_filtered": int(filt.shape[0]),
           "filtered_checksum": float(np.sum(filt.astype(np.float64) * np.arange(1, 5)))}
    guess = np.eye(4, dtype=np.float32)
    for name, kw in (("ndt_manual", {}), ("ndt_pcl", {"pcl_mode": True})):
        n = oracle_py.OracleNDT(**kw)
        n.set_input_target(tgt)
        occ, valid = n.grid_counts()
        T, st = n.scan_match(src, guess)
        out[name] = {"occupied": occ, "valid": valid, "iterations": st["iterations"], "passes": st["passes"], "pose": [float(v) for v in T.reshape(-1)],
                     "fitness": float(n.fitness())}
    out["icp_manual"] = [float(v) for v in oracle_py.icp_manual(tgt, src, guess).reshape(-1)]
    Tp, stp = oracle_py.icp_pcl(tgt, src, guess)
    out["icp_pcl"] = {"pose": [float(v) for v in Tp.reshape(-1)], "iterations": stp["iterations"], "correspondences": stp["correspondences"]}
    Tl, nc = oracle_py.icp_plane(tgt, src, guess)
    out["icp_plane"] = {"pose": [float(v) for v in Tl.reshape(-1)], "correspondences": nc}
    y = oracle_py.OracleYawInit(tgt, np.zeros(3, dtype=np.float32), 0.8)
    a, probs = y.yaw(src)
    out["yaw"] = {"wh": list(y.info()[:2]), "angle": float(a), "probs_sum": float(np.nansum(probs))}
    out["box"] = int(oracle_py.box_filter(tgt, [-30, 30, -10, 12, -2, 1], (2.5, -1.0, 0.25)).shape[0])
    return out


if __name__ == "__main__":
    json.dump(compute(), open(os.path.join(HERE, "oracle_regression.json"), "w"), indent=1)
    print("written")
