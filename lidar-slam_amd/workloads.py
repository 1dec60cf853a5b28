"""BASELINE.json's configs restated on the synthetic generator, in ONE place so that bench.py, the
profiling scripts and the -m gpu parity tests run exactly the same clouds, local map, guesses and parameters.

  cfg2  KITTI HDL-64E scan-to-local-map: 30-pose drive, local map = key-frames 0, 2, .., 20 (11 >= 10 -> filtered at
        0.3 m, front_end.cpp:348-424), test scans 21..29 with a constant-velocity guess + a seeded prediction error.
  cfg4  a raw ~132 k-point scan against the 2 M-point unfiltered submap (17 scans), no frame filter.
  cfg5  the 128-beam 2,097,152-point scan against the same submap.
  dense a second, KITTI-shaped bracket of cfg2: the same street with vegetation-like volumetric clutter and more
        facades.

Nothing here touches the GPU or forks after it has been initialised: the generators are numpy; the optional worker pool
uses the "spawn" start method.
"""
import os
import time

import numpy as np

from . import synth

NDT_PARAMS = dict(res=1.0, step_size=0.1, trans_eps=0.01, max_iter=30, outlier_ratio=0.55)   # config/mapping/front_end.yaml:24-29
N_POSES = 30
MAP_KEYFRAMES = list(range(0, 21, 2))      # key-frame every 2 m (key_frame_distance 2.0): 11 frames >= 10 -> filtered at 0.3
TEST_SCANS = list(range(21, 30))           # scans matched against that local map
CFG2_CACHE = "/tmp/lsa_bench_cfg2_v3.npz"
DENSE_CACHE = "/tmp/lsa_bench_dense_v2.npz"
DENSE_MAX_RANGE = 100.0


def _gen_one(args):
    kind, scene_seed, pose, seed = args
    if kind == "cfg2":
        return synth.gen_scan(synth.Scene(seed=scene_seed), pose, seed=seed)
    return synth.gen_scan(synth.DenseScene(seed=scene_seed), pose, seed=seed, max_range=DENSE_MAX_RANGE)


def _dataset(kind, path, workers, leader=True):
    poses = synth.trajectory(N_POSES)
    if leader and not os.path.exists(path):
        jobs = [(kind, 1, poses[i], 100 + i) for i in range(N_POSES)]
        if workers > 1:
            import multiprocessing as mp
            with mp.get_context("spawn").Pool(workers) as pool:   # spawn: safe even when the GPU is already initialised
                scans = pool.map(_gen_one, jobs)
        else:
            scans = [_gen_one(j) for j in jobs]
        tmp = path + ".tmp%d.npz" % os.getpid()
        np.savez(tmp, **{"s%d" % i: s for i, s in enumerate(scans)})
        os.replace(tmp, path)
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 600:
            raise RuntimeError("timed out waiting for the dataset cache %s" % path)
        time.sleep(0.2)
    z = np.load(path)
    return poses, [z["s%d" % i] for i in range(N_POSES)]


def cfg2_dataset(workers=1, leader=True):
    """30-pose trajectory through the synthetic street; cached in /tmp so the N = 1, 2, 4, 8 runs and the tests share it."""
    return _dataset("cfg2", CFG2_CACHE, workers, leader)


def dense_dataset(workers=1, leader=True):
    return _dataset("dense", DENSE_CACHE, workers, leader)


def local_map(poses, scans):
    """front_end.cpp:348-424: key-frames transformed into the map frame (pose 0) and concatenated; the caller filters at 0.3."""
    P0inv = np.linalg.inv(poses[0])
    parts = [synth.transform_points(P0inv @ poses[k], scans[k]) for k in MAP_KEYFRAMES]
    return np.ascontiguousarray(np.concatenate(parts, axis=0))


def guess(poses, k):
    """Constant-velocity prediction (front_end.cpp:238-240) from the true trajectory plus a seeded prediction error of the
    size a real vehicle's accelerations cause (sigma 8 cm / 0.5 deg yaw): the synthetic drive is almost exactly
    constant-velocity, which would make the prediction mm-accurate and the registration trivial (2 iterations)."""
    P0inv = np.linalg.inv(poses[0])
    Tk1, Tk2 = P0inv @ poses[k - 1], P0inv @ poses[k - 2]
    er = np.random.RandomState(7000 + k)
    delta = synth.pose_matrix(er.normal(0, 0.08), er.normal(0, 0.08), er.normal(0, 0.02), er.normal(0, 0.003), er.normal(0, 0.003), er.normal(0, 0.009))
    return (Tk1 @ (np.linalg.inv(Tk2) @ Tk1) @ delta).astype(np.float32)


def truth(poses, k):
    return (np.linalg.inv(poses[0]) @ poses[k]).astype(np.float32)


def submap_2m(poses, scans):
    """cfg4 / cfg5 target: 17 raw scans in the map frame, cut at 2,000,000 points, UNFILTERED."""
    P0i = np.linalg.inv(poses[0])
    parts = [synth.transform_points(P0i @ poses[i], scans[i]) for i in range(17)]
    return np.ascontiguousarray(np.concatenate(parts, 0))[:2000000]


def cfg4_source(poses):
    """cfg4: one raw HDL-64E scan (~132 k points, no frame filter) taken at pose 8, and its guess (the true pose)."""
    src = synth.gen_scan(synth.Scene(seed=1), poses[8], seed=56)
    return src, (np.linalg.inv(poses[0]) @ poses[8]).astype(np.float32)


def cfg5_source(poses):
    """cfg5: the 128-beam x 16384-column scan, every ray kept (clamped at 120 m): exactly 2,097,152 points."""
    src = synth.gen_scan(synth.Scene(seed=1), poses[8], seed=55, beams=128, az_steps=16384, max_range=120.0, clamp=True)
    return src, (np.linalg.inv(poses[0]) @ poses[8]).astype(np.float32)


def pose_err(Ta, Tb):
    """(translation error in m, geodesic rotation angle in rad) between two 4x4 poses."""
    dt = float(np.linalg.norm(Ta[:3, 3].astype(np.float64) - Tb[:3, 3].astype(np.float64)))
    R = Ta[:3, :3].astype(np.float64).T @ Tb[:3, :3].astype(np.float64)
    v = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return dt, float(np.arctan2(np.linalg.norm(v), (np.trace(R) - 1.0) / 2.0))
