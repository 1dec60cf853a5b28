"""BASELINE.json's configs restated on the synthetic generator, in ONE place so that bench.py, the
profiling scripts and the -m gpu parity tests run exactly the same clouds, local map, guesses and parameters.

  cfg2  KITTI HDL-64E scan-to-local-map: 30-pose drive, local map = key-frames 0, 2, .., 20 (11 >= 10 -> filtered at
        0.3 m, front_end.cpp:348-424), test scans 21..29 with a constant-velocity guess + a seeded prediction error.
  cfg4  a raw ~132 k-point scan against the 2 M-point unfiltered submap (17 scans), no frame filter.
  cfg5  the 128-beam 2,097,152-point scan against the same submap.
  dense a second, KITTI-shaped bracket of cfg2: the same street with vegetation-like volumetric clutter and more
        facades.

Nothing here touches the GPU or forks a process that holds a GPU context: the generators are numpy; the optional workers
are plain child programs.
"""
import os
import time

import numpy as np

if __name__ == "__main__":   # a dataset worker (see _dataset): import the sibling module without the package
    import importlib.util as _ilu
    import sys as _sys
    _spec = _ilu.spec_from_file_location("lsa_synth_worker", os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth.py"))
    synth = _ilu.module_from_spec(_spec)
    _spec.loader.exec_module(synth)
else:
    from . import synth

NDT_PARAMS = dict(res=1.0, step_size=0.1, trans_eps=0.01, max_iter=30, outlier_ratio=0.55)   # config/mapping/front_end.yaml:24-29
N_POSES = 30
MAP_KEYFRAMES = list(range(0, 21, 2))      # key-frame every 2 m (key_frame_distance 2.0): 11 frames >= 10 -> filtered at 0.3
TEST_SCANS = list(range(21, 30))           # scans matched against that local map
CFG2_CACHE = "/tmp/lsa_bench_cfg2_v3.npz"
DENSE_CACHE = "/tmp/lsa_bench_dense_v2.npz"
DENSE_MAX_RANGE = 100.0
# "kitti": the KITTI-residential-SHAPED workload: TerrainScene (rolling ground, vegetation belts, houses on
# the terrain), the HDL-64E's real two-block beam table, 120 m range, a 50-pose drive, local map = the reference's full window
# of local_frame_num = 20 key-frames (front_end.yaml:12) every 2 m, test scans 41..49
KITTI_CACHE = "/tmp/lsa_bench_kitti_v4.npz"
KITTI_POSES = 50
KITTI_MAP_KEYFRAMES = list(range(0, 40, 2))
KITTI_TEST_SCANS = list(range(41, 50))
KITTI_MAX_RANGE = 120.0


def _gen_one(kind, scene_seed, pose, seed):
    if kind == "cfg2":
        return synth.gen_scan(synth.Scene(seed=scene_seed), pose, seed=seed)
    if kind == "kitti":
        return synth.gen_scan(synth.TerrainScene(seed=scene_seed), pose, seed=seed, max_range=KITTI_MAX_RANGE, layout="hdl64e")
    return synth.gen_scan(synth.DenseScene(seed=scene_seed), pose, seed=seed, max_range=DENSE_MAX_RANGE)


def _dataset(kind, path, workers, leader=True, N_POSES=N_POSES):
    poses = synth.trajectory(N_POSES)
    if leader and not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid()
        if workers > 1:
            # worker PROCESSES started as plain child programs (this file run as a script over a slice of the poses): no fork
            # of a process that may hold a GPU context, no pickling of functions from this hyphenated package
            import subprocess
            import sys
            bounds = [(N_POSES * w) // workers for w in range(workers + 1)]
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), kind, str(bounds[w]), str(bounds[w + 1]), tmp, str(N_POSES)])
                     for w in range(workers) if bounds[w + 1] > bounds[w]]
            for pr in procs:
                if pr.wait() != 0:
                    raise RuntimeError("dataset worker failed (%s)" % kind)
            scans = [np.load("%s.%d.npy" % (tmp, i)) for i in range(N_POSES)]
            for i in range(N_POSES):
                os.remove("%s.%d.npy" % (tmp, i))
        else:
            scans = [_gen_one(kind, 1, poses[i], 100 + i) for i in range(N_POSES)]
        np.savez(tmp + ".npz", **{"s%d" % i: s for i, s in enumerate(scans)})
        os.replace(tmp + ".npz", path)
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


def kitti_dataset(workers=1, leader=True):
    return _dataset("kitti", KITTI_CACHE, workers, leader, N_POSES=KITTI_POSES)


def local_map(poses, scans, keyframes=None):
    """front_end.cpp:348-424: key-frames transformed into the map frame (pose 0) and concatenated; the caller filters at 0.3."""
    P0inv = np.linalg.inv(poses[0])
    parts = [synth.transform_points(P0inv @ poses[k], scans[k]) for k in (MAP_KEYFRAMES if keyframes is None else keyframes)]
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


def wide_map(seed=3, half=160.0, n_source=2097152):
    """A target that does NOT fit a 4 MiB per-XCD L2: an undulating surface over +-`half` m with 12
    points per 1 m column, a second (canopy / roof) layer over a third of the area: >= 100 k valid 1 m voxels, a dense probe
    table of ~320 x 320 x 16 cells (26 MB) and ~10 MB of records.  The source is `n_source` samples of the same surfaces
    spread over the whole extent — in random order (every wave's 14 points land in 14 unrelated places: the adversarial
    case for the table's cache residency) — and the guess a small offset."""
    rng = np.random.RandomState(seed)

    def height(x, y):
        return 3.0 * np.sin(x / 37.0) * np.cos(y / 29.0) + 1.5 * np.sin((x + y) / 11.0)

    def surface(n, zoff, noise, mask_frac=None):
        x = rng.uniform(-half, half, n)
        y = rng.uniform(-half, half, n)
        if mask_frac is not None:   # keep the points whose 16 m tile is "covered" (a seeded third of the tiles)
            tile = (np.floor(x / 16.0).astype(np.int64) * 73856093) ^ (np.floor(y / 16.0).astype(np.int64) * 19349663)
            keep = (tile % 100) < int(mask_frac * 100)
            x, y = x[keep], y[keep]
        z = height(x, y) + zoff + rng.normal(0.0, noise, x.shape[0])
        return np.stack([x, y, z, np.ones_like(x)], axis=1).astype(np.float32)

    cols = int((2 * half) ** 2)
    target = np.concatenate([surface(12 * cols, 0.0, 0.05), surface(24 * cols, 4.5, 0.35, mask_frac=0.33)], axis=0)
    src = np.concatenate([surface(int(n_source * 0.8), 0.0, 0.05), surface(int(n_source * 0.2 / 0.33) + 4096, 4.5, 0.35, mask_frac=0.33)], axis=0)
    src = np.ascontiguousarray(src[rng.permutation(src.shape[0])[:n_source]])
    guess = synth.pose_matrix(0.06, -0.04, 0.02, 0.0005, -0.0004, 0.0008).astype(np.float32)
    return np.ascontiguousarray(target), src, guess


def pose_err(Ta, Tb):
    """(translation error in m, geodesic rotation angle in rad) between two 4x4 poses."""
    dt = float(np.linalg.norm(Ta[:3, 3].astype(np.float64) - Tb[:3, 3].astype(np.float64)))
    R = Ta[:3, :3].astype(np.float64).T @ Tb[:3, :3].astype(np.float64)
    v = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return dt, float(np.arctan2(np.linalg.norm(v), (np.trace(R) - 1.0) / 2.0))


if __name__ == "__main__":
    _kind, _i0, _i1, _prefix = _sys.argv[1], int(_sys.argv[2]), int(_sys.argv[3]), _sys.argv[4]
    _poses = synth.trajectory(int(_sys.argv[5]) if len(_sys.argv) > 5 else N_POSES)
    for _i in range(_i0, _i1):
        np.save("%s.%d.npy" % (_prefix, _i), _gen_one(_kind, 1, _poses[_i], 100 + _i))
