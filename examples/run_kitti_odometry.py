#!/usr/bin/env python3
"""LiDAR odometry over a directory of KITTI velodyne scans with the device-resident front end.

    python examples/run_kitti_odometry.py <velodyne_dir> <poses_out.txt> [--max-scans N] [--gpu 0] [--batch 16] [--host-input]

<velodyne_dir> holds 000000.bin, 000001.bin, ... (float32 x, y, z, reflectance per point: the KITTI raw / odometry layout,
also what the reference's data_pretreat nodes republish).  Every scan goes through lsa_front_end_update* — frame filter,
NDT scan-to-local-map, key-frame / local-map maintenance exactly as FrontEnd::Update does it (front_end.cpp:88-424, YAML
defaults of config/mapping/front_end.yaml) — and one pose per scan is written in the KITTI odometry format (the first three
rows of the 4x4 pose, row-major, relative to the first scan).  No ROS, no PCL; needs an MI355X.
"""
import argparse
import glob
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_pkg():
    if "lidar_slam_amd" in sys.modules:
        return
    pkg_dir = os.path.join(ROOT, "lidar-slam_amd")
    spec = importlib.util.spec_from_file_location("lidar_slam_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["lidar_slam_amd"] = mod
    spec.loader.exec_module(mod)


def run(velodyne_dir, poses_out, max_scans=None, gpu=0, log=print, batch=16, host_input=False):
    import torch
    _load_pkg()
    from lidar_slam_amd import capi, cloud_io, front_end
    files = sorted(glob.glob(os.path.join(velodyne_dir, "*.bin")))
    if max_scans:
        files = files[:max_scans]
    if not files:
        raise SystemExit("no .bin scans in %s" % velodyne_dir)
    torch.cuda.set_device(gpu)
    ctx = capi.Context(gpu)
    fe = front_end.FrontEndHip(ctx)
    fe.SetInitPose(np.eye(4, dtype=np.float32))
    poses, key_frames, t_gpu = [], 0, 0.0
    # A directory of scans is a replay: the scans are at hand before their poses are wanted, so they go through the front end in
    # runs of `batch` (lsa_front_end_update_batch_device: the next scan's frame filter overlaps the registration of the current
    # one; the same poses as one Update per scan, bit for bit).
    for k0 in range(0, len(files), batch):
        clouds = [cloud_io.remove_nan(cloud_io.read_kitti_bin(path))[0] for path in files[k0:k0 + batch]]
        # --host-input: the (n, 4) host arrays go to the library as they are (lsa_front_end_update_batch / lsa_front_end_update: packed,
        # sent over the link and filtered while the previous scan is registered), so the time per scan below includes PCIe
        d_clouds = clouds if host_input else [torch.from_numpy(c).cuda(gpu) for c in clouds]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if batch > 1:
            Ts, stats = fe.UpdateBatch(d_clouds)
        else:
            _, T1 = fe.Update(d_clouds[0])
            Ts, stats = [T1], [fe.last_stats]
        t_gpu += time.perf_counter() - t0
        for j, (T, st) in enumerate(zip(Ts, stats)):
            key_frames += int(st["key_frame"])
            poses.append(np.asarray(T).astype(np.float64))
            if (k0 + j) % 100 == 0:
                log("scan %d / %d: %d points, %d key-frames so far, position %s" % (k0 + j, len(files), clouds[j].shape[0], key_frames, np.round(T[:3, 3], 2)))
    with open(poses_out, "w") as f:
        for T in poses:
            f.write(" ".join("%.9e" % v for v in T[:3, :].reshape(-1)) + "\n")
    log("%d scans, %d key-frames, %.3f ms per scan in the front end (%s)" % (len(files), key_frames, t_gpu / len(files) * 1e3,
                                                                             "host input, PCIe included" if host_input else "device-resident input"))
    fe.close()
    ctx.close()
    return poses


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("velodyne_dir")
    ap.add_argument("poses_out")
    ap.add_argument("--max-scans", type=int, default=None)
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--batch", type=int, default=16, help="scans per front-end call (1: one lsa_front_end_update_device call per scan)")
    ap.add_argument("--host-input", action="store_true", help="hand the scans over as host arrays (pack + PCIe inside the library, overlapped with the registration)")
    a = ap.parse_args()
    run(a.velodyne_dir, a.poses_out, a.max_scans, a.gpu, batch=max(1, a.batch), host_input=a.host_input)
