"""Device-resident frame filter (131 k-point scan, 0.5 m) and SetInputTarget of the 48 k-point local map: ms per call, three rounds."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa
import torch
from lidar_slam_amd import capi, cloud_filter, registration, workloads as wl
poses, scans = wl.cfg2_dataset(workers=1)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = capi.Context(0, s.cuda_stream)
ff = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5); mf = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
d = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in scans[12:21]]
out = torch.empty((200000, 4), dtype=torch.float32, device="cuda")
d_map_f = mf.filter_device(torch.from_numpy(wl.local_map(poses, scans)).cuda()).clone()
ndt = registration.NDTRegistrationHip(ctx, **wl.NDT_PARAMS)
torch.cuda.synchronize()
for rep in range(3):
    for x in d: ff.filter_device(x, out)
    t0 = time.perf_counter()
    for _ in range(60):
        for x in d: ff.filter_device(x, out)
    t1 = time.perf_counter()
    for _ in range(100): ndt.SetInputTarget(d_map_f)
    t2 = time.perf_counter()
    print("frame filter %.4f ms, SetInputTarget(48 k) %.4f ms" % ((t1 - t0) / (60 * len(d)) * 1e3, (t2 - t1) / 100 * 1e3), flush=True)
