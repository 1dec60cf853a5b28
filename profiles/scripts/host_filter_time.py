"""Host-pointer VoxelFilter::Filter of the headline's scans (PointXYZI rows in, rows out), nine scans in rotation: ms per call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa
from lidar_slam_amd import capi, cloud_filter, synth, workloads as wl
poses, scans = wl.cfg2_dataset(workers=1)
h = [synth.to_xyzi32(s) for s in scans[12:21]]
ctx = capi.Context(0)
f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
out = np.zeros((max(x.shape[0] for x in h), 8), dtype=np.float32)
for rep in range(3):
    for x in h: f.Filter(x, out)
    t0 = time.perf_counter()
    for _ in range(40):
        for x in h: f.Filter(x, out)
    print("host Filter: %.4f ms per call" % ((time.perf_counter() - t0) / (40 * len(h)) * 1e3), f.counters(), flush=True)
