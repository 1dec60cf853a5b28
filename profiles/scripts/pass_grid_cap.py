import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, torch
from lidar_slam_amd import capi, cloud_filter, registration
from lidar_slam_amd import workloads as wl
ctx = capi.Context(0)
poses, scans = wl.cfg2_dataset(workers=4)
mf = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3); ff = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
d_map_f = mf.filter_device(torch.from_numpy(wl.local_map(poses, scans)).cuda()).clone()
k = wl.TEST_SCANS[0]
src = ff.filter_device(torch.from_numpy(np.ascontiguousarray(scans[k])).cuda()).clone()
ndt = registration.NDTRegistrationHip(ctx, **wl.NDT_PARAMS)
ndt.SetInputTarget(d_map_f); ndt.set_source(src)
g = wl.guess(poses, k); p = np.zeros(6); capi.lib().lsa_matrix_to_pose(capi.fptr(capi.colmajor(g)), capi.dptr(p))
for rep in range(3):
    (score, gr, H, pairs, npts), ms = ndt.derivatives_timed(g, p, kind=0, repeat=200)
print("grid cap", os.environ.get("LSA_PASS_GRID_CAP"), "points", int(npts), "us per launch (200 back to back)", round(ms * 1e3, 2))
