"""ScanMatch in PCL-semantics mode (the "NDT" plugin, ndt_registration.cpp:46-66) on the nine bench scans: ms per call, iterations, passes."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, torch  # noqa: F401
from lidar_slam_amd import capi, cloud_filter, registration
from lidar_slam_amd import workloads as wl
poses, scans = wl.cfg2_dataset(workers=8)
ctx = capi.Context(0)
lm = torch.from_numpy(wl.local_map(poses, scans)).cuda()
mf = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
ff = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
tgt = mf.filter_device(lm).clone()
srcs = [ff.filter_device(torch.from_numpy(np.ascontiguousarray(scans[k])).cuda()).clone() for k in wl.TEST_SCANS]
guesses = [wl.guess(poses, k) for k in wl.TEST_SCANS]
for mode in (False, True):
    ndt = registration.NDTRegistrationHip(ctx, pcl_mode=mode, **wl.NDT_PARAMS)
    ndt.SetInputTarget(tgt)
    for rep in range(3):
        t0 = time.perf_counter(); its = []; ps = []
        for _ in range(20):
            for s, g in zip(srcs, guesses):
                ok, T = ndt.ScanMatch(s, g)
                st = ndt.last_stats
                its.append(st["iterations"]); ps.append(st["passes"])
        dt = (time.perf_counter() - t0) / (20 * len(srcs))
    print("%s: %.4f ms per ScanMatch, iterations %.1f, passes %.1f -> %.1f us per pass" % ("PCL-semantics" if mode else "NDTMANUAL", dt * 1e3, np.mean(its), np.mean(ps), dt * 1e6 / np.mean(ps)))
