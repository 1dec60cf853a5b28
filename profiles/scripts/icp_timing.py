import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")); import conftest
import numpy as np, torch
from lidar_slam_amd import capi, cloud_filter, registration, synth, workloads as wl
poses, scans = wl.cfg2_dataset(workers=4)
ctx = capi.Context(0)
mf = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3); ff = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
d_map_f = mf.filter_device(torch.from_numpy(wl.local_map(poses, scans)).cuda()).clone()
srcs = [ff.filter_device(torch.from_numpy(np.ascontiguousarray(scans[k])).cuda()).clone() for k in wl.TEST_SCANS]
guesses = [wl.guess(poses, k) for k in wl.TEST_SCANS]
for mode, name in ((capi.ICP_MODE_MANUAL, "manual"), (capi.ICP_MODE_PLANE, "plane")):
    icp = registration.ICPRegistrationHip(ctx, 2.0, 30, mode=mode)
    icp.SetInputTarget(d_map_f)
    ts = []
    Ts = []
    for rep in range(3):
        for s, g in zip(srcs, guesses):
            torch.cuda.synchronize(); t0 = time.perf_counter(); _, T = icp.ScanMatch(s, g); ts.append(time.perf_counter() - t0)
            if rep == 0: Ts.append(T)
    print(name, "coarse grid for far queries" if not os.environ.get("LSA_ICP_NO_COARSE") else "fine grid only",
          "every search run" if os.environ.get("LSA_ICP_NO_SKIP") else "proven searches skipped", "median ms per 30-iteration ScanMatch %.4f" % (np.median(ts) * 1e3), "checksum %.9f" % float(sum(T.astype(np.float64).sum() for T in Ts)))
    print("   searches per iteration of the last ScanMatch:", icp.search_counts(), " armed passes / fallbacks:", icp.armed_counts())
    icp.close()
