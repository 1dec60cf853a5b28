# timings of the widened rows on the cfg1/cfg2-style data (not part of bench.py's metric)
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, torch
from lidar_slam_amd import capi, registration, cloud_filter, matching, synth
scene = synth.Scene(seed=1)
poses = synth.trajectory(22)
ctx = capi.Context(0)
P0i = np.linalg.inv(poses[0])
parts = [synth.transform_points(P0i @ poses[k], synth.gen_scan(scene, poses[k], seed=100 + k)) for k in range(0, 21, 2)]
raw_map = np.ascontiguousarray(np.concatenate(parts, 0))
d_raw = torch.from_numpy(raw_map).cuda()
mf = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3); ff = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
torch.cuda.synchronize()
d_map = mf.filter_device(d_raw).clone()
scan = synth.gen_scan(scene, poses[21], seed=121)
d_scan = torch.from_numpy(scan).cuda(); torch.cuda.synchronize()
d_src = ff.filter_device(d_scan).clone()
guess = (P0i @ poses[20]).astype(np.float32)
print("map", d_map.shape[0], "src", d_src.shape[0], flush=True)
def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / reps
res = {}
for name, mk in (("ndt_manual", lambda: registration.NDTRegistrationHip(ctx)), ("ndt_pcl", lambda: registration.NDTRegistrationHip(ctx, pcl_mode=True)),
                 ("icp_manual", lambda: registration.ICPRegistrationHip(ctx, 2.0, 30)), ("icp_plane", lambda: registration.ICPRegistrationHip(ctx, 2.0, 30, mode=capi.ICP_MODE_PLANE)),
                 ("icp_pcl", lambda: registration.ICPRegistrationHip(ctx, 2.0, 30, mode=capi.ICP_MODE_PCL))):
    r = mk()
    res[name + "_set_target_ms"] = timeit(lambda: r.SetInputTarget(d_map), 10)
    res[name + "_scan_match_ms"] = timeit(lambda: r.ScanMatch(d_src, guess), 20)
    _, T = r.ScanMatch(d_src, guess)
    truth = (P0i @ poses[21]).astype(np.float32)
    res[name + "_err_m"] = float(np.linalg.norm(T[:3, 3] - truth[:3, 3]))
    res[name + "_fitness_ms"] = timeit(lambda: r.GetFitnessScore(), 20)
    res[name + "_fitness"] = float(r.GetFitnessScore())
    r.close()
bf = cloud_filter.BoxFilterHip(ctx, [-50, 50, -50, 50, -5, 10]); bf.SetOrigin([10, 0, 0])
res["box_filter_%dk_ms" % (d_raw.shape[0] // 1000)] = timeit(lambda: bf.filter_device(d_raw), 20)
y = matching.YawInitHip(ctx, 0.8)
origin = (P0i @ poses[20])[:3, 3].astype(np.float32)
res["yaw_build_ms"] = timeit(lambda: y.SetLocalMap(d_map, origin), 10)
sensor_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
res["yaw_search_%dk_pts_ms" % (scan.shape[0] // 1000)] = timeit(lambda: y.GetInitialYawAngle(sensor_scan), 10)
for k, v in res.items(): print("%-32s %.4f" % (k, v))
