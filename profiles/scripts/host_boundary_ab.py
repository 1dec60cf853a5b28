import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, torch
from lidar_slam_amd import capi, cloud_filter, registration, synth
from lidar_slam_amd import workloads as wl
ctx = capi.Context(0)
poses, scans = wl.cfg2_dataset(workers=8)
mf = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
d_map_f = mf.filter_device(torch.from_numpy(wl.local_map(poses, scans)).cuda()).clone()
ndt = registration.NDTRegistrationHip(ctx, **wl.NDT_PARAMS)
ndt.SetInputTarget(d_map_f)
ks = wl.TEST_SCANS
raws = [synth.to_xyzi32(scans[k]) for k in ks]
gs = [wl.guess(poses, k) for k in ks]
arms = {}
for name, env in (("copy", "0"), ("pull", "1")):
    os.environ["LSA_FILTER_PULL"] = env
    arms[name] = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
out8 = np.zeros((max(r.shape[0] for r in raws), 8), dtype=np.float32)
for rnd in range(4):
    for name, f in arms.items():
        tf, ta = [], []
        for it in range(60):
            k = it % len(ks)
            t0 = time.perf_counter()
            _, f8 = f.Filter(raws[k], out8)
            t1 = time.perf_counter()
            ndt.ScanMatch(f8, gs[k])
            t2 = time.perf_counter()
            if it >= 10:
                tf.append(t1 - t0); ta.append(t2 - t1)
        print("round %d %s: filter %.1f us (min %.1f) align %.1f us (min %.1f) total %.1f us  %s" % (rnd, name, np.median(tf) * 1e6, np.min(tf) * 1e6, np.median(ta) * 1e6, np.min(ta) * 1e6,
              (np.median(tf) + np.median(ta)) * 1e6, f.counters()), flush=True)
for f in arms.values():
    f.close()
