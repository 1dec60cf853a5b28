import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, torch
from lidar_slam_amd import capi, front_end
from lidar_slam_amd import workloads as wl
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
poses, scans = (wl.cfg2_dataset(workers=8) if which == "cfg2" else wl.kitti_dataset(8))
ctx = capi.Context(0)
d_all = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in scans]
torch.cuda.synchronize()
for rep in range(4):
    fe = front_end.FrontEndHip(ctx); fe.SetInitPose(poses[0].astype(np.float32))
    rows = []; prev = fe.map_counters()
    t00 = time.perf_counter()
    for i, d in enumerate(d_all):
        t0 = time.perf_counter()
        fe.Update(d)
        t1 = time.perf_counter()
        if rep >= 2:
            torch.cuda.synchronize()
        t2 = time.perf_counter()
        c = fe.map_counters()
        tag = "".join(k[0:2] + str(int(c[k] - prev[k])) + " " for k in c if c[k] != prev[k])
        prev = c
        rows.append((i, fe.last_key_frame, (t1 - t0) * 1e6, (t2 - t0) * 1e6, tag))
    torch.cuda.synchronize()
    tot = time.perf_counter() - t00
    print("rep %d (%s): %d scans %.2f ms = %.0f scans/s" % (rep, "sync after every Update" if rep >= 2 else "no sync", len(d_all), tot * 1e3, len(d_all) / tot))
    if rep in (1, 3):
        for r in rows:
            print("  scan %2d kf %d call %7.1f us  call+sync %7.1f us  %s" % r)
    fe.close()
