import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, torch
from lidar_slam_amd import capi, front_end, synth
from lidar_slam_amd import workloads as wl
poses, scans = wl.cfg2_dataset(workers=8)
ctx = capi.Context(0)
h_all = [synth.to_xyzi32(s) for s in scans]                                   # PointXYZI rows (32 B), as a ROS / PCL caller holds them
d_all = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in scans]
torch.cuda.synchronize()
for mode, clouds in (("host clouds (lsa_front_end_update)", h_all), ("device clouds (lsa_front_end_update_device)", d_all)):
    for rep in range(4):
        fe = front_end.FrontEndHip(ctx); fe.SetInitPose(poses[0].astype(np.float32))
        ts = []
        t00 = time.perf_counter()
        for c in clouds:
            t0 = time.perf_counter(); _, T = fe.Update(c); ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        tot = time.perf_counter() - t00
        ts = np.array(ts) * 1e6
        print("%s rep %d: %d scans %.2f ms = %.0f scans/s; steady-state call (scans 20..29) median %.1f us; pose %s" % (
            mode, rep, len(clouds), tot * 1e3, len(clouds) / tot, np.median(ts[20:]), np.array2string(T[:3, 3], precision=4)), flush=True)
        fe.close()
for rep in range(4):
    fe = front_end.FrontEndHip(ctx); fe.SetInitPose(poses[0].astype(np.float32))
    torch.cuda.synchronize()
    t00 = time.perf_counter()
    Ts, kf = fe.UpdateBatch(h_all, stats="key_frame")
    torch.cuda.synchronize()
    tot = time.perf_counter() - t00
    print("host clouds as ONE run (lsa_front_end_update_batch) rep %d: %d scans %.2f ms = %.0f scans/s; pose %s  filter %s" % (
        rep, len(h_all), tot * 1e3, len(h_all) / tot, np.array2string(Ts[-1][:3, 3], precision=4), fe.frame_filter_counters()), flush=True)
    fe.close()
