# FrontEnd::Update loop (front_end.cpp:88-424) device-resident: per-scan and per-key-frame cost on a 60-scan drive
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, torch
from lidar_slam_amd import capi, front_end, synth
scene = synth.Scene(seed=1)
N = 60
poses = synth.trajectory(N, step=1.0)
t0 = time.time()
scans = [synth.gen_scan(scene, poses[i], seed=500 + i) for i in range(N)]
print("gen %.1f s, %d pts/scan" % (time.time() - t0, scans[0].shape[0]), flush=True)
ctx = capi.Context(0)
d_scans = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in scans]
torch.cuda.synchronize()
for rep in range(2):
    fe = front_end.FrontEndHip(ctx)
    fe.SetInitPose(poses[0].astype(np.float32))
    times, kf = [], []
    for i in range(N):
        t0 = time.perf_counter()
        ok, T = fe.Update(d_scans[i])
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        kf.append(fe.last_stats["key_frame"])
    err = np.linalg.norm(T[:3, 3] - poses[N - 1][:3, 3])
    times, kf = np.array(times), np.array(kf, dtype=bool)
    print("rep %d: %d key-frames; per-scan median %.3f ms (no key-frame) / %.3f ms (key-frame), total %.1f ms for %d scans = %.0f scans/s; end drift %.3f m; map %d pts -> target %d" % (
        rep, kf.sum(), np.median(times[~kf][2:]), np.median(times[kf][1:]), times[1:].sum(), N - 1, (N - 1) / times[1:].sum() * 1e3, err,
        fe.last_stats["local_map_points"], fe.last_stats["target_points"]), flush=True)
    fe.close()
