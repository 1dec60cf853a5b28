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
for rep in range(4):
    fe = front_end.FrontEndHip(ctx)
    fe.set_incremental(rep >= 2)   # reps 0-1: full rebuild of filter output + voxel table per key-frame; reps 2-3: incremental local map
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
    kfi = np.nonzero(kf)[0]
    late = kfi[kfi >= 2 * 22]   # key-frames of the steady state: window full (20 key-frames), filtered map, evictions
    print("   key-frame ms by phase: unfiltered (<10 kf) %.3f, filtered/filling %.3f, steady state (window slides) %.3f; counters %s" % (
        np.median(times[kfi[1:9]]), np.median(times[kfi[10:20]]) if len(kfi) > 20 else -1, np.median(times[late]) if len(late) else -1, fe.map_counters()), flush=True)
    print("rep %d (%s): %d key-frames; per-scan median %.3f ms (no key-frame) / %.3f ms (key-frame), total %.1f ms for %d scans = %.0f scans/s; end drift %.3f m; map %d pts -> target %d" % (
        rep, "incremental" if rep >= 2 else "full rebuild", kf.sum(), np.median(times[~kf][2:]), np.median(times[kf][1:]), times[1:].sum(), N - 1, (N - 1) / times[1:].sum() * 1e3, err,
        fe.last_stats["local_map_points"], fe.last_stats["target_points"]), flush=True)
    fe.close()
# the same drive as ONE run of lsa_front_end_update_batch_device (next scan's frame filter overlapped, parked passes, deferred
# final pass): twice, the second on warm buffers
for rep in range(3):
    fe = front_end.FrontEndHip(ctx)
    fe.SetInitPose(poses[0].astype(np.float32))
    fe.UpdateBatch(d_scans[:1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Ts, sts = fe.UpdateBatch(d_scans[1:])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    err = np.linalg.norm(Ts[-1][:3, 3] - poses[N - 1][:3, 3])
    print("batch rep %d: %d scans in %.2f ms = %.0f scans/s (%d key-frames); end drift %.3f m; counters %s" % (
        rep, N - 1, dt * 1e3, (N - 1) / dt, sum(s["key_frame"] for s in sts), err, fe.map_counters()), flush=True)
    fe.close()
