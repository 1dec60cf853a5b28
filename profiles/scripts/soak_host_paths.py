# Soak of the HOST-cloud paths added in round 4 (chunk flags, pulling mark kernel, helper threads, rotating buffers): each path is
# driven for SOAK_SECONDS (default 60) over the cfg2 scans and EVERY result is compared bit for bit with the first pass through the
# same scans on the copy-only path (LSA_FILTER_PULL=0) or on device clouds.  Writes one JSON line.
import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, torch
from lidar_slam_amd import capi, cloud_filter, front_end, registration, synth
from lidar_slam_amd import workloads as wl

T_SOAK = float(os.environ.get("SOAK_SECONDS", "60"))
poses, scans = wl.cfg2_dataset(workers=8)
h = [synth.to_xyzi32(s) for s in scans]
d = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in scans]
torch.cuda.synchronize()
N = len(scans)
out = {"scans": N, "points_per_scan": int(scans[0].shape[0]), "seconds_per_path": T_SOAK}


def progress(name, calls, bad, t0):
    print("[soak] %s: %d calls, %d mismatches, %.0f s" % (name, calls, bad, time.perf_counter() - t0), flush=True)


# ---- A. lsa_voxel_filter_apply (host in, host out), the pulled rows against the copied rows
os.environ["LSA_FILTER_PULL"] = "0"
ctx = capi.Context(0)
f0 = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
ref_f = [f0.Filter(c)[1].copy() for c in h]
f0.close()
os.environ.pop("LSA_FILTER_PULL")
f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
calls = bad = 0
t0 = time.perf_counter(); t_say = t0
while time.perf_counter() - t0 < T_SOAK:
    for i in range(N):
        got = f.Filter(h[i])[1]
        bad += not (got.shape == ref_f[i].shape and np.array_equal(got.view(np.uint32), ref_f[i].view(np.uint32)))
        calls += 1
    if time.perf_counter() - t_say > 20:
        t_say = time.perf_counter(); progress("filter", calls, bad, t0)
out["filter_apply_host"] = {"calls": calls, "mismatches": int(bad), "counters": f.counters(), "calls_per_s": round(calls / (time.perf_counter() - t0))}
progress("filter", calls, bad, t0)
f.close()

# ---- B. lsa_pipeline_scan_match_host against Filter(host) + ScanMatch one scan at a time
ca, cb = capi.Context(0), capi.Context(0)
ndt = registration.NDTRegistrationHip(ctx)
ndt.SetInputTarget(d[0])
guesses = [(np.linalg.inv(poses[0]) @ poses[max(i - 1, 0)]).astype(np.float32) for i in range(N)]
plain = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
ref_p = []
for i in range(N):
    _, f8 = plain.Filter(h[i])
    _, T = ndt.ScanMatch(f8, guesses[i])
    ref_p.append((T.copy(), ndt.last_stats["score"], f8.shape[0]))
fa, fb = cloud_filter.VoxelFilterHip(ca, 0.5, 0.5, 0.5), cloud_filter.VoxelFilterHip(cb, 0.5, 0.5, 0.5)
calls = bad = n_scans = 0
t0 = time.perf_counter(); t_say = t0
rng = np.random.default_rng(7)
while time.perf_counter() - t0 < T_SOAK:
    a = int(rng.integers(0, N - 1)); b = int(rng.integers(a + 1, N + 1))      # a batch of 1 .. N scans starting anywhere
    Ts, sts, nfs = registration.pipeline_scan_match_host([fa, fb], ndt, h[a:b], guesses[a:b])
    for j in range(b - a):
        T, sc, nf = ref_p[a + j]
        bad += not (np.array_equal(Ts[j], T) and sts[j]["score"] == sc and nfs[j] == nf)
    calls += 1; n_scans += b - a
    if time.perf_counter() - t_say > 20:
        t_say = time.perf_counter(); progress("pipeline", calls, bad, t0)
out["pipeline_scan_match_host"] = {"calls": calls, "scans": n_scans, "mismatches": int(bad), "counters": [fa.counters(), fb.counters()],
                                   "armed": ndt.armed_stats() if hasattr(ndt, "armed_stats") else None,
                                   "scans_per_s": round(n_scans / (time.perf_counter() - t0))}
progress("pipeline", calls, bad, t0)
fa.close(); fb.close(); plain.close(); ndt.close(); ca.close(); cb.close()

# ---- C / D. the front end from host clouds: one Update per cloud, and the whole drive as one run, against device clouds
fe = front_end.FrontEndHip(ctx); fe.SetInitPose(poses[0].astype(np.float32))
ref_T = np.stack([fe.Update(c)[1] for c in d])
ref_map = fe.map_verify()
fe.close()
for name in ("front_end_update_host", "front_end_update_batch_host"):
    drives = bad = 0
    counters = None
    t0 = time.perf_counter(); t_say = t0
    while time.perf_counter() - t0 < T_SOAK:
        fe = front_end.FrontEndHip(ctx); fe.SetInitPose(poses[0].astype(np.float32))
        if name == "front_end_update_host":
            Ts = np.stack([fe.Update(c)[1] for c in h])
        else:
            cut = 1 + drives % (N - 1)                                         # two runs, cut somewhere else every drive
            Ts = np.stack(list(fe.UpdateBatch(h[:cut])[0]) + list(fe.UpdateBatch(h[cut:])[0]))
        bad += not np.array_equal(Ts, ref_T)
        if drives % 16 == 0:
            bad += fe.map_verify() != ref_map
        counters = fe.frame_filter_counters()
        fe.close()
        drives += 1
        if time.perf_counter() - t_say > 20:
            t_say = time.perf_counter(); progress(name, drives, bad, t0)
    out[name] = {"drives": drives, "scans": drives * N, "mismatches": int(bad), "last_drive_filter_counters": counters,
                 "scans_per_s_incl_create": round(drives * N / (time.perf_counter() - t0))}
    progress(name, drives, bad, t0)
ctx.close()
out["all_identical"] = all(v["mismatches"] == 0 for v in out.values() if isinstance(v, dict))
print(json.dumps(out))
