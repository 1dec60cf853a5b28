"""Where a persistent ScanMatch spends its time (workgroup 0's clock stamps per pass), cfg2 scans; and A/B timing of the
persistent launch against the per-pass path through the C-ABI (no Python in the timed loop: lsa_pipeline_scan_match)."""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa
import torch
from lidar_slam_amd import capi, cloud_filter, registration, workloads as wl

def main():
    dense = len(sys.argv) > 1 and sys.argv[1] == "dense"
    poses, scans = (wl.dense_dataset if dense else wl.cfg2_dataset)(workers=1)
    s = torch.cuda.Stream(); torch.cuda.set_stream(s)
    ctx = capi.Context(0, s.cuda_stream)
    mf = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3); ff = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    d_map_f = mf.filter_device(torch.from_numpy(wl.local_map(poses, scans)).cuda()).clone()
    d_src = [ff.filter_device(torch.from_numpy(np.ascontiguousarray(scans[k])).cuda()).clone() for k in wl.TEST_SCANS]
    guesses = [wl.guess(poses, k) for k in wl.TEST_SCANS]
    n = registration.NDTRegistrationHip(ctx, **wl.NDT_PARAMS)
    n.SetInputTarget(d_map_f)
    L = capi.lib()
    out = {}
    for mode in (True, False):
        n.set_persistent(mode)
        for _ in range(300):
            for d_s, g in zip(d_src, guesses): n.ScanMatch(d_s, g)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 300
        for _ in range(reps):
            for d_s, g in zip(d_src, guesses): n.ScanMatch(d_s, g)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (reps * len(d_src))
        out["persistent" if mode else "per_pass"] = dt * 1e6
        print("persistent" if mode else "per-pass  ", "%.1f us per ScanMatch (python loop)" % (dt * 1e6), n.align_counters(), flush=True)
    n.set_persistent(True)
    buf = (C.c_uint64 * 240)(); nn = C.c_size_t(0)
    ctx.check(L.lsa_ndt_align_stamps(n.h, 1, None, 0, None))
    rows = []
    for d_s, g in zip(d_src, guesses):
        for _ in range(3): n.ScanMatch(d_s, g)
        ctx.check(L.lsa_ndt_align_stamps(n.h, 1, buf, 240, C.byref(nn)))
        st = np.array(buf[:], dtype=np.int64).reshape(40, 6)
        st = st[st[:, 0] > 0]
        t = (st - st[0, 0]) / 100.0   # us
        rows.append(t)
        print("M", d_s.shape[0], "passes", len(t))
        for r in t:
            print("   start %.2f  body %.2f  publish %.2f  flags %.2f  reduce %.2f  solve %.2f" % (r[0], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4]))
    ph = np.concatenate([np.stack([r[:, 1] - r[:, 0], r[:, 2] - r[:, 1], r[:, 3] - r[:, 2], r[:, 4] - r[:, 3], r[:, 5] - r[:, 4]], 1) for r in rows])
    out["phase_us_mean"] = dict(zip(["body", "publish", "flags", "reduce", "solve"], ph.mean(0).round(2).tolist()))
    out["phase_us_first_pass"] = dict(zip(["body", "publish", "flags", "reduce", "solve"], np.stack([np.array([r[0, 1] - r[0, 0], r[0, 2] - r[0, 1], r[0, 3] - r[0, 2], r[0, 4] - r[0, 3], r[0, 5] - r[0, 4]]) for r in rows]).mean(0).round(2).tolist()))
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "align_stamps%s.json" % ("_dense" if dense else "")), "w"))

if __name__ == "__main__":
    main()
