"""Fixed cost of one lsa_pipeline_scan_match batch: what the driver's `--steps 20 --warmup 5` pays per batch beyond
20 x the steady-state step.  bench.py's own sequence (ramp batch, warm-up batch, ONE timed batch between device-wide
waits; arguments marshalled before the clock starts) for K = 20 ... 10000 in ONE process on one box.
Writes gpurun_out/batch_fixed_cost.json."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401,E402
import torch  # noqa: E402
from lidar_slam_amd import capi, cloud_filter, registration  # noqa: E402
from lidar_slam_amd import workloads as wl  # noqa: E402

L = capi.lib()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.set_stream(s1)
ctx, ctx_f = capi.Context(0, s1.cuda_stream), capi.Context(0, s2.cuda_stream)
poses, scans = wl.cfg2_dataset(workers=4)
mf = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
ff = cloud_filter.VoxelFilterHip(ctx_f, 0.5, 0.5, 0.5)
d_map_f = mf.filter_device(torch.from_numpy(wl.local_map(poses, scans)).cuda()).clone()
d_raw = [torch.from_numpy(np.ascontiguousarray(scans[k])).cuda() for k in wl.TEST_SCANS]
guess_cm = np.stack([capi.colmajor(wl.guess(poses, k)) for k in wl.TEST_SCANS]).astype(np.float32)
cap = max(r.shape[0] for r in d_raw)
bufs = [torch.empty((cap, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
ndt = registration.NDTRegistrationHip(ctx, **wl.NDT_PARAMS)
ndt.SetInputTarget(d_map_f)
torch.cuda.synchronize()


def prep(count):
    idx = [i % 9 for i in range(count)]
    ptrs = (C.c_void_p * count)(*[d_raw[k].data_ptr() for k in idx])
    nn = (C.c_size_t * count)(*[d_raw[k].shape[0] for k in idx])
    pred = np.ascontiguousarray(guess_cm[idx])
    out = np.zeros((count, 16), dtype=np.float32)
    st = (capi.NdtStats * count)()
    nf = (C.c_size_t * count)()

    def call():
        rc = L.lsa_pipeline_scan_match(ff.h, ndt.h, count, ptrs, nn, capi.fptr(pred), bufs[0].data_ptr(), bufs[1].data_ptr(), 0, 1, capi.fptr(out), st, nf)
        assert rc == 0
    call.keep = (ptrs, nn, pred, out, st, nf)
    return call


def timed(K, ramp=12000, warmup=5):
    prep(ramp)()
    prep(warmup)()
    c = prep(K)
    ctx.synchronize(); ctx_f.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    c()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) * 1e6 / K, (t2 - t1) * 1e6


res = {"us_per_step": {}, "final_wait_us": {}}
for K in (20, 10000, 20, 1000, 100, 20, 50, 10000, 20):
    a, b = timed(K)
    res["us_per_step"].setdefault(str(K), []).append(round(a, 2))
    res["final_wait_us"].setdefault(str(K), []).append(round(b, 1))
steady = min(res["us_per_step"]["10000"])
res["fixed_us_per_batch_of_20"] = [round((v - steady) * 20, 1) for v in res["us_per_step"]["20"]]
res["armed"] = ndt.armed_stats()
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "batch_fixed_cost.json"), "w"), indent=1)
