import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, torch
from lidar_slam_amd import capi, registration, cloud_filter, synth
scene = synth.Scene(seed=1)
poses = synth.trajectory(24)
P0i = np.linalg.inv(poses[0])
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
ctx = capi.Context(0, ts.cuda_stream); fs_ = torch.cuda.Stream(); ctx_f = capi.Context(0, fs_.cuda_stream)
parts = [synth.transform_points(P0i @ poses[k], synth.gen_scan(scene, poses[k], seed=100 + k)) for k in range(0, 21, 2)]
mf = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
d_map = mf.filter_device(torch.from_numpy(np.ascontiguousarray(np.concatenate(parts, 0))).cuda()).clone()
ndt = registration.NDTRegistrationHip(ctx); ndt.SetInputTarget(d_map)
ff = cloud_filter.VoxelFilterHip(ctx_f, 0.5, 0.5, 0.5)
scan = synth.gen_scan(scene, poses[21], seed=121)
d_scan = torch.from_numpy(scan).cuda(); torch.cuda.synchronize()
d_src = ff.filter_device(d_scan).clone()
tiny = d_scan[:200].contiguous()
guess = (P0i @ poses[20]).astype(np.float32)
L = capi.lib()
L.lsa_pipeline_scan_match.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(capi.NdtStats), C.POINTER(C.c_size_t)]
def run(name, scans, count=3000):
    ptrs = (C.c_void_p * count)(*[scans[i % len(scans)].data_ptr() for i in range(count)])
    nn = (C.c_size_t * count)(*[scans[i % len(scans)].shape[0] for i in range(count)])
    pred = np.ascontiguousarray(np.tile(capi.colmajor(guess), (count, 1)).astype(np.float32))
    out = np.zeros((count, 16), dtype=np.float32); st = (capi.NdtStats * count)(); nf = (C.c_size_t * count)()
    bufs = [torch.empty((scan.shape[0], 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = L.lsa_pipeline_scan_match(ff.h, ndt.h, count, ptrs, nn, capi.fptr(pred), bufs[0].data_ptr(), bufs[1].data_ptr(), 0, 1, capi.fptr(out), st, nf)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "us/scan %.1f" % (dt / count * 1e6), "passes", st[count - 1].passes, "nf", nf[count - 1], flush=True)
# baseline: registration alone, back to back
def alone(count=3000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(count): ndt.ScanMatch(d_src, guess)
    torch.cuda.synchronize(); print("match alone us/scan %.1f" % ((time.perf_counter() - t0) / count * 1e6), ndt.last_stats["passes"], flush=True)
alone(); alone()
run("pipeline, full 131k-point scans in the filter", [d_scan])
run("pipeline, 200-point scans in the filter (same launches, no GPU work)", [tiny])
