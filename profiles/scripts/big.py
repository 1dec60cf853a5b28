import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest
import torch
from lidar_slam_amd import capi, registration, cloud_filter, synth
scene = synth.Scene(seed=1)
poses = synth.trajectory(30)
t0=time.time()
P0inv = np.linalg.inv(poses[0])
parts = [synth.transform_points(P0inv @ poses[k], synth.gen_scan(scene, poses[k], seed=100+k)) for k in range(0, 17)]
target = np.ascontiguousarray(np.concatenate(parts, 0))[:2000000]
src5 = synth.gen_scan(scene, poses[8], seed=55, beams=128, az_steps=16384, max_range=120.0, clamp=True)
src4 = synth.gen_scan(scene, poses[8], seed=56)
print("gen %.1fs target %d src5 %d src4 %d" % (time.time()-t0, target.shape[0], src5.shape[0], src4.shape[0]), flush=True)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = capi.Context(0, s.cuda_stream)
d_t = torch.from_numpy(target).to(dev); d_s5 = torch.from_numpy(src5).to(dev); d_s4 = torch.from_numpy(src4).to(dev)
ndt = registration.NDTRegistrationHip(ctx, 1.0, 0.1, 0.0, 30, 0.55)
torch.cuda.synchronize()
for rep in range(3):
    t0=time.perf_counter(); ndt.SetInputTarget(d_t); torch.cuda.synchronize(); st_ms=(time.perf_counter()-t0)*1e3
print("set_target(2M) %.3f ms" % st_ms, ndt.grid_info(), flush=True)
guess = (P0inv @ poses[8]).astype(np.float32)
res = {}
for name, d_s in (("cfg4_src120k", d_s4), ("cfg5_src2M", d_s5)):
    ndt.set_source(d_s)
    p = np.zeros(6); Tc = capi.colmajor(guess); capi.lib().lsa_matrix_to_pose(capi.fptr(Tc), capi.dptr(p))
    for kind in (0, 1, 2):
        (score, g, H, pairs, npts), ms = ndt.derivatives_timed(guess, p, kind=kind, repeat=20)
        M = int(npts); kbar = pairs / M
        alg = M * (16 + 108 + 48 * kbar)
        print("%s kind%d: %.3f ms/pass  M=%d Kbar=%.2f  alg %.1f MB -> %.1f GB/s (%.2f%% of 8TB/s); pairs/s %.2e" % (name, kind, ms, M, kbar, alg/1e6, alg/ms/1e6, alg/ms/1e6/80, pairs/ms*1e3), flush=True)
        res["%s_kind%d_ms" % (name, kind)] = ms
    t0=time.perf_counter(); ok, T = ndt.ScanMatch(d_s, guess); torch.cuda.synchronize(); al=(time.perf_counter()-t0)*1e3
    st = ndt.last_stats
    print("%s align %.2f ms  iters %d passes %d ls %d" % (name, al, st["iterations"], st["passes"], st["ls_iterations"]), flush=True)
    res[name + "_align_ms"] = al; res[name+"_passes"] = st["passes"]
# filter at 2M
f = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
for rep in range(3):
    t0=time.perf_counter(); out = f.filter_device(d_t); torch.cuda.synchronize(); fm=(time.perf_counter()-t0)*1e3
print("filter 0.3 of 2M: %.3f ms -> %d voxels; alg bytes %.1f MB -> %.1f GB/s" % (fm, out.shape[0], (16*target.shape[0]+16*out.shape[0])/1e6, (16*target.shape[0]+16*out.shape[0])/fm/1e6))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "big.json"), "w"))
