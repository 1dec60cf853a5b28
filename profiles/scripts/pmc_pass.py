# runs a few cfg5-sized derivative passes (kind 0) for PMC collection
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest
import torch
from lidar_slam_amd import capi, registration, synth
CACHE = "/tmp/pmc_pass_cache.npz"   # (a PMC collection runs this script once per counter group)
poses = synth.trajectory(30)
P0inv = np.linalg.inv(poses[0])
if os.path.exists(CACHE):
    z = np.load(CACHE); target, src5 = z["target"], z["src5"]
else:
    scene = synth.Scene(seed=1)
    parts = [synth.transform_points(P0inv @ poses[k], synth.gen_scan(scene, poses[k], seed=100+k)) for k in range(0, 17)]
    target = np.ascontiguousarray(np.concatenate(parts, 0))[:2000000]
    src5 = synth.gen_scan(scene, poses[8], seed=55, beams=128, az_steps=16384, max_range=120.0, clamp=True)
    np.savez(CACHE, target=target, src5=src5)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = capi.Context(0, s.cuda_stream)
d_t = torch.from_numpy(target).to(dev); d_s5 = torch.from_numpy(src5).to(dev)
ndt = registration.NDTRegistrationHip(ctx, 1.0, 0.1, 0.0, 30, 0.55)
ndt.SetInputTarget(d_t)
guess = (P0inv @ poses[8]).astype(np.float32)
ndt.set_source(d_s5)
p = np.zeros(6); Tc = capi.colmajor(guess); capi.lib().lsa_matrix_to_pose(capi.fptr(Tc), capi.dptr(p))
for i in range(5):
    out = ndt.derivatives(guess, p, 0)
print("pairs", out[3], "npts", out[4])
