# (record of an experiment: the LSA_PASS_PC switch existed only while r04_pass_producer_consumer.hip was compiled into the library)
# A/B of the producer / consumer pass kernel (LSA_PASS_PC=1, csrc/ndt_derivatives.hip: k_ndt_pass_pc) against the one-wave form at
# BASELINE.json configs[4] size (2,097,152-point source against the 2 M-point target): sums, pair counts and time per launch.
# Each arm is its own process (the switch is read once); this script is the arm, pass_pc_ab.sh runs both.
import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest
import torch
from lidar_slam_amd import capi, registration, synth
scene = synth.Scene(seed=1)
poses = synth.trajectory(30)
P0inv = np.linalg.inv(poses[0])
parts = [synth.transform_points(P0inv @ poses[k], synth.gen_scan(scene, poses[k], seed=100 + k)) for k in range(0, 17)]
target = np.ascontiguousarray(np.concatenate(parts, 0))[:2000000]
src5 = synth.gen_scan(scene, poses[8], seed=55, beams=128, az_steps=16384, max_range=120.0, clamp=True)
small = int(os.environ.get("AB_SMALL", "0"))
if small:
    src5 = np.ascontiguousarray(src5[:small])
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = capi.Context(0, s.cuda_stream)
d_t = torch.from_numpy(target).to(dev); d_s5 = torch.from_numpy(src5).to(dev)
ndt = registration.NDTRegistrationHip(ctx, 1.0, 0.1, 0.0, 30, 0.55)
ndt.SetInputTarget(d_t)
guess = (P0inv @ poses[8]).astype(np.float32)
ndt.set_source(d_s5)
p = np.zeros(6); Tc = capi.colmajor(guess); capi.lib().lsa_matrix_to_pose(capi.fptr(Tc), capi.dptr(p))
res = []
for rep in range(3):
    (score, g, H, pairs, npts), ms = ndt.derivatives_timed(guess, p, kind=0, repeat=20)
    res.append(ms * 1e3)
print(json.dumps({"pc": os.environ.get("LSA_PASS_PC", "0"), "points": int(npts), "pairs": int(pairs), "score": repr(float(score)), "g": [repr(float(v)) for v in g],
                  "H00": repr(float(H[0, 0])), "H55": repr(float(H[5, 5])), "us_per_launch": [round(v, 2) for v in res]}))
