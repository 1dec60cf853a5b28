"""GPU: the multi-GPU exchange step through REAL RCCL (backend "nccl"), on the one GPU a test box has: a process group of
size 1.  The sum over one rank is the identity, so ScanMatch with the all-reduce callback installed must reproduce the
single-GPU result bit for bit — what this exercises is everything a world of 8 also needs: RCCL initialisation, the
zero-copy alias of the library's 32-double device buffer as a torch tensor, and the ordering of kernel -> collective ->
copy on one HIP stream.  (The world-2 sharding arithmetic is covered on CPU by tests/test_dist_cpu.py over gloo.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import conftest
import torch
import torch.distributed as dist
from lidar_slam_amd import capi, registration, synth, dist as lsa_dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = capi.Context(0, s.cuda_stream)
scene = synth.Scene(seed=1)
s0 = synth.gen_scan(scene, synth.pose_matrix(), seed=10)
s1 = synth.gen_scan(scene, synth.pose_matrix(1.0, 0.05, 0.01, 0.002, -0.003, 0.02), seed=11)[::6]
d_t = torch.from_numpy(s0).cuda(); d_s = torch.from_numpy(np.ascontiguousarray(s1)).cuda()
torch.cuda.synchronize()
ndt = registration.NDTRegistrationHip(ctx)
ndt.SetInputTarget(d_t)
guess = np.eye(4, dtype=np.float32)
_, T_single = ndt.ScanMatch(d_s, guess)
st_single = dict(ndt.last_stats)
ndt.set_allreduce(lsa_dist.make_rccl_allreduce())
lo, hi = lsa_dist.shard_bounds(d_s.shape[0], 0, 1)
_, T_rccl = ndt.ScanMatch(d_s[lo:hi], guess)
st_rccl = dict(ndt.last_stats)
ndt.set_allreduce(None)
# the same exchange step through the library's own RCCL call (lsa_rccl_*, no Python in the per-pass path)
native = lsa_dist.NativeRccl(ctx, 0, 1)
t = torch.arange(32, dtype=torch.float64, device="cuda") + 0.5
torch.cuda.synchronize()
native.allreduce_(t)
torch.cuda.synchronize()
assert torch.equal(t.cpu(), torch.arange(32, dtype=torch.float64) + 0.5)     # the sum over one rank
native.attach(ndt)
_, T_native = ndt.ScanMatch(d_s[lo:hi], guess)
st_native = dict(ndt.last_stats)
capi.lib().lsa_ndt_set_allreduce_rccl(ndt.h, None)
native.close()
assert np.array_equal(T_single, T_native) and st_native["passes"] == st_single["passes"] and st_native["score"] == st_single["score"]
assert np.array_equal(T_single, T_rccl), (T_single, T_rccl)
assert st_single["passes"] == st_rccl["passes"] and st_single["score"] == st_rccl["score"]
ndt.close(); ctx.close()
dist.destroy_process_group()
print("RCCL_PATH_OK", st_rccl["passes"])
'''


@pytest.mark.gpu
def test_scan_match_through_rccl_world_1(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = tmp_path / "child.py"
    script.write_text("ROOT = %r\n" % ROOT + CHILD)
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_PATH_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
