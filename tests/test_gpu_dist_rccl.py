"""GPU: the multi-GPU exchange step through REAL RCCL (backend "nccl"), on the one GPU a test box has: a process group of
size 1.  The sum over one rank is the identity, so ScanMatch (NDT, ICPMANUAL, point-to-plane ICP) and GetFitnessScore with
the all-reduce installed must reproduce the single-GPU result bit for bit — what this exercises is everything a world of 8 also needs: RCCL initialisation, the
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
# fitness (GetFitnessScore: a 2-double sum) and the ICP plugins (one 32-double block per iteration) over the same exchange step
fit_single = ndt.GetFitnessScore()
native = lsa_dist.NativeRccl(ctx, 0, 1)
native.attach(ndt)
_, T_n2 = ndt.ScanMatch(d_s[lo:hi], guess)
fit_native = ndt.GetFitnessScore()
capi.lib().lsa_ndt_set_allreduce_rccl(ndt.h, None)
assert np.array_equal(T_n2, T_single) and fit_native == fit_single, (fit_native, fit_single)
for mode in (0, capi.ICP_MODE_PLANE):
    icp = registration.ICPRegistrationHip(ctx, 2.0, 30, mode=mode)
    icp.SetInputTarget(d_t)
    _, Ti = icp.ScanMatch(d_s, guess)
    st_i = dict(icp.last_stats); fit_i = icp.GetFitnessScore()
    icp.set_allreduce(lsa_dist.make_rccl_allreduce())          # torch.distributed's RCCL group through the callback
    _, Tc = icp.ScanMatch(d_s[lo:hi], guess)
    st_c = dict(icp.last_stats); fit_c = icp.GetFitnessScore()
    icp.set_allreduce(None)
    native.attach_icp(icp)                                      # the library's own ncclAllReduce on the handle's stream
    _, Tn = icp.ScanMatch(d_s[lo:hi], guess)
    st_n = dict(icp.last_stats); fit_n = icp.GetFitnessScore()
    capi.lib().lsa_icp_set_allreduce_rccl(icp.h, None)
    assert st_i["iterations"] > 1
    assert np.array_equal(Ti, Tc) and np.array_equal(Ti, Tn), (mode, Ti, Tc, Tn)
    assert st_i == st_c == st_n, (mode, st_i, st_c, st_n)
    assert fit_i == fit_c == fit_n, (mode, fit_i, fit_c, fit_n)
    icp.close()
native.close()
ndt.close(); ctx.close()
dist.destroy_process_group()
print("RCCL_PATH_OK", st_rccl["passes"])
'''


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.gpu
def test_scan_match_through_rccl_world_1(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = tmp_path / "child.py"
    script.write_text("ROOT = %r\n" % ROOT + CHILD)
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_PATH_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


CHILD2 = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import conftest
import ctypes as C
import torch
import torch.distributed as dist
from lidar_slam_amd import capi, cloud_filter, registration, synth, dist as lsa_dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)                       # both ranks on the ONE GPU of the test box (RCCL refuses that: gloo staged)
dist.init_process_group("gloo", rank=rank, world_size=world)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = capi.Context(0, s.cuda_stream)
fs = torch.cuda.Stream()
ctx_f = capi.Context(0, fs.cuda_stream)
scene = synth.Scene(seed=1)
T = [synth.pose_matrix(), synth.pose_matrix(1.0, 0.05, 0.01, 0.002, -0.003, 0.02), synth.pose_matrix(2.0, 0.11, 0.0, 0.001, -0.002, 0.04)]
scans = [synth.gen_scan(scene, T[i], seed=10 + i, beams=32, az_steps=1024) for i in range(3)]
d_t = torch.from_numpy(scans[0]).cuda()
d_scans = [torch.from_numpy(np.ascontiguousarray(scans[i])).cuda() for i in (1, 2)]
torch.cuda.synchronize()
ff = cloud_filter.VoxelFilterHip(ctx_f, 0.5, 0.5, 0.5)
ndt = registration.NDTRegistrationHip(ctx)
ndt.SetInputTarget(d_t)
L = capi.lib()
L.lsa_pipeline_scan_match.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_float),
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(capi.NdtStats), C.POINTER(C.c_size_t)]
cap = max(x.shape[0] for x in d_scans)
buf = [torch.empty((cap, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
pred = np.stack([capi.colmajor(np.eye(4, dtype=np.float32)), capi.colmajor(T[1].astype(np.float32))]).astype(np.float32)
ptrs = (C.c_void_p * 2)(*[x.data_ptr() for x in d_scans]); nn = (C.c_size_t * 2)(*[x.shape[0] for x in d_scans])

def run(r, w):
    out = np.zeros((2, 16), dtype=np.float32); st = (capi.NdtStats * 2)(); nf = (C.c_size_t * 2)()
    rc = L.lsa_pipeline_scan_match(ff.h, ndt.h, 2, ptrs, nn, capi.fptr(pred), buf[0].data_ptr(), buf[1].data_ptr(), r, w, capi.fptr(out), st, nf)
    ctx.check(rc)
    return out, [st[i].as_dict() for i in range(2)], [int(nf[i]) for i in range(2)]

whole, st_w, nf_w = run(0, 1)                  # every rank: the unsharded reference, no collective
ndt.set_allreduce(lsa_dist.make_gloo_staged_allreduce())
shard, st_s, nf_s = run(rank, world)           # this rank's shard of every filtered scan, one all-reduce per pass
ndt.set_allreduce(None)
assert nf_s == nf_w
for i in range(2):
    assert st_s[i]["iterations"] == st_w[i]["iterations"] and st_s[i]["passes"] == st_w[i]["passes"], (st_s[i], st_w[i])
    assert abs(st_s[i]["pairs_per_pass"] - st_w[i]["pairs_per_pass"]) < 1e-9 * st_w[i]["pairs_per_pass"]   # the all-reduce summed every shard
    assert np.max(np.abs(shard[i] - whole[i])) <= 1e-6, (shard[i] - whole[i])   # the reduction order differs, nothing else
# every rank took the identical solve: compare the poses across the ranks bit for bit
t = torch.from_numpy(shard.copy())
gathered = [torch.zeros_like(t) for _ in range(world)]
dist.all_gather(gathered, t)
assert all(torch.equal(g, gathered[0]) for g in gathered)
ndt.close(); ff.close(); ctx_f.close(); ctx.close()
dist.barrier(); dist.destroy_process_group()
print("PIPELINE_SHARDED_OK rank %d" % rank)
'''


@pytest.mark.gpu
def test_pipeline_scan_match_sharded_two_ranks_one_gpu(tmp_path):
    """lsa_pipeline_scan_match(rank, world) with world = 2: two processes on the one GPU of the box, the 32-double block summed over
    gloo (staged through the host) once per pass — the rehearsal gpurun_out/b2.json ran by hand in round 1, as a test: same
    counts as the unsharded run, poses within reduction-order rounding of it and bit-identical ACROSS the ranks."""
    script = tmp_path / "child2.py"
    script.write_text("ROOT = %r\n" % ROOT + CHILD2)
    procs = []
    port = str(_free_port())
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(r), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("PIPELINE_SHARDED_OK rank %d" % r) in so, (so[-2000:], se[-3000:])
