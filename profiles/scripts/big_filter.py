"""2 M-point voxel filter + target build, for rocprofv3 --kernel-trace --stats (per-kernel breakdown at cfg4/cfg5 size)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401  (puts the package alias on the path)
from lidar_slam_amd import capi, registration, cloud_filter, synth

scene = synth.Scene(seed=1)
poses = synth.trajectory(30)
P0inv = np.linalg.inv(poses[0])
parts = [synth.transform_points(P0inv @ poses[k], synth.gen_scan(scene, poses[k], seed=100 + k)) for k in range(0, 17)]
target = np.ascontiguousarray(np.concatenate(parts, 0))[:2000000]
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = capi.Context(0, s.cuda_stream)
d_t = torch.from_numpy(target).to(dev)
reps = int(os.environ.get("REPS", "20"))
for leaf in [float(x) for x in os.environ.get("LEAVES", "0.3,1.3").split(",") if x]:
    f = cloud_filter.VoxelFilterHip(ctx, leaf, leaf, leaf)
    for _ in range(3): out = f.filter_device(d_t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = f.filter_device(d_t)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / reps
    alg = 16 * target.shape[0] + 16 * out.shape[0]
    print("filter %.1f m of 2M: %.3f ms -> %d voxels; alg %.1f MB -> %.1f GB/s" % (leaf, ms, out.shape[0], alg / 1e6, alg / ms / 1e6), flush=True)
if os.environ.get("NO_TARGET"):
    sys.exit(0)
ndt = registration.NDTRegistrationHip(ctx, 1.0, 0.1, 0.0, 30, 0.55)
for _ in range(3): ndt.SetInputTarget(d_t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): ndt.SetInputTarget(d_t)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / reps
gi = ndt.grid_info()
print("set_target(2M): %.3f ms" % ms, gi, flush=True)
