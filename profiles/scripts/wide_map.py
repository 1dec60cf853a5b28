"""A few derivative passes (kind 0) against the wide map (workloads.wide_map: a target whose 25 MB dense probe table and
16 MB of records cannot live in a 4 MiB per-XCD L2) for PMC collection: argv[1] = random | sorted (source order)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa
import torch
from lidar_slam_amd import capi, registration, workloads as wl

def main():
    order = sys.argv[1] if len(sys.argv) > 1 else "random"
    tgt, src, g = wl.wide_map()
    if order == "sorted":
        src = np.ascontiguousarray(src[np.lexsort((np.floor(src[:, 0]), np.floor(src[:, 1]), np.floor(src[:, 2])))])
    s = torch.cuda.Stream(); torch.cuda.set_stream(s)
    ctx = capi.Context(0, s.cuda_stream)
    ndt = registration.NDTRegistrationHip(ctx, 1.0, 0.1, 0.0, 30, 0.55)
    ndt.SetInputTarget(torch.from_numpy(tgt).cuda())
    ndt.set_source(torch.from_numpy(src).cuda())
    p = np.zeros(6); capi.lib().lsa_matrix_to_pose(capi.fptr(capi.colmajor(g)), capi.dptr(p))
    for _ in range(5):
        out = ndt.derivatives(g, p, 0)
    print(order, "pairs", out[3], "npts", out[4], ndt.grid_info())

if __name__ == "__main__":
    main()
