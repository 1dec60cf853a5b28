"""The voxel-index chain at the sizes the pipeline uses, for a per-kernel trace: frame filter (131 k-point scan, 0.5 m), local-map
filter (1.45 M points, 0.3 m), SetInputTarget of the filtered map (48 k points) and of the 2 M-point submap, 2 M-point filter."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa
import torch
from lidar_slam_amd import capi, cloud_filter, registration, workloads as wl

def main():
    poses, scans = wl.cfg2_dataset(workers=1)
    s = torch.cuda.Stream(); torch.cuda.set_stream(s)
    ctx = capi.Context(0, s.cuda_stream)
    ff = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5); mf = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
    d_scan = torch.from_numpy(np.ascontiguousarray(scans[21])).cuda()
    d_map = torch.from_numpy(wl.local_map(poses, scans)).cuda()
    d_2m = torch.from_numpy(wl.submap_2m(poses, scans)).cuda()
    ndt = registration.NDTRegistrationHip(ctx, **wl.NDT_PARAMS)
    out = torch.empty((d_2m.shape[0], 4), dtype=torch.float32, device="cuda")
    def timed(fn, reps):
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    print("frame filter 0.5 m, %d pts: %.4f ms" % (d_scan.shape[0], timed(lambda: ff.filter_device(d_scan, out), 300)))
    print("map filter 0.3 m, %d pts: %.4f ms" % (d_map.shape[0], timed(lambda: mf.filter_device(d_map, out), 30)))
    d_map_f = mf.filter_device(d_map).clone()
    print("SetInputTarget %d pts: %.4f ms" % (d_map_f.shape[0], timed(lambda: ndt.SetInputTarget(d_map_f), 100)))
    print("filter 0.3 m, 2 M pts: %.4f ms -> %d voxels" % (timed(lambda: mf.filter_device(d_2m, out), 20), mf.filter_device(d_2m, out).shape[0]))
    print("SetInputTarget 2 M pts: %.4f ms" % timed(lambda: ndt.SetInputTarget(d_2m), 20))

if __name__ == "__main__":
    main()
