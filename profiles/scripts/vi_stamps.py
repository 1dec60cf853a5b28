"""Phase stamps of the 2 M-point voxel-filter build (lsa_voxel_filter_stamps): where a workgroup of k_mark / k_accum spends its time and
when the workgroups run relative to one another.  One build with stamps on (after warm-up builds without)."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401
import torch
from lidar_slam_amd import capi, cloud_filter, synth

scene = synth.Scene(seed=1)
poses = synth.trajectory(30)
P0inv = np.linalg.inv(poses[0])
parts = [synth.transform_points(P0inv @ poses[k], synth.gen_scan(scene, poses[k], seed=100 + k)) for k in range(0, 17)]
target = np.ascontiguousarray(np.concatenate(parts, 0))[:2000000]
if os.environ.get("SCAN"):   # one raw scan (the frame filter's input) instead of the 2 M-point map
    target = np.ascontiguousarray(synth.gen_scan(scene, poses[8], seed=56))
ctx = capi.Context(0)
d_t = torch.from_numpy(target).cuda()
leaf = float(os.environ.get("LEAF", "0.3"))
f = cloud_filter.VoxelFilterHip(ctx, leaf, leaf, leaf)
for _ in range(5):
    out = f.filter_device(d_t)
L = capi.lib()
L.lsa_voxel_filter_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t)]
per = C.c_size_t(0)
ctx.check(L.lsa_voxel_filter_stamps(f.h, 1, None, 0, C.byref(per)))
for _ in range(3):
    out = f.filter_device(d_t)
buf = (C.c_uint64 * (3 * per.value))()
ctx.check(L.lsa_voxel_filter_stamps(f.h, 0, buf, 3 * per.value, C.byref(per)))
a = np.frombuffer(buf, dtype=np.uint64).reshape(3, -1, 8).astype(np.int64)
names = (("k_mark", ["clear", "points", "barrier", "flush issue", "flush ack"]),
         ("k_accum", ["clear", "points", "barrier", "ranks", "flush issue", "flush ack"]),
         ("k_scan", ["definition", "front count", "own tile", "prefix known", "prefix out", "cells issue", "cells ack"]))
for k, (name, ph) in enumerate(names):
    s = a[k]
    s = s[(s[:, 0] != 0) & (s[:, len(ph)] != 0)]   # (k_scan: workgroups beyond the bitmap's tiles leave at once)
    if not len(s):
        continue
    t0 = s[:, 0].min()
    nph = len(ph)
    end = s[:, nph].max()
    print("%s: %d workgroups, first entry -> last flush ack %.2f us" % (name, len(s), (end - t0) / 100.0))
    starts = np.sort((s[:, 0] - t0) / 100.0)
    print("  entry times (us after the first): median %.2f, 75 %% %.2f, 90 %% %.2f, max %.2f" % (np.median(starts), np.percentile(starts, 75), np.percentile(starts, 90), starts.max()))
    life = (s[:, nph] - s[:, 0]) / 100.0
    print("  workgroup lifetime: median %.2f us, max %.2f" % (np.median(life), life.max()))
    for i, p in enumerate(ph):
        d = (s[:, i + 1] - s[:, i]) / 100.0
        print("  %-12s median %.2f us, mean %.2f, 90 %% %.2f, max %.2f" % (p, np.median(d), d.mean(), np.percentile(d, 90), d.max()))
    if name == "k_accum":
        order = np.argsort(life)
        slow = s[order[-20:]]
        print("  runs that found no slot (went straight to memory): total %d in %d workgroups; the 20 slowest workgroups: %s" % (
            s[:, 7].sum(), (s[:, 7] > 0).sum(), " ".join(str(int(x)) for x in slow[:, 7])))
        print("  the 20 slowest: lifetimes %s" % " ".join("%.1f" % x for x in life[order[-20:]]))
        for i, p in enumerate(ph):
            print("    %-12s of the 20 slowest: median %.2f" % (p, np.median((slow[:, i + 1] - slow[:, i]) / 100.0)))
    early = s[(s[:, 0] - t0) < 100]   # workgroups that started within the first microsecond
    late = s[(s[:, 0] - t0) >= 100]
    for nm, g in (("first round", early), ("later", late)):
        if len(g):
            print("  %s (%d): " % (nm, len(g)) + ", ".join("%s %.2f" % (p, np.median((g[:, i + 1] - g[:, i]) / 100.0)) for i, p in enumerate(ph)))
m, ac = a[0], a[1]
m, ac = m[m[:, 0] != 0], ac[ac[:, 0] != 0]
if len(m) and len(ac):
    print("k_mark first entry -> k_mark last flush ack %.2f us -> k_accum first entry %.2f us (k_scan runs in between) -> k_accum last flush ack %.2f us" % (
        (m[:, 5].max() - m[:, 0].min()) / 100.0, (ac[:, 0].min() - m[:, 0].min()) / 100.0, (ac[:, 6].max() - m[:, 0].min()) / 100.0))
f.close(); ctx.close()
