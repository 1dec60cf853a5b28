# Where the three kernels of an incremental local-map update (csrc/local_map.hip) spend their time: phase stamps of every workgroup
# (lsa_front_end_map_stamps: the device's 100 MHz wall clock at phase boundaries) on the steady-state key-frames of the 60-scan drive.
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, torch
from lidar_slam_amd import capi, front_end, synth
scene = synth.Scene(seed=1)
N = 60
poses = synth.trajectory(N, step=1.0)
scans = [synth.gen_scan(scene, poses[i], seed=500 + i) for i in range(N)]
ctx = capi.Context(0)
d_scans = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in scans]
torch.cuda.synchronize()
L = capi.lib()
L.lsa_front_end_map_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t)]
names = {0: ["entry", "loads issued", "aggregated in LDS", "lines found", "atomics back", "region written"],
         1: ["entry", "lines loaded", "N sums in LDS", "N lines found", "N atomics back", "region written"],
         2: ["entry", "regions scanned", "line loaded", "record computed", "stores drained", "done counted", "status published"]}
kern = ["k_lm_points", "k_lm_filter_dirty", "k_lm_ndt_dirty"]
for rep in range(2):
    fe = front_end.FrontEndHip(ctx)
    fe.SetInitPose(poses[0].astype(np.float32))
    acc = {k: [] for k in range(3)}
    for i in range(N):
        fe.Update(d_scans[i])
        torch.cuda.synchronize()
        if i == 44:
            ctx.check(L.lsa_front_end_map_stamps(fe.h, 1, None, 0, None))
        if i > 44 and fe.last_key_frame:
            per = C.c_size_t(0)
            buf = np.zeros(3 * 4096 * 8, dtype=np.uint64)
            ctx.check(L.lsa_front_end_map_stamps(fe.h, 1, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size, C.byref(per)))
            for k in range(3):
                a = buf[k * per.value:(k + 1) * per.value].reshape(-1, 8).astype(np.float64)
                a = a[a[:, 0] > 0]
                if len(a):
                    acc[k].append(a)
    fe.close()
    if rep == 0:
        continue
    t_prev_end = None
    for k in range(3):
        print(kern[k])
        rows = acc[k]
        n_wg = np.median([len(a) for a in rows])
        starts = np.array([a[:, 0].min() for a in rows])
        for j, nm in enumerate(names[k]):
            med, mx, last = [], [], []
            for a in rows:
                v = a[:, j]
                v = v[v > 0]
                if len(v) == 0:
                    continue
                rel = (v - a[:, 0].min()) / 100.0   # us since the kernel's first workgroup started
                med.append(np.median(rel)); mx.append(rel.max())
            if med:
                print("   %-20s median over workgroups %6.2f us   slowest workgroup %6.2f us   (%d updates, ~%d workgroups)" % (nm, np.median(med), np.median(mx), len(med), n_wg))
    # gaps between the kernels of one update
    g01 = [b[:, 0].min() - a[:, 5].max() for a, b in zip(acc[0], acc[1])]
    g12 = [b[:, 0].min() - a[:, 5].max() for a, b in zip(acc[1], acc[2])]
    tot = [max(c[:, 4].max(), c[:, 6].max()) - a[:, 0].min() for a, c in zip(acc[0], acc[2])]
    print("gap points -> filter_dirty %.2f us, filter_dirty -> ndt_dirty %.2f us; first workgroup of k_lm_points to the published status %.2f us" % (
        np.median(g01) / 100, np.median(g12) / 100, np.median(tot) / 100))
