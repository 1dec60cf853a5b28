# the 60-scan drive of fe_timing.py scan by scan with NOTHING waited for between two Updates (the loop front_end_node runs); traced by
# profiles/scripts/fe_loop_trace.sh
import sys, os, time
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
for rep in range(4):
    fe = front_end.FrontEndHip(ctx)
    fe.SetInitPose(poses[0].astype(np.float32))
    fe.Update(d_scans[0])
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for i in range(1, N):
        t1 = time.perf_counter()
        fe.Update(d_scans[i])
        ts.append((time.perf_counter() - t1, fe.last_key_frame))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ts = np.array(ts)
    steady = ts[44:]
    kf, nk = steady[steady[:, 1] == 1, 0], steady[steady[:, 1] == 0, 0]
    print("loop rep %d: %.2f ms = %.0f scans/s; steady state: call of a key-frame scan %.1f us, of the scan after one %.1f us" % (
        rep, dt * 1e3, (N - 1) / dt, np.median(kf) * 1e6, np.median(nk) * 1e6), flush=True)
    fe.close()
