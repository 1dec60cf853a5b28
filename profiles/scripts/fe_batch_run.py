# the 60-scan drive of fe_timing.py as ONE run of lsa_front_end_update_batch_device, three times (profiles/scripts/fe_batch_trace.sh traces it)
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
for rep in range(3):
    fe = front_end.FrontEndHip(ctx)
    fe.SetInitPose(poses[0].astype(np.float32))
    fe.UpdateBatch(d_scans[:1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Ts, sts = fe.UpdateBatch(d_scans[1:])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("batch rep %d: %.2f ms = %.0f scans/s; iters %s" % (rep, dt * 1e3, (N - 1) / dt, [s["ndt"]["iterations"] for s in sts][:20]), flush=True)
    fe.close()
