"""GPU: the device-resident front end against a restatement of FrontEnd::Update /
UpdateWithNewFrame (front_end.cpp:88-341, :348-424) assembled from the ORACLE's filter and NDT.

The comparison is a trajectory: 14 scans of the synthetic street, key-frames every 2 m, local map of <= 4 key-frames
here so that the sliding window and the filtered-map branch (local_frame_num lowered, >= 3 frames => filtered, see the
test body) are both exercised in a short run.  Tolerance per pose: 1e-3 m / 1e-4 rad — looser than the single-ScanMatch
bar because each pose feeds the next prediction and every key-frame rebuilds the target from the poses so far.
"""
import numpy as np
import pytest

from lidar_slam_amd import capi, front_end, synth

pytestmark = pytest.mark.gpu


class OracleFrontEnd:
    """Plain-Python restatement of the reference loop on top of oracle_py (the checker)."""

    def __init__(self, cfg, min_frames_to_filter=10):
        import oracle_py
        self.op = oracle_py
        self.o = oracle_py.Oracle()
        self.cfg = cfg
        self.ndt = oracle_py.OracleNDT(cfg.ndt_res, cfg.ndt_step_size, cfg.ndt_trans_eps, cfg.ndt_max_iter, cfg.ndt_outlier_ratio)
        self.frames = []
        self.min_filter = min_frames_to_filter
        self.init = np.eye(4, dtype=np.float32)
        self.last = self.predict = self.last_kf = self.init
        self.step = np.eye(4, dtype=np.float32)

    def _rebuild(self):
        parts = []
        for cloud, pose in self.frames:
            c = cloud.copy()
            x, y, z = cloud[:, 0], cloud[:, 1], cloud[:, 2]
            P = pose.astype(np.float32)
            for r in range(3):  # pcl::transformPointCloud, f32, left to right
                c[:, r] = ((P[r, 0] * x + P[r, 1] * y) + P[r, 2] * z) + P[r, 3]
            parts.append(c)
        m = np.concatenate(parts, 0)
        if len(self.frames) >= self.min_filter and self.cfg.use_map_filter:
            m, _ = self.o.voxel_filter(synth.to_xyzi32(m), float(self.cfg.map_leaf[0]))
        self.ndt.set_input_target(synth.to_xyzi32(m))

    def _add(self, cloud, pose, reset_w):
        c = cloud.copy()
        if reset_w:
            c[:, 3] = 1.0
        self.frames.append((c, pose.copy()))
        while len(self.frames) > self.cfg.local_frame_num:
            self.frames.pop(0)
        self._rebuild()

    def update(self, cloud):
        filt, _ = self.o.voxel_filter(synth.to_xyzi32(cloud), float(self.cfg.frame_leaf[0]))
        if not self.frames:
            self._add(cloud, self.init, False)
            return self.init.copy(), True
        cur, _ = self.ndt.scan_match(synth.to_xyzi32(filt), self.predict)
        self.step = (np.linalg.inv(self.last.astype(np.float64)) @ cur.astype(np.float64)).astype(np.float32)
        self.predict = (cur.astype(np.float32) @ self.step).astype(np.float32)
        self.last = cur
        kf = False
        if np.abs(self.last_kf[:3, 3] - cur[:3, 3]).sum() > self.cfg.key_frame_distance:
            self._add(cloud, cur, bool(self.cfg.reset_intensity))
            self.last_kf = cur
            kf = True
        return cur, kf


def _pose_err(Ta, Tb):
    dt = float(np.linalg.norm(Ta[:3, 3].astype(np.float64) - Tb[:3, 3].astype(np.float64)))
    R = Ta[:3, :3].astype(np.float64).T @ Tb[:3, :3].astype(np.float64)
    v = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return dt, float(np.arctan2(np.linalg.norm(v), (np.trace(R) - 1.0) / 2.0))


def test_front_end_trajectory_matches_oracle_loop(scene):
    poses = synth.trajectory(14)
    scans = [synth.gen_scan(scene, poses[i], seed=300 + i, beams=32, az_steps=900) for i in range(14)]
    for s in scans:
        s[:, 3] = np.random.RandomState(len(s)).uniform(0.1, 1.0, s.shape[0]).astype(np.float32)  # raw "reflectance"
    cfg = front_end.default_config()
    cfg.local_frame_num = 4
    ctx = capi.Context(0)
    fe = front_end.FrontEndHip(ctx, cfg)
    ref = OracleFrontEnd(cfg, min_frames_to_filter=10)  # the reference hard-codes 10 (front_end.cpp:394)
    fe.SetInitPose(np.eye(4, dtype=np.float32))
    n_kf = 0
    worst = (0.0, 0.0)
    for i, s in enumerate(scans):
        ok, T = fe.Update(synth.to_xyzi32(s))
        Tr, kf_ref = ref.update(s)
        st = fe.last_stats
        assert ok is True
        assert bool(st["key_frame"]) == kf_ref, (i, st, kf_ref)
        n_kf += int(kf_ref)
        dt, ang = _pose_err(T, Tr)
        worst = (max(worst[0], dt), max(worst[1], ang))
        assert dt <= 1e-3 and ang <= 1e-4, (i, dt, ang)
        assert st["local_map_frames"] == len(ref.frames)
    assert n_kf >= 5  # the window (4) slid at least once
    # and the odometry is physically right: final pose within 10 cm of the true relative pose
    T_true = np.linalg.inv(poses[0]) @ poses[-1]
    assert np.linalg.norm(T[:3, 3] - T_true[:3, 3]) < 0.15
    fe.close()
    ctx.close()


def test_front_end_filtered_map_branch_and_device_input(scene):
    """>= 10 key-frames: the local map handed to SetInputTarget is the 0.3 m-filtered one (front_end.cpp:408-410);
    device-resident input gives the same poses as host input."""
    import torch
    poses = synth.trajectory(30, step=1.1)  # 2 steps = 2.2 m > key_frame_distance: a key-frame every other scan
    scans = [synth.gen_scan(scene, poses[i], seed=500 + i, beams=32, az_steps=900) for i in range(30)]
    cfg = front_end.default_config()
    ctx = capi.Context(0)
    fe_h = front_end.FrontEndHip(ctx, cfg)
    fe_d = front_end.FrontEndHip(ctx, cfg)
    saw_filtered = False
    for s in scans:
        _, Th = fe_h.Update(synth.to_xyzi32(s))
        _, Td = fe_d.Update(torch.from_numpy(np.ascontiguousarray(s)).cuda())
        assert np.array_equal(Th, Td)
        st = fe_h.last_stats
        if st["local_map_frames"] >= 10:
            assert st["target_points"] < st["local_map_points"]
            saw_filtered = True
        else:
            assert st["target_points"] == st["local_map_points"]
    assert saw_filtered
    T_true = np.linalg.inv(poses[0]) @ poses[-1]
    assert np.linalg.norm(Th[:3, 3] - T_true[:3, 3]) < 0.3
    fe_h.close(); fe_d.close(); ctx.close()


def test_kitti_directory_example(scene, tmp_path):
    """examples/run_kitti_odometry.py end to end: KITTI-format .bin files in a directory -> front end -> poses file in the
    KITTI odometry format; the trajectory matches the drive the scans were generated on."""
    import importlib.util
    import os
    from lidar_slam_amd import cloud_io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_kitti_odometry", os.path.join(root, "examples", "run_kitti_odometry.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    poses = synth.trajectory(8)
    vdir = tmp_path / "velodyne"
    vdir.mkdir()
    for i in range(8):
        s = synth.gen_scan(scene, poses[i], seed=700 + i, beams=32, az_steps=900)
        if i == 3:
            s[5, 1] = np.nan   # the reader's removeNaN path
        cloud_io.write_kitti_bin(str(vdir / ("%06d.bin" % i)), s)
    out = tmp_path / "poses.txt"
    got = ex.run(str(vdir), str(out), log=lambda *a: None)
    rows = np.loadtxt(str(out))
    assert rows.shape == (8, 12) and len(got) == 8
    T_last = np.eye(4)
    T_last[:3, :] = rows[-1].reshape(3, 4)
    T_true = np.linalg.inv(poses[0]) @ poses[-1]
    assert np.linalg.norm(T_last[:3, 3] - T_true[:3, 3]) < 0.15
