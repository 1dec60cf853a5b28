"""GPU: BoxFilter (box_filter.cpp -> pcl::CropBox, UNPINNED third-party) against the oracle: same points, same order."""
import numpy as np
import pytest

from lidar_slam_amd import capi, cloud_filter, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def test_box_filter_matches_oracle(ctx, scan_pair):
    import oracle_py
    s0 = scan_pair[0].copy()
    s0[::5, 3] = 0.25
    s0[11, 1] = np.nan
    s0[12, 0] = np.inf
    p8 = synth.to_xyzi32(s0)
    size = [-30.0, 30.0, -10.0, 12.0, -2.0, 1.0]    # matching.yaml-style [min_x, max_x, min_y, max_y, min_z, max_z]
    f = cloud_filter.BoxFilterHip(ctx, node={"box_filter_size": size})
    assert np.array_equal(f.GetEdge(), np.float32(size))
    for origin in ((0.0, 0.0, 0.0), (12.5, -3.0, 0.5)):
        f.SetOrigin(origin)
        e = f.GetEdge()
        assert np.array_equal(e, np.float32(size) + np.repeat(np.float32(origin), 2))   # box_filter.cpp:62-68
        ref = oracle_py.box_filter(p8, size, origin)
        _, out = f.Filter(p8)
        assert 0 < ref.shape[0] < p8.shape[0]
        assert np.array_equal(out[:, [0, 1, 2, 4]], ref)          # bit-exact copy, input order
    # in == out at the boundary (viewer.cpp style) and points exactly on the faces are kept (inclusive comparison)
    f.SetOrigin((0.0, 0.0, 0.0))
    edge_pts = synth.to_xyzi32(np.float32([[-30, 0, 0, 1], [30, 12, 1, 1], [30.000002, 0, 0, 1], [0, -10.000001, 0, 1], [0, 0, -2, 1]]))
    _, out = f.Filter(edge_pts)
    assert np.array_equal(out[:, :3], edge_pts[[0, 1, 4], :3])
    f.SetSize([-1.0, 1.0, -1.0, 1.0, -1.0, 1.0])
    _, out = f.Filter(p8)
    assert np.array_equal(out[:, [0, 1, 2, 4]], oracle_py.box_filter(p8, [-1, 1, -1, 1, -1, 1]))
    _, out = f.Filter(p8[:0])
    assert out.shape[0] == 0
    f.close()


def test_box_filter_device_path_at_map_size(ctx, scene):
    """2.4 M-point map crop (Matching::ResetLocalMap's job): order preserved, against a torch mask on the same data."""
    import torch
    poses = synth.trajectory(20)
    parts = [synth.transform_points(poses[k], synth.gen_scan(scene, poses[k], seed=300 + k)) for k in range(20)]
    cloud = np.ascontiguousarray(np.concatenate(parts, 0))
    d = torch.from_numpy(cloud).cuda()
    size = np.float32([-20, 25, -15, 15, -3, 4])
    origin = np.float32([8.0, 1.0, 0.0])
    f = cloud_filter.BoxFilterHip(ctx, size)
    f.SetOrigin(origin)
    torch.cuda.synchronize()
    out = f.filter_device(d)
    e = f.GetEdge()
    m = (d[:, 0] >= e[0]) & (d[:, 0] <= e[1]) & (d[:, 1] >= e[2]) & (d[:, 1] <= e[3]) & (d[:, 2] >= e[4]) & (d[:, 2] <= e[5])
    assert torch.equal(out, d[m])
    assert 1000 < out.shape[0] < d.shape[0]
    f.close()
