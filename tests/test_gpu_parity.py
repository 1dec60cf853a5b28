"""GPU parity tests proper: every check goes through the C-ABI (ctypes -> liblidar_slam_amd.so -> HIP kernels)
and compares with the CPU oracle on the same seeded inputs.

Tolerances (stated per test):
  * voxel filter: integer structure (count, voxel membership, order) exact; coordinates within a few f32 ulp
    (PCL sums in f32 in an unspecified order, the GPU sums exactly and rounds once).
  * NDT voxel table: cell set / counts / valid flags exact; centroid 1e-9 m, inverse covariance 1e-6 relative
    (the oracle reproduces the reference's cancellation-prone formula, the GPU the shift-invariant one).
  * derivative pass: (a) on IDENTICAL voxel tables (the oracle's cells overwritten with the GPU's records) score /
    gradient / Hessian agree to 1e-11 relative to the largest entry (f64 both sides, different summation order);
    (b) on each side's own table 1e-7 relative (the oracle reproduces the reference's cancellation-prone covariance
    formula whose round-off is amplified by the 1/100 eigenvalue floor; the GPU evaluates the shift-invariant form).
  * ScanMatch final pose: <= 1e-4 m and <= 1e-5 rad against the oracle (faithful f64 mode).
"""
import numpy as np
import pytest

from lidar_slam_amd import capi, cloud_filter, registration, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _pose_err(Ta, Tb):
    dt = float(np.linalg.norm(Ta[:3, 3].astype(np.float64) - Tb[:3, 3].astype(np.float64)))
    R = Ta[:3, :3].astype(np.float64).T @ Tb[:3, :3].astype(np.float64)
    # geodesic angle via the skew part (arccos of the trace is ill-conditioned near 0 for f32 matrices)
    v = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    ang = float(np.arctan2(np.linalg.norm(v), (np.trace(R) - 1.0) / 2.0))
    return dt, ang


# --------------------------------------------------------------------------------------------- filter
@pytest.mark.parametrize("leaf", [0.5, 0.3, 1.3])
def test_voxel_filter_matches_oracle(ctx, oracle, scan_pair, leaf):
    s0, s1, _, _ = scan_pair
    pts = synth.to_xyzi32(s1)
    pts[:, 4] = np.random.RandomState(3).uniform(0, 1, pts.shape[0]).astype(np.float32)  # mixed static weights
    ref, rc = oracle.voxel_filter(pts, leaf)
    assert rc == 0
    f = cloud_filter.VoxelFilterHip(ctx, leaf, leaf, leaf)
    ok, out = f.Filter(pts)
    assert ok is True
    assert out.shape[0] == ref.shape[0]  # one output per occupied voxel, same voxel set
    got = out[:, [0, 1, 2, 4]]
    # order == ascending voxel index; values == mean of the members
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    scale = np.maximum(np.abs(ref.astype(np.float64)), 1.0)
    assert np.max(err / scale) < 4e-6, np.max(err / scale)
    assert np.all(out[:, 3] == 1.0)
    f.close()


def test_voxel_filter_edge_cases(ctx, oracle):
    f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    # empty input
    ok, out = f.Filter(np.zeros((0, 8), dtype=np.float32))
    assert ok and out.shape[0] == 0
    # single point, duplicates, negative coordinates, points exactly on voxel faces
    pts = np.array([[0, 0, 0], [0.5, 0.5, 0.5], [-0.5, -0.5, -0.5], [-0.5, -0.5, -0.5], [1e-7, -1e-7, 0.25], [99.75, -99.75, 3.0]], dtype=np.float32)
    p8 = synth.to_xyzi32(np.concatenate([pts, np.arange(6, dtype=np.float32)[:, None]], axis=1))
    ref, _ = oracle.voxel_filter(p8, 0.5)
    ok, out = f.Filter(p8)
    assert out.shape[0] == ref.shape[0]
    assert np.allclose(out[:, [0, 1, 2, 4]], ref, rtol=1e-6, atol=1e-6)
    # NaN rows are skipped like PCL does for non-dense clouds
    p8n = p8.copy()
    p8n[1, 0] = np.nan
    refn, _ = oracle.voxel_filter(p8n, 0.5)
    ok, outn = f.Filter(p8n)
    assert outn.shape[0] == refn.shape[0]
    # in == out aliasing (viewer.cpp:207): result identical
    import ctypes as C
    buf = p8.copy()
    n_out = C.c_size_t(0)
    ctx.check(capi.lib().lsa_voxel_filter_apply(f.h, buf.ctypes.data, buf.shape[0], 32, 16, buf.ctypes.data, buf.shape[0], 32, 16, C.byref(n_out)))
    assert n_out.value == ref.shape[0]
    assert np.allclose(buf[: n_out.value][:, [0, 1, 2, 4]], ref, rtol=1e-6, atol=1e-6)
    # the Python mirror with the caller's own output cloud (written in place, a view is returned), also with in == out
    own = np.zeros((p8.shape[0], 8), dtype=np.float32)
    ok, view = f.Filter(p8, own)
    assert ok and view.base is own and view.shape[0] == ref.shape[0] and np.allclose(view[:, [0, 1, 2, 4]], ref, rtol=1e-6, atol=1e-6)
    both = p8.copy()
    ok, view = f.Filter(both, both)
    assert ok and view.base is both and np.allclose(view[:, [0, 1, 2, 4]], ref, rtol=1e-6, atol=1e-6)
    # leaf too small for the extent: PCL warns and passes the input through
    tiny = cloud_filter.VoxelFilterHip(ctx, 1e-2, 1e-2, 1e-2)
    big = synth.to_xyzi32(np.array([[0, 0, 0, 1], [500, 500, 50, 1]], dtype=np.float32))
    ok, out = tiny.Filter(big)
    assert ok and out.shape[0] == 2 and np.array_equal(out[:, :3], big[:, :3])
    tiny.close()
    f.close()


def test_voxel_filter_idempotent_and_deterministic(ctx, scan_pair):
    s0, _, _, _ = scan_pair
    pts = synth.to_xyzi32(s0)
    f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    _, a = f.Filter(pts)
    _, b = f.Filter(pts)
    assert np.array_equal(a, b)  # fixed-point sums: bitwise reproducible
    perm = np.random.RandomState(0).permutation(pts.shape[0])
    _, c = f.Filter(pts[perm])
    assert np.array_equal(a, c)  # ... and independent of the input order
    f.close()


# --------------------------------------------------------------------------------------------- NDT grid
def test_ndt_grid_matches_oracle(ctx, scan_pair):
    import oracle_py
    s0, _, _, _ = scan_pair
    tgt = synth.to_xyzi32(s0)
    tgt[:, 4] = np.random.RandomState(5).uniform(0, 1, tgt.shape[0]).astype(np.float32)
    o = oracle_py.OracleNDT()
    o.set_input_target(tgt)
    oi, od = o.grid_dump()
    n = registration.NDTRegistrationHip(ctx)
    n.SetInputTarget(tgt)
    gi, gr = n.grid_dump()
    info = n.grid_info()
    occ, valid = o.grid_counts()
    assert info["occupied"] == occ == gi.shape[0]
    assert info["valid"] == valid
    assert np.array_equal(gi[:, :4], oi[:, :4])  # same cells, same order, same point counts
    o_valid = oi[:, 4] >= 6
    assert np.array_equal(gi[:, 4] == 1, o_valid)
    assert np.max(np.abs(gr[:, :3] - od[:, :3])) < 1e-9          # centroids
    assert np.max(np.abs(gr[:, 9] - od[:, 12])) < 1e-9           # mean static weight
    ic = od[:, 3:12].reshape(-1, 3, 3)
    ic_sym = np.stack([ic[:, 0, 0], 0.5 * (ic[:, 0, 1] + ic[:, 1, 0]), 0.5 * (ic[:, 0, 2] + ic[:, 2, 0]), ic[:, 1, 1],
                       0.5 * (ic[:, 1, 2] + ic[:, 2, 1]), ic[:, 2, 2]], axis=1)
    sc = np.max(np.abs(ic_sym[o_valid]), axis=1, keepdims=True)
    rel = np.abs(gr[o_valid, 3:9] - ic_sym[o_valid]) / sc
    assert np.max(rel) < 1e-6, np.max(rel)
    n.close()


# --------------------------------------------------------------------------------------------- derivative pass
@pytest.mark.parametrize("same_table", [True, False])
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_derivative_pass_matches_oracle(ctx, oracle, scan_pair, kind, same_table):
    import oracle_py
    tol = 1e-11 if same_table else 1e-7
    s0, s1, _, _ = scan_pair
    tgt = synth.to_xyzi32(s0)
    tgt[:, 4] = np.random.RandomState(5).uniform(0.2, 1, tgt.shape[0]).astype(np.float32)
    filt, _ = oracle.voxel_filter(synth.to_xyzi32(s1), 0.5)
    src = synth.to_xyzi32(filt)
    o = oracle_py.OracleNDT()
    o.set_input_target(tgt)
    n = registration.NDTRegistrationHip(ctx)
    n.SetInputTarget(tgt)
    n.set_source(src)
    if same_table:
        o.override_cells(*n.grid_dump())
    for p in ([0.0, 0, 0, 0, 0, 0], [0.9, 0.04, 0.0, 0.002, -0.003, 0.02], [1.3, -0.2, 0.05, -0.01, 0.02, -0.03], [0.5, 0.1, 0.0, 3.1, 0.01, 0.0]):
        p = np.array(p)
        T = oracle.pose_to_matrix(p)
        if kind == 2:
            o.derivatives(src, T, p, True)  # the reference's computeHessian reuses the last angle vectors
            Href = o.hessian(src, T, p)
            s, g, H, pairs, npts = n.derivatives(T, p, 2)
            assert np.max(np.abs(H - 0.5 * (Href + Href.T))) <= tol * np.max(np.abs(Href))
            continue
        sref, gref, Href, pairs_ref = o.derivatives(src, T, p, compute_hessian=(kind == 0))
        s, g, H, pairs, npts = n.derivatives(T, p, kind)
        assert int(pairs) == pairs_ref and int(npts) == src.shape[0]
        assert abs(s - sref) <= tol * abs(sref)
        assert np.max(np.abs(g - gref)) <= tol * np.max(np.abs(gref))
        if kind == 0:
            assert np.max(np.abs(H - 0.5 * (Href + Href.T))) <= tol * np.max(np.abs(Href))
    n.close()


def test_derivative_pass_tile_and_round_boundaries(ctx, oracle, scan_pair):
    """The pass kernel works in 14-point tiles and takes pairs off its queue in rounds of 64, carrying what a tile leaves over
    into the next one.  Source sizes around the tile size and its multiples, on a dense target (several pairs per point, so
    that the queue crosses 64 and 128 at different places), against the oracle on identical voxel tables: pair counts
    exact, sums to 1e-11."""
    import oracle_py
    s0, s1, _, _ = scan_pair
    tgt = synth.to_xyzi32(s0)
    filt, _ = oracle.voxel_filter(synth.to_xyzi32(s1), 0.5)
    src_all = synth.to_xyzi32(filt)
    o = oracle_py.OracleNDT()
    o.set_input_target(tgt)
    n = registration.NDTRegistrationHip(ctx)
    n.SetInputTarget(tgt)
    o.override_cells(*n.grid_dump())
    p = np.array([0.9, 0.04, 0.0, 0.002, -0.003, 0.02])
    T = oracle.pose_to_matrix(p)
    rng = np.random.RandomState(5)
    for ns in (1, 2, 13, 14, 15, 27, 28, 29, 41, 42, 43, 55, 56, 57, 64, 70, 111, 112, 113, 14 * 4 * 3 - 1, 14 * 4 * 3, 14 * 4 * 3 + 1, 777):
        src = src_all[rng.choice(src_all.shape[0], ns, replace=False)]
        n.set_source(src)
        for kind in (0, 1):
            sref, gref, Href, pairs_ref = o.derivatives(src, T, p, compute_hessian=(kind == 0))
            sc, g, H, pairs, npts = n.derivatives(T, p, kind)
            assert int(pairs) == pairs_ref and int(npts) == ns, (ns, kind, pairs, pairs_ref)
            assert abs(sc - sref) <= 1e-11 * max(abs(sref), 1e-300), (ns, kind)
            assert np.max(np.abs(g - gref)) <= 1e-11 * max(np.max(np.abs(gref)), 1e-300), (ns, kind)
            if kind == 0:
                assert np.max(np.abs(H - 0.5 * (Href + Href.T))) <= 1e-11 * max(np.max(np.abs(Href)), 1e-300), ns
    n.close()


def test_derivative_pass_deterministic_and_linear(ctx, scan_pair, oracle):
    """Size-independent properties at full scan size (133 k source points, unfiltered):
    bitwise reproducibility, and additivity over a split of the source cloud (the multi-GPU invariant)."""
    s0, s1, _, _ = scan_pair
    tgt = synth.to_xyzi32(s0)
    src = synth.to_xyzi32(s1)
    n = registration.NDTRegistrationHip(ctx)
    n.SetInputTarget(tgt)
    p = np.array([0.9, 0.04, 0.0, 0.002, -0.003, 0.02])
    T = oracle.pose_to_matrix(p)
    n.set_source(src)
    a = n.derivatives(T, p, 0)
    b = n.derivatives(T, p, 0)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    half = src.shape[0] // 2
    n.set_source(src[:half])
    a1 = n.derivatives(T, p, 0)
    n.set_source(src[half:])
    a2 = n.derivatives(T, p, 0)
    assert abs((a1[0] + a2[0]) - a[0]) <= 1e-12 * abs(a[0])
    assert np.max(np.abs(a1[2] + a2[2] - a[2])) <= 1e-11 * np.max(np.abs(a[2]))
    assert int(a1[3] + a2[3]) == int(a[3])
    n.close()


# --------------------------------------------------------------------------------------------- ScanMatch
@pytest.mark.parametrize("guess_kind", ["identity", "offset"])
def test_scan_match_matches_oracle(ctx, oracle, scan_pair, guess_kind):
    """cfg1: filter(0.5) -> SetInputTarget(raw scan0) -> ScanMatch(filtered scan1), YAML parameters.
    Tolerance: 1e-4 m / 1e-5 rad on the final pose; identical iteration / pass counts."""
    import oracle_py
    s0, s1, T0, T1 = scan_pair
    tgt = synth.to_xyzi32(s0)
    f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    _, src = f.Filter(synth.to_xyzi32(s1))
    o = oracle_py.OracleNDT()
    o.set_input_target(tgt)
    guess = np.eye(4, dtype=np.float32)
    if guess_kind == "offset":
        guess = synth.pose_matrix(0.7, -0.1, 0.0, -0.004, 0.003, 0.01).astype(np.float32)
    T_ref, st_ref = o.scan_match(src, guess)
    n = registration.NDTRegistrationHip(ctx)
    assert n.SetInputTarget(tgt) is True
    ok, T = n.ScanMatch(src, guess)
    assert ok is True
    st = n.last_stats
    dt, ang = _pose_err(T, T_ref)
    assert dt <= 1e-4 and ang <= 1e-5, (dt, ang)
    assert st["iterations"] == st_ref["iterations"] and st["passes"] == st_ref["passes"] and st["converged"] == 1
    assert abs(st["trans_probability"] - st_ref["trans_probability"]) <= 1e-8 * abs(st_ref["trans_probability"])
    # and the answer is physically right: relative motion of the synthetic pair within the NDT's own accuracy
    T_true = (np.linalg.inv(T0) @ T1)
    assert np.linalg.norm(T[:3, 3] - T_true[:3, 3]) < 0.1
    n.close()
    f.close()


def test_scan_match_errors(ctx):
    n = registration.NDTRegistrationHip(ctx)
    with pytest.raises(capi.LsaError) as ei:
        n.ScanMatch(np.zeros((4, 8), dtype=np.float32), np.eye(4, dtype=np.float32))
    assert ei.value.code == capi.LSA_ERR_NO_TARGET
    n.close()


def test_voxel_filter_async_pipeline(ctx, scan_pair):
    """lsa_voxel_filter_apply_device_async / _wait: same bits as the synchronous call, on a second context (HIP stream),
    back-to-back over alternating buffers the way bench.py pipelines filter(k+1) under ScanMatch(k)."""
    import ctypes as C
    import torch
    s0, s1, _, _ = scan_pair
    ctx2 = capi.Context(0)  # own stream
    f_sync = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    f_async = cloud_filter.VoxelFilterHip(ctx2, 0.5, 0.5, 0.5)
    d_in = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in (s0, s1)]
    ref = [f_sync.filter_device(d).clone() for d in d_in]
    outs = [torch.empty((d.shape[0], 4), dtype=torch.float32, device="cuda") for d in d_in]
    torch.cuda.synchronize()
    L = capi.lib()
    n = C.c_size_t(0)
    ctx2.check(L.lsa_voxel_filter_apply_device_async(f_async.h, d_in[0].data_ptr(), d_in[0].shape[0], outs[0].data_ptr()))
    # a second call while one is in flight is refused
    assert L.lsa_voxel_filter_apply_device_async(f_async.h, d_in[1].data_ptr(), d_in[1].shape[0], outs[1].data_ptr()) == capi.LSA_ERR_INVALID
    for it in range(6):
        k = it & 1
        ctx2.check(L.lsa_voxel_filter_wait(f_async.h, C.byref(n)))
        assert n.value == ref[k].shape[0]
        got = outs[k][: n.value].clone()
        ctx2.check(L.lsa_voxel_filter_apply_device_async(f_async.h, d_in[1 - k].data_ptr(), d_in[1 - k].shape[0], outs[1 - k].data_ptr()))
        assert torch.equal(got, ref[k])
    ctx2.check(L.lsa_voxel_filter_wait(f_async.h, C.byref(n)))
    assert L.lsa_voxel_filter_wait(f_async.h, C.byref(n)) == capi.LSA_ERR_INVALID  # nothing in flight
    f_async.close(); f_sync.close(); ctx2.close()


def test_row_handover_equals_stream_wait(ctx, oracle, scan_pair, monkeypatch):
    """A pass hands its workgroup rows to the host through check-worded rows that are added as they arrive; with
    LSA_NO_ROW_POLLING the host waits for the stream first (the fallback a late row takes).  Same bits either way."""
    s0, s1, _, _ = scan_pair
    filt, _ = oracle.voxel_filter(synth.to_xyzi32(s1), 0.5)
    src = synth.to_xyzi32(filt)
    g = synth.pose_matrix(0.3, -0.1, 0.02, 0.002, -0.001, 0.015).astype(np.float32)
    out = []
    for wait in (False, True):
        if wait:
            monkeypatch.setenv("LSA_NO_ROW_POLLING", "1")
        n = registration.NDTRegistrationHip(ctx)   # the switch is read when the handle is created
        n.SetInputTarget(synth.to_xyzi32(s0))
        ok, T = n.ScanMatch(src, g)
        out.append((T, dict(n.last_stats), n.derivatives(g, oracle.matrix_to_pose(g), 0)))
        n.close()
    assert np.array_equal(out[0][0], out[1][0])
    assert all(np.array_equal(out[0][1][k], out[1][1][k]) for k in out[0][1])
    for a, b in zip(out[0][2], out[1][2]):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_device_resident_path(ctx, oracle, scan_pair):
    """Same results when clouds are already resident in HBM (torch tensors as plain device memory)."""
    import torch
    s0, s1, _, _ = scan_pair
    f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    _, src_h = f.Filter(synth.to_xyzi32(s1))
    d_s1 = torch.from_numpy(np.ascontiguousarray(s1)).cuda()
    d_src = f.filter_device(d_s1)
    assert np.array_equal(d_src.cpu().numpy(), src_h[:, [0, 1, 2, 4]])
    n_h = registration.NDTRegistrationHip(ctx)
    n_d = registration.NDTRegistrationHip(ctx)
    n_h.SetInputTarget(synth.to_xyzi32(s0))
    d_s0 = torch.from_numpy(np.ascontiguousarray(s0)).cuda()
    n_d.SetInputTarget(d_s0)
    g = np.eye(4, dtype=np.float32)
    _, Th = n_h.ScanMatch(src_h, g)
    _, Td = n_d.ScanMatch(d_src.contiguous(), g)
    assert np.array_equal(Th, Td)
    n_h.close(); n_d.close(); f.close()


def test_voxel_filter_speculative_grid_reuse(oracle, scan_pair):
    """Consecutive scans reuse the previous grid definition (no bounding-box launches); a scan that pokes out of it must
    be detected and rebuilt: every result equals the oracle's, whatever the history of the handle."""
    ctx = capi.Context(0)
    f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    s0, s1, _, _ = scan_pair
    rng = np.random.RandomState(5)
    history = [s0, s1, s0[::2], s1 * np.float32([1.0, 1.0, 1.0, 1.0]),
               s0 + np.float32([30.0, -12.0, 1.5, 0.0]),          # shifted far outside the margin: miss -> exact rebuild
               s0 + np.float32([30.5, -12.0, 1.5, 0.0]),          # inside the new definition again
               np.concatenate([s1, np.float32([[400.0, 0.0, 0.0, 1.0]])]),   # one far outlier
               s1[:50], s0]
    for k, pts in enumerate(history):
        p8 = synth.to_xyzi32(pts)
        ref, _ = oracle.voxel_filter(p8, 0.5)
        _, out = f.Filter(p8)
        assert out.shape[0] == ref.shape[0], (k, out.shape, ref.shape)
        assert np.allclose(out[:, [0, 1, 2, 4]], ref, rtol=2e-6, atol=2e-5), k
    f.close()
    ctx.close()


def test_host_filter_pulled_rows_equal_copied_rows(scene, monkeypatch):
    """lsa_voxel_filter_apply on full-size host clouds: from the second scan of a stream on, the mark kernel pulls the packed rows out of
    pinned memory itself while both host threads are still packing them (chunk flags, no copy command).  The same history through a
    handle with the pull switched off (LSA_FILTER_PULL=0: packed halves sent by copy commands): every output bit for bit the same —
    including a scan shifted out of the reused cell box (detected by the pulling kernel, rebuilt exactly from the rows it left in
    device memory), an xyz-only cloud (12-byte rows, no intensity), and a short cloud in between (below the pull threshold)."""
    scans = [synth.gen_scan(scene, synth.pose_matrix(float(k), 0.1 * k, 0.0, 0.0, 0.0, 0.01 * k), seed=4100 + k) for k in range(5)]
    assert all(sc.shape[0] >= 65536 for sc in scans)
    history = [scans[0], scans[1], scans[2], scans[3] + np.float32([60.0, -25.0, 2.0, 0.0]), scans[4] + np.float32([60.0, -25.0, 2.0, 0.0]),
               scans[1][:3000], scans[2], scans[0]]
    outs = {}
    for name, env in (("pull", None), ("copy", "0")):
        if env is None:
            monkeypatch.delenv("LSA_FILTER_PULL", raising=False)
        else:
            monkeypatch.setenv("LSA_FILTER_PULL", env)
        ctx = capi.Context(0)
        f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
        res = []
        for pts in history:
            _, out = f.Filter(synth.to_xyzi32(pts))
            res.append(out.copy())
        # an xyz-only cloud (12-byte rows, no intensity column): same path, other packing
        xyz = np.ascontiguousarray(scans[0][:, :3])
        _, out3 = f.Filter(xyz)
        res.append(out3.copy())
        outs[name] = (res, f.counters())
        f.close(); ctx.close()
    monkeypatch.delenv("LSA_FILTER_PULL", raising=False)
    (rp, cp), (rc, cc) = outs["pull"], outs["copy"]
    assert cp["pulled"] >= 4 and cc["pulled"] == 0, (cp, cc)
    assert cp["misses"] >= 1 and cp["misses"] == cc["misses"], (cp, cc)
    for k, (a, b) in enumerate(zip(rp, rc)):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), k
    assert rp[0].shape[0] > 5000


def test_pipeline_host_clouds_equal_sequential_host_calls(scene):
    """lsa_pipeline_scan_match_host (host PointXYZI clouds, the next two scans packed and crossing the link while the current one is
    registered; two or three filter handles on contexts of their own) against Filter(host) + ScanMatch(host) one scan at a time: poses,
    statistics and filtered sizes bit for bit — with the device pulling the rows itself and with copy-free uploads only
    (LSA_FILTER_PULL=0), for batches of 1, 2, 3 and 11 scans (every rotation of the buffers)."""
    import os
    poses = synth.trajectory(12, step=1.0)
    scans = [synth.gen_scan(scene, poses[i], seed=5200 + i) for i in range(12)]
    h = [synth.to_xyzi32(sc) for sc in scans]
    ctx, ca, cb, cc = capi.Context(0), capi.Context(0), capi.Context(0), capi.Context(0)
    import torch
    ndt = registration.NDTRegistrationHip(ctx)
    ndt.SetInputTarget(torch.from_numpy(np.ascontiguousarray(scans[0])).cuda())
    guesses = [(np.linalg.inv(poses[0]) @ poses[max(i - 1, 0)]).astype(np.float32) for i in range(12)]
    plain = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    ref = []
    for i in range(1, 12):
        _, f8 = plain.Filter(h[i])
        _, T = ndt.ScanMatch(f8, guesses[i])
        ref.append((T, dict(ndt.last_stats), f8.shape[0]))
    for env in (None, "0"):
        if env is None:
            os.environ.pop("LSA_FILTER_PULL", None)
        else:
            os.environ["LSA_FILTER_PULL"] = env
        fa, fb, fc3 = cloud_filter.VoxelFilterHip(ca, 0.5, 0.5, 0.5), cloud_filter.VoxelFilterHip(cb, 0.5, 0.5, 0.5), cloud_filter.VoxelFilterHip(cc, 0.5, 0.5, 0.5)
        for a, b, flt in ((1, 2, [fa, fb]), (2, 4, [fa, fb, fc3]), (4, 7, [fa, fb]), (1, 12, [fa, fb, fc3]), (1, 12, [fa, fb])):
            Ts, sts, nfs = registration.pipeline_scan_match_host(flt, ndt, h[a:b], guesses[a:b])
            for j in range(b - a):
                T, st, nf = ref[a - 1 + j]
                assert np.array_equal(Ts[j], T), (env, a, b, j)
                assert nfs[j] == nf and sts[j]["iterations"] == st["iterations"] and sts[j]["passes"] == st["passes"] and sts[j]["score"] == st["score"], (env, a, b, j, sts[j], st)
        if env is None:
            assert fa.counters()["pulled"] + fb.counters()["pulled"] + fc3.counters()["pulled"] >= 12
        fa.close(); fb.close(); fc3.close()
    os.environ.pop("LSA_FILTER_PULL", None)
    plain.close(); ndt.close(); ca.close(); cb.close(); cc.close(); ctx.close()


def test_voxel_filter_reserve_changes_no_result(scan_pair):
    """lsa_voxel_filter_reserve (buffers, a bitmap large enough for the first build, helper threads and pinned staging rows ahead of the
    first call): the same outputs bit for bit as an unreserved handle on the host and on the device path, and no bitmap growth."""
    import ctypes as C
    import torch
    s0, s1, _, _ = scan_pair
    L = capi.lib()
    L.lsa_voxel_filter_reserve.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    ctx = capi.Context(0)
    plain = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    ahead = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    ctx.check(L.lsa_voxel_filter_reserve(ahead.h, 160000, 3))
    for pts in (s0, s1, s0[::3], s1):
        p8 = synth.to_xyzi32(pts)
        _, a = plain.Filter(p8)
        _, b = ahead.Filter(p8)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        d = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
        assert torch.equal(plain.filter_device(d), ahead.filter_device(d))
    assert ahead.counters()["growths"] == 0 and plain.counters()["growths"] >= 1
    plain.close(); ahead.close(); ctx.close()


def test_pipeline_batch_equals_sequential_calls(oracle, scene):
    """lsa_pipeline_scan_match (filter of scan k+1 under the registration of scan k, inside the library) returns exactly
    what Filter + ScanMatch called scan by scan return."""
    import ctypes as C
    import torch
    ctx, ctx_f = capi.Context(0), capi.Context(0)
    poses = synth.trajectory(6)
    scans = [synth.gen_scan(scene, poses[i], seed=900 + i, beams=32, az_steps=900) for i in range(6)]
    d_scans = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in scans]
    P0i = np.linalg.inv(poses[0])
    tgt = np.concatenate([synth.transform_points(P0i @ poses[i], scans[i]) for i in range(2)], 0)
    ndt = registration.NDTRegistrationHip(ctx)
    ndt.SetInputTarget(synth.to_xyzi32(tgt))
    ff = cloud_filter.VoxelFilterHip(ctx_f, 0.5, 0.5, 0.5)
    guesses = [(P0i @ poses[max(0, i - 1)]).astype(np.float32) for i in range(6)]
    torch.cuda.synchronize()
    seq = []
    for i in range(6):
        f = ff.filter_device(d_scans[i]).clone()
        torch.cuda.synchronize()
        _, T = ndt.ScanMatch(f, guesses[i])
        seq.append((T, f.shape[0], ndt.last_stats["passes"], dict(ndt.last_stats)))
    L = capi.lib()
    L.lsa_pipeline_scan_match.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_float),
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(capi.NdtStats), C.POINTER(C.c_size_t)]
    n = 6
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in d_scans])
    nn = (C.c_size_t * n)(*[t.shape[0] for t in d_scans])
    pred = np.ascontiguousarray(np.stack([capi.colmajor(g) for g in guesses]).astype(np.float32))
    out = np.zeros((n, 16), dtype=np.float32)
    st = (capi.NdtStats * n)()
    nf = (C.c_size_t * n)()
    cap = max(t.shape[0] for t in d_scans)
    bufs = [torch.empty((cap, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    rc = L.lsa_pipeline_scan_match(ff.h, ndt.h, n, ptrs, nn, capi.fptr(pred), bufs[0].data_ptr(), bufs[1].data_ptr(), 0, 1, capi.fptr(out), st, nf)
    assert rc == 0
    for i in range(6):
        assert int(nf[i]) == seq[i][1] and st[i].passes == seq[i][2]
        assert np.array_equal(capi.from_colmajor(out[i]), seq[i][0])
        # the statistics too: inside a batch the final derivative pass of a ScanMatch is not waited for (it only feeds them);
        # its sums complete the block behind the next scan's first mail, or before the batch returns
        ref_st = seq[i][3]
        got = st[i].as_dict()
        for key in ("iterations", "ls_iterations", "converged", "score", "trans_probability", "pairs_per_pass"):
            assert got[key] == ref_st[key], (i, key, got[key], ref_st[key])
        assert np.array_equal(got["final_p"], ref_st["final_p"])
    # argument checking
    assert L.lsa_pipeline_scan_match(ff.h, ndt.h, n, ptrs, nn, capi.fptr(pred), bufs[0].data_ptr(), bufs[1].data_ptr(), 2, 2, capi.fptr(out), st, nf) != 0
    ndt.close(); ff.close(); ctx.close(); ctx_f.close()


def test_concurrent_pipelines_on_one_gpu(scene):
    """Three independent scan streams (own contexts / HIP streams / filter / registration handles), each driven by its own
    host thread through lsa_pipeline_scan_match at the same time, return bit for bit what one stream returns alone: handles
    share nothing but the device (bench.py's `concurrent_streams`)."""
    import ctypes as C
    import threading
    import torch
    poses = synth.trajectory(6)
    scans = [synth.gen_scan(scene, poses[i], seed=940 + i, beams=32, az_steps=900) for i in range(6)]
    d_scans = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in scans]
    P0i = np.linalg.inv(poses[0])
    tgt = synth.to_xyzi32(np.concatenate([synth.transform_points(P0i @ poses[i], scans[i]) for i in range(2)], 0))
    guesses = np.ascontiguousarray(np.stack([capi.colmajor((P0i @ poses[max(0, i - 1)]).astype(np.float32)) for i in range(6)]).astype(np.float32))
    cap = max(t.shape[0] for t in d_scans)
    L = capi.lib()
    L.lsa_pipeline_scan_match.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_float),
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(capi.NdtStats), C.POINTER(C.c_size_t)]
    reps = 40   # 240 scans per stream: long enough for the three streams to really overlap

    class Pipe:
        def __init__(self, sid):
            self.ctx, self.ctx_f = capi.Context(0), capi.Context(0)
            self.ndt = registration.NDTRegistrationHip(self.ctx)
            self.ndt.SetInputTarget(tgt)
            self.ff = cloud_filter.VoxelFilterHip(self.ctx_f, 0.5, 0.5, 0.5)
            self.idx = [(i + sid) % 6 for i in range(6 * reps)]
            n = len(self.idx)
            self.ptrs = (C.c_void_p * n)(*[d_scans[k].data_ptr() for k in self.idx])
            self.nn = (C.c_size_t * n)(*[d_scans[k].shape[0] for k in self.idx])
            self.pred = np.ascontiguousarray(guesses[self.idx])
            self.out = np.zeros((n, 16), dtype=np.float32)
            self.st = (capi.NdtStats * n)()
            self.nf = (C.c_size_t * n)()
            self.bufs = [torch.empty((cap, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
            self.rc = None

        def run(self):
            self.rc = L.lsa_pipeline_scan_match(self.ff.h, self.ndt.h, len(self.idx), self.ptrs, self.nn, capi.fptr(self.pred), self.bufs[0].data_ptr(),
                                                self.bufs[1].data_ptr(), 0, 1, capi.fptr(self.out), self.st, self.nf)

        def close(self):
            self.ndt.close(); self.ff.close(); self.ctx.close(); self.ctx_f.close()

    torch.cuda.synchronize()
    alone = Pipe(0)
    alone.run()
    assert alone.rc == 0
    ref = {k: (alone.out[i].copy(), int(alone.nf[i]), alone.st[i].passes) for i, k in enumerate(alone.idx[:6])}
    pipes = [Pipe(s) for s in range(3)]
    th = [threading.Thread(target=p.run) for p in pipes]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for p in pipes:
        assert p.rc == 0
        for i, k in enumerate(p.idx):
            assert np.array_equal(p.out[i], ref[k][0]) and int(p.nf[i]) == ref[k][1] and p.st[i].passes == ref[k][2]
        p.close()
    alone.close()


def test_pipeline_from_a_fresh_thread_uses_the_contexts_device(scene):
    """lsa_pipeline_scan_match allocates its third filtered-cloud buffer before any call has entered the filter's context; a worker
    thread that has never touched HIP has device 0 current, whatever device the context lives on.  The batch is driven from a fresh
    thread — on the LAST visible device too when the box has more than one (a buffer on the wrong device faults or gives other poses;
    a one-GPU box only runs the device-0 leg) — and must return the poses of the calling thread's run."""
    import ctypes as C
    import threading
    import torch
    poses = synth.trajectory(4)
    scans = [synth.gen_scan(scene, poses[i], seed=960 + i, beams=32, az_steps=900) for i in range(4)]
    P0i = np.linalg.inv(poses[0])
    tgt = synth.to_xyzi32(np.concatenate([synth.transform_points(P0i @ poses[i], scans[i]) for i in range(2)], 0))
    guesses = np.ascontiguousarray(np.stack([capi.colmajor((P0i @ poses[max(0, i - 1)]).astype(np.float32)) for i in range(4)]).astype(np.float32))
    L = capi.lib()
    L.lsa_pipeline_scan_match.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_float),
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(capi.NdtStats), C.POINTER(C.c_size_t)]
    idx = [i % 4 for i in range(12)]

    def batch(dev, in_thread):
        with torch.cuda.device(dev):
            d_scans = [torch.from_numpy(np.ascontiguousarray(s)).to("cuda:%d" % dev) for s in scans]
            bufs = [torch.empty((max(t.shape[0] for t in d_scans), 4), dtype=torch.float32, device="cuda:%d" % dev) for _ in range(2)]
            torch.cuda.synchronize(dev)
        ctx, ctx_f = capi.Context(dev), capi.Context(dev)
        ndt = registration.NDTRegistrationHip(ctx)
        ndt.SetInputTarget(tgt)
        ff = cloud_filter.VoxelFilterHip(ctx_f, 0.5, 0.5, 0.5)
        n = len(idx)
        ptrs = (C.c_void_p * n)(*[d_scans[k].data_ptr() for k in idx])
        nn = (C.c_size_t * n)(*[d_scans[k].shape[0] for k in idx])
        pred = np.ascontiguousarray(guesses[idx])
        out = np.zeros((n, 16), dtype=np.float32)
        st, nf = (capi.NdtStats * n)(), (C.c_size_t * n)()
        res = {}

        def run():
            res["rc"] = L.lsa_pipeline_scan_match(ff.h, ndt.h, n, ptrs, nn, capi.fptr(pred), bufs[0].data_ptr(), bufs[1].data_ptr(), 0, 1, capi.fptr(out), st, nf)
        if in_thread:
            t = threading.Thread(target=run)   # a thread that has made no HIP call: its current device is 0
            t.start(); t.join()
        else:
            run()
        assert res["rc"] == 0, L.lsa_last_error(ctx_f.h)
        ndt.close(); ff.close(); ctx_f.close(); ctx.close()
        return out.copy(), [int(v) for v in nf]
    ref, ref_n = batch(0, False)
    devs = [0] + ([torch.cuda.device_count() - 1] if torch.cuda.device_count() > 1 else [])
    for dev in devs:
        got, got_n = batch(dev, True)
        assert got_n == ref_n and np.array_equal(got, ref), dev


def test_bitmap_probe_path_matches_dense_table_path(oracle, scan_pair):
    """Targets whose cell box is too large for the dense probe table fall back to the bitmap probe; forced here on a small
    target: both structures must select the same (point, voxel) pairs and give the same sums and the same ScanMatch."""
    import oracle_py
    ctx = capi.Context(0)
    s0, s1, _, _ = scan_pair
    tgt = synth.to_xyzi32(s0)
    filt, _ = oracle.voxel_filter(synth.to_xyzi32(s1), 0.5)
    src = synth.to_xyzi32(filt)
    T = synth.pose_matrix(0.4, 0.02, 0.0, 0.001, -0.002, 0.01).astype(np.float32)
    p = oracle.matrix_to_pose(T)
    res = {}
    for name, limit in (("dense", 16 << 20), ("bitmap", 0)):
        n = registration.NDTRegistrationHip(ctx)
        ctx.check(capi.lib().lsa_ndt_set_dense_limit(n.h, limit))
        n.SetInputTarget(tgt)
        n.set_source(src)
        out = [n.derivatives(T, p, kind) for kind in (0, 1, 2)]
        _, Tm = n.ScanMatch(src, np.eye(4, dtype=np.float32))
        res[name] = (out, Tm, dict(n.last_stats))
        n.close()
    for kind in range(3):
        a, b = res["dense"][0][kind], res["bitmap"][0][kind]
        assert int(a[3]) == int(b[3]) and int(a[4]) == int(b[4])                       # same pairs, same points
        assert abs(a[0] - b[0]) <= 1e-12 * max(1.0, abs(a[0]))
        assert np.max(np.abs(a[1] - b[1])) <= 1e-11 * max(1.0, np.max(np.abs(a[1])))
        assert np.max(np.abs(a[2] - b[2])) <= 1e-11 * max(1.0, np.max(np.abs(a[2])))
    assert res["dense"][2]["passes"] == res["bitmap"][2]["passes"]
    assert np.max(np.abs(res["dense"][1] - res["bitmap"][1])) <= 1e-6
    o = oracle_py.OracleNDT()
    o.set_input_target(tgt)
    Tr, _ = o.scan_match(src, np.eye(4, dtype=np.float32))
    assert np.linalg.norm(res["bitmap"][1][:3, 3] - Tr[:3, 3]) <= 1e-4
    ctx.close()
