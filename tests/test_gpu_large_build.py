"""GPU: the voxel-index build of LARGE clouds (>= 512 k points; voxel_index.hip: byte marks + pack kernel, sums into rows keyed by the
cell for VoxelFilter, rank-keyed rows behind the scan for SetInputTarget) against the small-cloud build and the oracle.

The two builds share every per-point expression and sum exact integers, so wherever both can run their outputs must agree bit for
bit; the oracle pins a 600 k-point case (seconds on the CPU); at 2 M points the properties of tests/test_gpu_edge_and_scale.py apply.
"""
import numpy as np
import pytest

from lidar_slam_amd import capi, cloud_filter, registration, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cloud600k(scene):
    poses = synth.trajectory(8)
    P0inv = np.linalg.inv(poses[0])
    parts = [synth.transform_points(P0inv @ poses[k], synth.gen_scan(scene, poses[k], seed=900 + k)) for k in range(5)]
    c = np.ascontiguousarray(np.concatenate(parts, 0))[:600000]
    rng = np.random.RandomState(5)
    c[:, 3] = rng.uniform(0, 1, c.shape[0]).astype(np.float32)   # mixed static weights
    return c


def _filter_device(ctx, pts, leaf, env=None, monkeypatch=None, keep=False):
    import torch
    if monkeypatch is not None:
        if env is None:
            monkeypatch.delenv("LSA_VI_BY_CELL", raising=False)
        else:
            monkeypatch.setenv("LSA_VI_BY_CELL", env)
    f = cloud_filter.VoxelFilterHip(ctx, leaf, leaf, leaf)
    d = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
    outs = [f.filter_device(d).clone().cpu().numpy() for _ in range(3)]   # first build (bounding box), then two on the reused cell box
    counters = f.counters()
    f.close()
    return outs, counters


@pytest.mark.parametrize("leaf", [0.3, 1.3])
def test_large_filter_rows_by_cell_equal_rows_by_rank_and_the_oracle(scene, oracle, cloud600k, monkeypatch, leaf):
    """600 k points through VoxelFilter: rows keyed by the cell (one pass over the points) and rows keyed by the rank (marks, scan,
    then sums) give the same bits, build after build; the oracle's voxel set and means within the bar of the small-cloud test."""
    ctx = capi.Context(0)
    by_cell, c1 = _filter_device(ctx, cloud600k, leaf, None, monkeypatch)
    by_rank, c0 = _filter_device(ctx, cloud600k, leaf, "0", monkeypatch)
    monkeypatch.delenv("LSA_VI_BY_CELL", raising=False)
    for a in by_cell[1:] + by_rank:
        assert a.shape == by_cell[0].shape and np.array_equal(a.view(np.uint32), by_cell[0].view(np.uint32))
    ref, rc = oracle.voxel_filter(synth.to_xyzi32(cloud600k), leaf)
    assert rc == 0 and ref.shape[0] == by_cell[0].shape[0]
    err = np.abs(by_cell[0].astype(np.float64) - ref.astype(np.float64)) / np.maximum(np.abs(ref.astype(np.float64)), 1.0)
    assert err.max() < 4e-6, err.max()
    ctx.close()


def test_large_filter_equals_small_path_on_the_same_points(scene, cloud600k):
    """The same 600 k points as ONE large build and as the small-cloud build of their first 500 k + the large build's view of it:
    a cloud just below the threshold (small path: LDS-deduplicated bitmap atomics, rank-keyed rows) and the same cloud padded with
    NaN rows above it (large path; non-finite rows are skipped) must give identical outputs."""
    import torch
    ctx = capi.Context(0)
    below = cloud600k[:500000]
    assert below.shape[0] < 512 * 1024
    padded = np.concatenate([below, np.full((40000, 4), np.nan, dtype=np.float32)], 0)
    assert padded.shape[0] >= 512 * 1024
    f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    a = f.filter_device(torch.from_numpy(below).cuda()).clone()
    b = f.filter_device(torch.from_numpy(padded).cuda()).clone()
    c = f.filter_device(torch.from_numpy(below).cuda()).clone()   # back on the small path, rows must have been left clean
    assert torch.equal(a, b) and torch.equal(a, c)
    f.close(); ctx.close()


def test_large_filter_point_outside_the_reused_box_and_growth(scene, cloud600k):
    """Speculative reuse of the previous grid definition on the large path: a later cloud with a point far outside the box (the
    summing kernel itself reports it; what it had summed by then is cleared with the rest) is rebuilt exactly, and a cloud whose box
    needs a larger bitmap grows it (byte map and rows with it) — each output equal to a fresh handle's."""
    import torch
    ctx = capi.Context(0)

    def fresh(pts):
        g = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
        o = g.filter_device(torch.from_numpy(np.ascontiguousarray(pts)).cuda()).clone()
        g.close()
        return o

    f = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
    base = cloud600k
    out0 = f.filter_device(torch.from_numpy(base).cuda()).clone()
    assert torch.equal(out0, fresh(base))
    moved = base.copy()
    moved[1234, :3] = [150.0, -90.0, 12.0]          # one point far outside the box the first build defined
    out1 = f.filter_device(torch.from_numpy(moved).cuda()).clone()
    assert torch.equal(out1, fresh(moved))
    assert f.counters()["misses"] >= 1
    out2 = f.filter_device(torch.from_numpy(base).cuda()).clone()   # and back: nothing of the miss may linger in rows or marks
    assert torch.equal(out2, out0)
    wide = base.copy()
    wide[:, :2] *= 3.0                               # nine times the footprint: the bitmap has to grow
    out3 = f.filter_device(torch.from_numpy(wide).cuda()).clone()
    assert torch.equal(out3, fresh(wide))
    out4 = f.filter_device(torch.from_numpy(base).cuda()).clone()
    assert torch.equal(out4, out0)
    f.close(); ctx.close()


def test_large_target_table_equals_the_oracle_pass(scene, oracle, cloud600k):
    """SetInputTarget of 600 k points (large path of the voxel table: byte marks, pack, scan, rank-keyed 10-sum rows) and one
    derivative pass of a filtered scan against it: score, gradient and Hessian against the oracle's on the same table inputs
    (1e-9 relative: the pass tolerance of tests/test_gpu_parity.py), pair counts exact."""
    import oracle_py
    ctx = capi.Context(0)
    tgt8 = synth.to_xyzi32(cloud600k)
    n = registration.NDTRegistrationHip(ctx)
    n.SetInputTarget(tgt8)
    o = oracle_py.OracleNDT()
    o.set_input_target(tgt8)
    oi, od = o.grid_dump()
    gi, gr = n.grid_dump()
    info = n.grid_info()
    occ, valid = o.grid_counts()
    assert info["occupied"] == occ == gi.shape[0] and info["valid"] == valid
    assert np.array_equal(gi[:, :4], oi[:, :4])                   # same cells, same order, same point counts
    assert np.array_equal(gi[:, 4] == 1, oi[:, 4] >= 6)
    assert np.max(np.abs(gr[:, :3] - od[:, :3])) < 1e-9           # centroids
    assert np.max(np.abs(gr[:, 9] - od[:, 12])) < 1e-9            # mean static weight
    poses = synth.trajectory(8)
    src = synth.gen_scan(scene, poses[2], seed=950)[::12]
    guess = (np.linalg.inv(poses[0]) @ poses[2]).astype(np.float32)
    _, T = n.ScanMatch(synth.to_xyzi32(src), guess)
    Tr, st_r = o.scan_match(synth.to_xyzi32(src), guess)
    assert n.last_stats["iterations"] == st_r["iterations"] and n.last_stats["passes"] == st_r["passes"]
    assert np.max(np.abs(T - Tr)) <= 1e-5
    n.close(); ctx.close()


def test_large_filter_randomized_against_the_oracle(oracle):
    """Clouds of random size around the large-path threshold (400 k - 900 k points), random extent, leaf, share of NaN rows and of
    duplicated points, in random order (no sensor structure: runs of equal cells are rare, tiles hold many distinct cells — the slot
    table overflows into direct adds), through ONE handle (path switches, box reuse and growth from build to build): voxel set, order
    and means against the oracle; a second pass over the same cloud must repeat the first bit for bit."""
    import torch
    rng = np.random.RandomState(77)
    ctx = capi.Context(0)
    handles = {}
    for case in range(8):
        n = int(rng.randint(400000, 900000))
        leaf = float(rng.choice([0.2, 0.3, 0.5, 1.0, 2.0]))
        ext = np.array([rng.uniform(20, 120), rng.uniform(20, 120), rng.uniform(2, 12)], dtype=np.float32)
        kind = case % 3
        if kind == 0:      # uniform volume: nearly every point its own cell at small leaves
            pts = (rng.uniform(-1, 1, (n, 3)) * ext).astype(np.float32)
        elif kind == 1:    # a few dense blobs + sparse background: hot cells
            c = (rng.uniform(-1, 1, (12, 3)) * ext).astype(np.float32)
            pts = (c[rng.randint(0, 12, n)] + rng.normal(0, 0.4, (n, 3))).astype(np.float32)
            pts[: n // 5] = (rng.uniform(-1, 1, (n // 5, 3)) * ext).astype(np.float32)
        else:              # planes (ground + walls), sensor-like sweep order kept
            t = np.sort(rng.uniform(0, 1, n)).astype(np.float32)
            pts = np.stack([np.cos(6.28 * 40 * t) * (5 + 60 * t), np.sin(6.28 * 40 * t) * (5 + 60 * t), rng.choice([-1.7, 0.5, 3.0], n) + rng.normal(0, 0.02, n)], 1).astype(np.float32)
        w = rng.uniform(0, 1, n).astype(np.float32)
        cloud = np.concatenate([pts, w[:, None]], 1)
        cloud[rng.rand(n) < 0.01, rng.randint(0, 3)] = np.nan
        dup = rng.randint(0, n, n // 50)
        cloud[rng.randint(0, n, n // 50)] = cloud[dup]
        f = handles.get(leaf)
        if f is None:
            f = handles[leaf] = cloud_filter.VoxelFilterHip(ctx, leaf, leaf, leaf)
        d = torch.from_numpy(np.ascontiguousarray(cloud)).cuda()
        a = f.filter_device(d).clone()
        b = f.filter_device(d).clone()
        assert torch.equal(a, b), case
        ref, rc = oracle.voxel_filter(synth.to_xyzi32(cloud), leaf)
        assert rc == 0 and ref.shape[0] == a.shape[0], (case, ref.shape, a.shape)
        got = a.cpu().numpy().astype(np.float64)
        err = np.abs(got - ref.astype(np.float64)) / np.maximum(np.abs(ref.astype(np.float64)), 1.0)
        assert err.max() < 4e-6, (case, kind, leaf, err.max())
    for f in handles.values():
        f.close()
    ctx.close()


def test_scan_filter_one_kernel_equals_mark_scan_accumulate(scene, oracle, monkeypatch):
    """A 131 k-point scan through VoxelFilter with the sums keyed by the cell (one kernel sums and sets the cells' bits, then scan and
    finalize: the default) and with LSA_VI_BY_CELL_SMALL=0 (mark, scan, accumulate by rank, finalize): the same bits on the device
    path, scan after scan through one handle (box reuse, a scan that leaves the box, NaN rows), and the oracle's voxel set."""
    import torch
    poses = synth.trajectory(6, step=1.5)
    scans = [synth.gen_scan(scene, poses[i], seed=4400 + i) for i in range(6)]
    scans[3] = scans[3].copy(); scans[3][100:140, 0] = np.nan
    scans[4] = scans[4].copy(); scans[4][7, :3] = [140.0, -95.0, 20.0]     # outside the box the earlier scans defined
    outs = {}
    for env in (None, "0"):
        if env is None:
            monkeypatch.delenv("LSA_VI_BY_CELL_SMALL", raising=False)
        else:
            monkeypatch.setenv("LSA_VI_BY_CELL_SMALL", env)
        ctx = capi.Context(0)
        f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
        outs[env] = [f.filter_device(torch.from_numpy(np.ascontiguousarray(s)).cuda()).clone().cpu().numpy() for s in scans]
        f.close(); ctx.close()
    monkeypatch.delenv("LSA_VI_BY_CELL_SMALL", raising=False)
    for a, b, s in zip(outs[None], outs["0"], scans):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        ref, rc = oracle.voxel_filter(synth.to_xyzi32(s), 0.5)
        assert rc == 0 and ref.shape[0] == a.shape[0]
        err = np.abs(a.astype(np.float64) - ref.astype(np.float64)) / np.maximum(np.abs(ref.astype(np.float64)), 1.0)
        assert err.max() < 4e-6, err.max()


def test_one_handle_through_every_build_path(scene, oracle, cloud600k):
    """ONE VoxelFilter handle driven through every build path in random order — host scans (first one copied, later ones pulled over the
    link by the summing kernel), device scans (one kernel: sums by cell + bits), a 600 k-point device cloud (byte marks + pack), a scan
    that leaves the reused cell box, an empty and an all-NaN cloud — each output against the oracle's voxel set and means: whatever a
    build leaves behind (rows, bits, byte marks, the reused grid definition) must not show in the next one."""
    import torch
    rng = np.random.RandomState(31)
    poses = synth.trajectory(5, step=1.3)
    scans = [synth.gen_scan(scene, poses[i], seed=4700 + i) for i in range(5)]
    far = scans[2].copy(); far[11, :3] = [-160.0, 120.0, 25.0]
    clouds = {"scan0": scans[0], "scan1": scans[1], "scan2": scans[2], "scan3": scans[3], "scan4": scans[4], "far": far, "big": cloud600k,
              "empty": np.zeros((0, 4), dtype=np.float32), "nan": np.full((70000, 4), np.nan, dtype=np.float32)}
    refs = {}
    for k, c in clouds.items():
        if c.shape[0] and np.isfinite(c[:, :3]).all(axis=1).any():
            refs[k] = oracle.voxel_filter(synth.to_xyzi32(c), 0.5)[0]
        else:
            refs[k] = np.zeros((0, 4), dtype=np.float32)
    ctx = capi.Context(0)
    f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    names = list(clouds)
    for step in range(40):
        k = names[rng.randint(len(names))]
        host = rng.rand() < 0.5 and k != "big"
        c = clouds[k]
        if host:
            _, out = f.Filter(synth.to_xyzi32(c))
            got = out[:, [0, 1, 2, 4]]
        else:
            got = f.filter_device(torch.from_numpy(np.ascontiguousarray(c)).cuda()).cpu().numpy()
        ref = refs[k]
        assert got.shape[0] == ref.shape[0], (step, k, host, got.shape, ref.shape)
        if ref.shape[0]:
            err = np.abs(got.astype(np.float64) - ref.astype(np.float64)) / np.maximum(np.abs(ref.astype(np.float64)), 1.0)
            assert err.max() < 4e-6, (step, k, host, err.max())
    c = f.counters()
    assert c["pulled"] >= 3 and c["misses"] >= 1
    f.close(); ctx.close()


@pytest.mark.parametrize("n_points", [120000, 600000])
def test_phase_stamps_change_nothing_and_are_ordered(scene, cloud600k, n_points):
    """lsa_voxel_filter_stamps (diagnostics, include/lidar_slam_amd.h): with the stamps on, a scan-sized and a large cloud give the
    bits they give without; every workgroup's record of the accumulate and the scan kernel is non-decreasing in time (the waits and
    the extra barrier that only exist in this mode sit where every wave of the workgroup reaches them); off again: same bits."""
    import ctypes as C
    import torch
    ctx = capi.Context(0)
    d = torch.from_numpy(np.ascontiguousarray(cloud600k[:n_points])).cuda()
    f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    plain = [f.filter_device(d).clone() for _ in range(2)]
    L = capi.lib()
    L.lsa_voxel_filter_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t)]
    per = C.c_size_t(0)
    ctx.check(L.lsa_voxel_filter_stamps(f.h, 1, None, 0, C.byref(per)))
    stamped = [f.filter_device(d).clone() for _ in range(2)]
    buf = (C.c_uint64 * (3 * per.value))()
    ctx.check(L.lsa_voxel_filter_stamps(f.h, 0, buf, 3 * per.value, C.byref(per)))
    ctx.check(L.lsa_voxel_filter_stamps(f.h, 0, None, 0, C.byref(per)))
    after = f.filter_device(d).clone()
    for o in plain[1:] + stamped + [after]:
        assert torch.equal(o, plain[0])
    a = np.frombuffer(buf, dtype=np.uint64).reshape(3, -1, 8).astype(np.int64)
    acc = a[1][a[1][:, 0] != 0]          # k_accum: entry .. flush acknowledged in columns 0..6 (7: a counter)
    assert len(acc) >= n_points // 2048 and (np.diff(acc[:, :7], axis=1) >= 0).all()
    scn = a[2][(a[2][:, 0] != 0) & (a[2][:, 7] != 0)]   # k_scan: workgroups that own a tile of the bitmap
    assert len(scn) >= 1 and (np.diff(scn, axis=1) >= 0).all()
    assert (acc[:, 6].max() - acc[:, 0].min()) < 100 * 1000   # < 1 ms at 100 MHz: the clock is the device's wall clock
    f.close(); ctx.close()
