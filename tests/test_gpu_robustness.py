"""GPU: the failure paths and contracts the round-4 review named — each forced on purpose and checked against the
ordinary path's bits.

* a pulled host-cloud build whose kernel gives up waiting for the packed rows (voxel_filter.hip: PullSource::patience_ticks):
  the asynchronous paths send the rows again and filter the whole cloud, the synchronous call fails loudly; nothing ever filters a
  half-delivered cloud;
* FrontEnd::Update on a device cloud (front_end.cpp:88: the caller owns the cloud again when Update returns): the scan buffer may
  be refilled by work on the context's stream at once, key-frame or not;
* NDT handle: small -> large -> small host uploads back to back (ndt_api.hip: upload_points) keep their bits;
* VoxelFilter with intensities the fixed-point scales do not cover (NaN, infinities, 16-bit reflectance): coordinates and counts
  untouched, no arbitrary bits in the intensity means.
"""
import numpy as np
import pytest

from lidar_slam_amd import capi, cloud_filter, front_end, registration, synth

pytestmark = pytest.mark.gpu


def test_pull_timeout_async_resends_sync_fails(scene, monkeypatch):
    """LSA_PULL_TEST="1,3000": the pulling kernel's patience is one 10 ns tick and every packing thread sleeps 3 ms before its first
    chunk, so every pulled build times out.  lsa_pipeline_scan_match_host (asynchronous host filters) must deliver the poses and
    filtered sizes of the undisturbed run bit for bit and count the time-outs; lsa_voxel_filter_apply must fail with LSA_ERR_HIP."""
    import torch
    N = 6
    poses = synth.trajectory(N, step=1.0)
    scans = [synth.gen_scan(scene, poses[i], seed=5200 + i) for i in range(N)]
    h = [synth.to_xyzi32(sc) for sc in scans]
    assert all(x.shape[0] >= 65536 for x in h)
    guesses = [(np.linalg.inv(poses[0]) @ poses[max(i - 1, 0)]).astype(np.float32) for i in range(N)]
    runs = {}
    for name, env in (("plain", None), ("timed out", "1,3000")):
        if env is None:
            monkeypatch.delenv("LSA_PULL_TEST", raising=False)
        else:
            monkeypatch.setenv("LSA_PULL_TEST", env)
        ctx, ca, cb = capi.Context(0), capi.Context(0), capi.Context(0)
        ndt = registration.NDTRegistrationHip(ctx)
        ndt.SetInputTarget(torch.from_numpy(np.ascontiguousarray(scans[0])).cuda())
        fa, fb = cloud_filter.VoxelFilterHip(ca, 0.5, 0.5, 0.5), cloud_filter.VoxelFilterHip(cb, 0.5, 0.5, 0.5)
        Ts, sts, nfs = registration.pipeline_scan_match_host([fa, fb], ndt, h[1:], guesses[1:])
        runs[name] = (np.stack(Ts), [(s["iterations"], s["passes"], s["score"]) for s in sts], list(nfs),
                      fa.pull_timeouts() + fb.pull_timeouts(), fa.counters()["pulled"] + fb.counters()["pulled"])
        if env is not None:
            # the synchronous host call: first call builds the ordinary way, the second is pulled -> times out -> error, not a result
            plain = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
            plain.Filter(h[0])
            with pytest.raises(capi.LsaError) as e:
                plain.Filter(h[1])
            assert e.value.code == capi.LSA_ERR_HIP and "in vain" in str(e.value)
            assert plain.pull_timeouts() == 1
            plain.close()
        fa.close(); fb.close(); ndt.close(); ca.close(); cb.close(); ctx.close()
    monkeypatch.delenv("LSA_PULL_TEST", raising=False)
    ref, got = runs["plain"], runs["timed out"]
    assert ref[3] == 0 and ref[4] >= 2
    assert got[4] >= 2 and got[3] == got[4], got   # every pulled build timed out and was re-sent
    assert np.array_equal(got[0], ref[0]) and got[1] == ref[1] and got[2] == ref[2]


@pytest.mark.parametrize("frame_filter", [1, 0])
def test_scan_buffer_refilled_on_the_context_stream_right_after_update(scene, frame_filter):
    """One preallocated device scan buffer, refilled by a copy enqueued on the context's stream immediately after every Update
    (key-frames included: the local map's update runs on its own stream after Update has returned) — poses, statistics and the
    incremental map must be those of a drive over untouched per-scan tensors.  With the frame filter the update reads the copy
    the filter's first kernel left in the handle's own buffer; without one the context's stream is ordered behind the read."""
    import torch
    N = 16
    poses = synth.trajectory(N, step=1.1)
    scans = [synth.gen_scan(scene, poses[i], seed=6400 + i, beams=64 if frame_filter else 32, az_steps=2083 if frame_filter else 900) for i in range(N)]
    d = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in scans]
    torch.cuda.synchronize()
    cfg = front_end.default_config()
    cfg.local_frame_num = 4
    cfg.use_frame_filter = frame_filter

    def flat(st):
        return (st["key_frame"], st["filtered_points"], st["local_map_points"], st["target_points"], st["ndt"]["iterations"], st["ndt"]["score"])

    ctx = capi.Context(0)
    fe = front_end.FrontEndHip(ctx, cfg)
    ref_T, ref_st = [], []
    for c in d:
        _, T = fe.Update(c)
        ref_T.append(T); ref_st.append(flat(fe.last_stats))
    ref_map = fe.map_verify()
    fe.close(); ctx.close()
    assert sum(s[0] for s in ref_st) >= 6 and ref_map["mismatches"] == 0

    ctx = capi.Context(0)
    fe = front_end.FrontEndHip(ctx, cfg)
    stream = torch.cuda.ExternalStream(ctx.stream)
    n_max = max(c.shape[0] for c in d)
    buf = torch.empty((n_max, 4), dtype=torch.float32, device="cuda")
    junk = torch.full((n_max, 4), 7.0e3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    got_T, got_st = [], []
    for c in d:
        n = c.shape[0]
        with torch.cuda.stream(stream):
            buf[:n].copy_(c, non_blocking=True)
        _, T = fe.Update(buf[:n])
        with torch.cuda.stream(stream):
            buf.copy_(junk, non_blocking=True)   # at once: nothing of this scan may be read from the buffer any more
        got_T.append(T); got_st.append(flat(fe.last_stats))
    got_map = fe.map_verify()
    fe.close(); ctx.close()
    assert np.array_equal(np.stack(got_T), np.stack(ref_T))
    assert got_st == ref_st
    assert got_map == ref_map


def test_small_large_small_host_uploads_back_to_back(scene):
    """lsa_ndt_set_input_target with host clouds of 7 k, 131 k and 7 k points in direct succession on one handle (the large upload goes
    through chunked kernels that read the staging buffer the third call repacks): each target must equal the one a fresh handle
    builds from the same cloud — same ScanMatch result, bit for bit."""
    poses = synth.trajectory(3, step=1.0)
    big = synth.gen_scan(scene, poses[0], seed=7100)
    small_a, small_b = big[::17].copy(), big[5::19].copy()
    src = synth.gen_scan(scene, poses[1], seed=7101)[::16].copy()
    assert big.shape[0] > 65536 and small_a.shape[0] < 65536
    guess = np.eye(4, dtype=np.float32)

    def match_fresh(target):
        ctx = capi.Context(0)
        ndt = registration.NDTRegistrationHip(ctx)
        ndt.SetInputTarget(synth.to_xyzi32(target))
        _, T = ndt.ScanMatch(synth.to_xyzi32(src), guess)
        st = dict(ndt.last_stats)
        ndt.close(); ctx.close()
        return T, (st["iterations"], st["passes"], st["score"])

    want = [match_fresh(t) for t in (small_a, big, small_b)]
    ctx = capi.Context(0)
    ndt = registration.NDTRegistrationHip(ctx)
    for rep in range(20):
        # the three uploads with nothing in between, then the check of the LAST one; then the first two, checked after the large one
        ndt.SetInputTarget(synth.to_xyzi32(small_a))
        ndt.SetInputTarget(synth.to_xyzi32(big))
        ndt.SetInputTarget(synth.to_xyzi32(small_b))
        _, T = ndt.ScanMatch(synth.to_xyzi32(src), guess)
        st = ndt.last_stats
        assert np.array_equal(T, want[2][0]) and (st["iterations"], st["passes"], st["score"]) == want[2][1], rep
        ndt.SetInputTarget(synth.to_xyzi32(small_a))
        ndt.SetInputTarget(synth.to_xyzi32(big))
        _, T = ndt.ScanMatch(synth.to_xyzi32(src), guess)
        st = ndt.last_stats
        assert np.array_equal(T, want[1][0]) and (st["iterations"], st["passes"], st["score"]) == want[1][1], rep
    ndt.close(); ctx.close()


def test_filter_intensities_outside_the_fixed_point_range(scene, oracle):
    """Intensities the scales do not cover (|w| >= 1024 breaks the magic-number rounding at kw = 40): 16-bit reflectance, NaN, +-inf.
    xyz means, count and order must equal the oracle's on the same cloud with clean intensities, bit for bit; the intensity mean of
    a voxel holding only finite values must match the oracle's f32 mean (sparse voxels: no int64 overflow); a non-finite intensity
    counts as 0 (stated deviation from PCL, which would hand a NaN mean to the registration as a static weight) — never
    arbitrary bits; the device path and the incremental front end's sums see the same rule."""
    import torch
    rng = np.random.default_rng(99)
    base = synth.gen_scan(scene, synth.trajectory(1)[0], seed=8100)[::6].copy()   # sparse: few points per 0.5 m voxel
    n = base.shape[0]
    w = rng.uniform(0.0, 1.0, n).astype(np.float32)
    big = rng.random(n) < 0.2
    w[big] = rng.integers(2048, 65536, big.sum()).astype(np.float32)
    clean = base.copy(); clean[:, 3] = w
    dirty = clean.copy()
    bad = rng.random(n) < 0.05
    kinds = rng.integers(0, 3, n)
    dirty[bad & (kinds == 0), 3] = np.nan
    dirty[bad & (kinds == 1), 3] = np.inf
    dirty[bad & (kinds == 2), 3] = -np.inf
    zeroed = clean.copy(); zeroed[bad, 3] = 0.0
    ctx = capi.Context(0)
    f = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    _, out_dirty = f.Filter(synth.to_xyzi32(dirty))
    _, out_zeroed = f.Filter(synth.to_xyzi32(zeroed))
    dev = f.filter_device(torch.from_numpy(np.ascontiguousarray(dirty)).cuda()).cpu().numpy()
    ref, _ = oracle.voxel_filter(synth.to_xyzi32(zeroed), 0.5)
    f.close(); ctx.close()
    assert out_dirty.shape == out_zeroed.shape and out_dirty.shape[0] == ref.shape[0]
    # non-finite == 0, on both paths, every bit
    assert np.array_equal(out_dirty.view(np.uint32), out_zeroed.view(np.uint32))
    assert np.array_equal(dev.view(np.uint32), np.ascontiguousarray(out_dirty[:, [0, 1, 2, 4]]).view(np.uint32))
    # coordinates and intensity means as the oracle has them (the bar of test_voxel_filter_matches_oracle)
    got = out_dirty[:, [0, 1, 2, 4]].astype(np.float64)
    assert np.all(np.isfinite(got))
    err = np.abs(got - ref.astype(np.float64)) / np.maximum(np.abs(ref.astype(np.float64)), 1.0)
    assert err.max() < 4e-6, err.max()


def test_single_update_defers_its_final_pass_without_changing_a_bit(scene, monkeypatch):
    """FrontEnd::Update (one scan per call) posts its ScanMatch's final derivative pass and takes its rows only after the key-frame's
    first launches (front_end.hip: fe_update_filtered / fe_rebuild_map).  Three drives over the same 26 scans — local map switching to
    the filtered form at the tenth key-frame — must agree in every pose, every statistic the final pass completes (score, probability,
    pairs per pass) and the map: (a) LSA_FE_DEFER=0, the pass waited for first; (b) the default; (c) the default with an armed patience
    of a few microseconds, so that armed kernels — deferred ones among them — leave before their mail comes and their passes are redone
    the ordinary way BETWEEN the two halves of the map update, where the redo must still find the table the pass was posted for."""
    import ctypes as C
    import torch
    N = 26
    poses = synth.trajectory(N, step=1.1)
    d = [torch.from_numpy(np.ascontiguousarray(synth.gen_scan(scene, poses[i], seed=7100 + i))).cuda() for i in range(N)]
    torch.cuda.synchronize()

    def flat(st):
        nd = st["ndt"]
        return (st["key_frame"], st["filtered_points"], st["local_map_points"], st["target_points"], nd["iterations"], nd["passes"], nd["score"],
                nd["trans_probability"], nd["pairs_per_pass"], nd["converged"])

    def drive(defer, patience_us):
        monkeypatch.setenv("LSA_FE_DEFER", "1" if defer else "0")
        ctx = capi.Context(0)
        fe = front_end.FrontEndHip(ctx)
        L = capi.lib()
        L.lsa_front_end_registration.restype = C.c_void_p
        L.lsa_front_end_registration.argtypes = [C.c_void_p]
        ndt = C.c_void_p(L.lsa_front_end_registration(fe.h))
        if patience_us is not None:
            L.lsa_ndt_set_armed_patience.argtypes = [C.c_void_p, C.c_double]
            ctx.check(L.lsa_ndt_set_armed_patience(ndt, float(patience_us)))
        Ts, sts = [], []
        for c in d:
            _, T = fe.Update(c)
            Ts.append(T); sts.append(flat(fe.last_stats))
        st = capi.NdtArmedStats()
        L.lsa_ndt_armed_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        ctx.check(L.lsa_ndt_armed_stats(ndt, C.byref(st), 0))
        m = fe.map_verify()
        fe.close(); ctx.close()
        return np.stack(Ts), sts, m, st.as_dict()

    T0, s0, m0, a0 = drive(False, None)
    T1, s1, m1, a1 = drive(True, None)
    stalled = [drive(True, p) for p in (3.0, 6.0, 10.0)]   # around the link's round trip: some mails come in time, some do not
    monkeypatch.delenv("LSA_FE_DEFER", raising=False)
    assert sum(s[0] for s in s0) >= 12 and m0["mismatches"] == 0          # the filtered map is reached and verified against a full build
    assert a0["deferred"] == 0 and a1["deferred"] >= N // 2, (a0, a1)       # (b) really deferred final passes
    lost = sum(a["fallbacks"] + a["expired"] for _, _, _, a in stalled)
    print("deferred %d of %d ScanMatches; stalled drives: %s" % (a1["deferred"], N - 1, [(a["deferred"], a["fallbacks"], a["expired"]) for _, _, _, a in stalled]))
    assert lost >= 5, [a for _, _, _, a in stalled]                        # (c) really lost armed kernels
    for T, s, m, _ in [(T1, s1, m1, a1)] + stalled:
        assert np.array_equal(T, T0)
        assert s == s0
        assert m == m0
