"""GPU: the armed passes never leave the caller's stream stalled, and nothing about a slow or lost pass is silent.

Round 2's driver run showed two stalls: 50 plain ScanMatch calls after a batch took 0.84 ms each instead of
0.05, and every lsa_pipeline_scan_match batch ended with 0.3 ms of a parked kernel's patience inside the caller's timed
region.  These tests drive exactly those sequences and assert on the library's COUNTERS (lsa_ndt_armed_stats) — wall-clock
bounds are generous (a 20 ms row time-out is what they would catch), the counters are exact:

* a batch returns with nothing parked; the carry is opt-in for the plain calls;
* batch -> idle longer than the patience -> 50 standalone ScanMatches: no late rows, no redone passes;
* an armed kernel whose patience is about as long as the host's gap between two passes (so that the mail keeps landing
  inside the skew of the workgroups' patience clocks: some workgroups take it, some have left) gives the bits of the
  unarmed path, and every workgroup that leaves says so — the host never sits out a row limit;
* a source buffer refilled on ANOTHER stream, ordered by an event only, while a kernel is parked.
"""
import gc
import time

import numpy as np
import pytest

from lidar_slam_amd import capi, cloud_filter, registration
from lidar_slam_amd import workloads as wl

pytestmark = pytest.mark.gpu


ROW_LIMIT_US = 2500.0    # the library's own bound on an armed pass's rows (ndt_api.hip: kArmedRowLimitMs = 2 ms) + slack
HOST_STALL_US = 1000.0   # an interval this long between two looks of the calling thread at its rows is the OS, not the library


def _quiet(s, allow=2):
    """The outcomes that cost time (late rows, passes redone the ordinary way, kernels seen giving up) are COUNTED, never silent;
    what these tests were written against showed as dozens per run (one per batch, 123 per 22 k scans).  NONE is accepted unless the
    library's host-stall detector (armed_stats: host_gap_max_us = the longest interval between two consecutive looks of the calling
    thread at the rows it waited for, or between two passes of one ScanMatch) shows that the OS held the thread up for a millisecond
    or more — then up to `allow` pass, every result still identical.  Both numbers are printed."""
    n = s["late"] + s["rows_late"] + s["fallbacks"] + s["gave_up_seen"]
    stalled = s["host_gap_max_us"] >= HOST_STALL_US
    if n or stalled:
        print("armed outcomes other than the plain one: %d; host gap max %.1f us (%s)" % (n, s["host_gap_max_us"], "thread held up" if stalled else "no stall seen"))
    return n == 0 or (stalled and n <= allow)


def _rows_in_time(s):
    """No row waited for beyond the library's own limit — unless the detector saw the thread itself held up that long.  The limit is
    per workgroup ROW (row_wait_max_us); a whole pass (rows_wait_max_us: its rows taken one after the other) normally stays below it
    too, but a device that delivers slowly for a few milliseconds — seen once per ~100 k passes on the shared test boxes, with the host
    spinning all the while (host gap < 0.1 ms) and nothing late — is not the library's doing: it is printed, not failed."""
    if s["rows_wait_max_us"] >= ROW_LIMIT_US:
        print("a pass's rows took %.0f us in all (longest single row %.0f us, host gap max %.1f us)" % (s["rows_wait_max_us"], s["row_wait_max_us"], s["host_gap_max_us"]))
    return s["row_wait_max_us"] < ROW_LIMIT_US or s["host_gap_max_us"] >= HOST_STALL_US


def _armed_all(s, total):
    """every derivative pass ran as an armed pass — unless one was redone, after which the rest of that ScanMatch runs unarmed"""
    return s["passes"] == total if s["fallbacks"] == 0 else s["passes"] <= total


@pytest.fixture(scope="module")
def rig():
    import torch
    ctx, ctx_f = capi.Context(0), capi.Context(0)
    poses, scans = wl.cfg2_dataset(workers=1)
    map_filter = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
    frame_filter = cloud_filter.VoxelFilterHip(ctx_f, 0.5, 0.5, 0.5)
    d_map_f = map_filter.filter_device(torch.from_numpy(wl.local_map(poses, scans)).cuda()).clone()
    d_raw = [torch.from_numpy(np.ascontiguousarray(scans[k])).cuda() for k in wl.TEST_SCANS]
    d_src = [frame_filter.filter_device(r).clone() for r in d_raw]
    torch.cuda.synchronize()
    map_filter.close()
    cap = max(r.shape[0] for r in d_raw)
    bufs = [torch.empty((cap, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    r = dict(ctx=ctx, ctx_f=ctx_f, ff=frame_filter, d_map_f=d_map_f, d_raw=d_raw, d_src=d_src, bufs=bufs,
             guesses=[wl.guess(poses, k) for k in wl.TEST_SCANS])
    yield r
    frame_filter.close(); ctx_f.close(); ctx.close()


def _ndt(rig, **kw):
    n = registration.NDTRegistrationHip(rig["ctx"], **wl.NDT_PARAMS)
    n.SetInputTarget(rig["d_map_f"])
    return n


def _reference(rig):
    """poses / statistics of the nine bench scans with the armed passes OFF (one ordinary launch per pass)"""
    n = _ndt(rig)
    n.set_armed(False)
    ref = []
    for d_s, g in zip(rig["d_src"], rig["guesses"]):
        _, T = n.ScanMatch(d_s, g)
        ref.append((T, n.last_stats["passes"], n.last_stats["iterations"]))
    n.close()
    return ref


def test_nothing_parked_by_default_and_batch_cancels_behind_its_last_scan(rig):
    n = _ndt(rig)
    ref = _reference(rig)
    # plain ScanMatch: armed passes inside the call, the kernel of the pass that never came is cancelled at its end
    for k in range(3):
        _, T = n.ScanMatch(rig["d_src"][k], rig["guesses"][k])
        s = n.armed_stats()
        assert np.array_equal(T, ref[k][0]) and s["parked"] == 0
    s = n.armed_stats()
    assert _armed_all(s, sum(r[1] for r in ref[:3])) and s["cancels"] >= 3 and _quiet(s, 1), s
    # the batch carries between its scans (one cancel for the whole batch, behind the last scan) and returns with nothing parked
    n.armed_stats(reset=True)
    idx = [i % 9 for i in range(27)]
    Ts, sts, nfs = registration.pipeline_scan_match(rig["ff"], n, [rig["d_raw"][i] for i in idx], [rig["guesses"][i] for i in idx], rig["bufs"])
    s = n.armed_stats()
    assert s["parked"] == 0 and _quiet(s, 1), s
    assert s["cancels"] <= 1 + s["expired"] + s["fallbacks"] and _armed_all(s, sum(st["passes"] for st in sts)), s
    for i, T in zip(idx, Ts):
        assert np.array_equal(T, ref[i][0])
    # ... so a wait of the caller's own on the stream right behind the batch costs microseconds, not the kernel's patience
    import torch
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("device-wide wait right behind a batch: %.1f us" % (dt * 1e6))
    assert dt < 5e-3, dt          # (the counter above is the assertion; a parked kernel would show as >= 300 us here)
    # opting in: the kernel stays parked, any other call on the context (here the library's own synchronise) releases it
    n.set_armed_carry(True)
    n.ScanMatch(rig["d_src"][0], rig["guesses"][0])
    assert n.armed_stats()["parked"] == 1
    rig["ctx"].synchronize()
    assert n.armed_stats()["parked"] == 0
    # a batch on a handle whose owner opted in leaves its last kernel parked, as asked
    registration.pipeline_scan_match(rig["ff"], n, rig["d_raw"][:2], rig["guesses"][:2], rig["bufs"])
    assert n.armed_stats()["parked"] == 1
    n.set_armed_carry(False)
    assert n.armed_stats()["parked"] == 0
    n.close()


def test_deferred_final_pass_completes_every_statistic(rig):
    """Inside a batch the last derivative pass of a ScanMatch is posted but not waited for (the pose does not depend on it);
    its sums complete the scan's statistics behind the next scan's first mail.  A batch long enough to pass the library's
    periodic stream wait (which cancels the parked kernel and makes the sequence numbers skip one: the deferred pass and the
    next pass would then share a mailbox slot and a row-buffer half) must return, scan for scan, the poses AND statistics of
    plain sequential calls; no pass may be lost or redone."""
    n = _ndt(rig)
    plain = _ndt(rig)
    plain.set_armed(False)
    ref = []
    for d_s, g in zip(rig["d_src"], rig["guesses"]):
        _, T = plain.ScanMatch(d_s, g)
        ref.append((T, dict(plain.last_stats)))
    plain.close()
    idx = [i % 9 for i in range(700)]
    Ts, sts, nfs = registration.pipeline_scan_match(rig["ff"], n, [rig["d_raw"][i] for i in idx], [rig["guesses"][i] for i in idx], rig["bufs"])
    s = n.armed_stats()
    assert s["deferred"] >= 650 and _quiet(s, 3) and s["parked"] == 0, s
    assert s["cancels"] >= 3          # the periodic stream wait did occur (every 512 armed passes)
    for i, T, st in zip(idx, Ts, sts):
        Tr, sr = ref[i]
        assert np.array_equal(T, Tr)
        for key in ("iterations", "passes", "ls_iterations", "converged", "score", "trans_probability", "pairs_per_pass"):
            assert st[key] == sr[key], (i, key, st[key], sr[key])
        assert np.array_equal(st["final_p"], sr["final_p"])
    n.close()


@pytest.mark.parametrize("carry", [False, True], ids=["default", "carry_opt_in"])
def test_batch_idle_then_standalone_scan_matches(rig, carry):
    """The driver's sequence: batches, an idle gap longer than the patience, then 50 plain ScanMatch calls."""
    ref = _reference(rig)
    n = _ndt(rig)
    n.set_armed_carry(carry)
    idx = [i % 9 for i in range(45)]
    for _ in range(2):
        registration.pipeline_scan_match(rig["ff"], n, [rig["d_raw"][i] for i in idx], [rig["guesses"][i] for i in idx], rig["bufs"])
        time.sleep(0.002)             # > patience: with the carry a parked kernel expires here
    n.armed_stats(reset=True)
    ts = []
    gc.collect(); gc.disable()
    try:
        for i in range(50):
            t0 = time.perf_counter()
            _, T = n.ScanMatch(rig["d_src"][i % 9], rig["guesses"][i % 9])
            ts.append(time.perf_counter() - t0)
            assert np.array_equal(T, ref[i % 9][0])
    finally:
        gc.enable()
    s = n.armed_stats()
    ts = np.array(ts) * 1e6
    print("carry %s: ScanMatch min / median / p90 / max = %.1f / %.1f / %.1f / %.1f us; %s" % (carry, ts.min(), np.median(ts), np.percentile(ts, 90), ts.max(), s))
    assert _quiet(s), s
    assert _rows_in_time(s), s
    assert _armed_all(s, sum(ref[i % 9][1] for i in range(50))), s
    # only the kernel the last batch left parked (opt-in) had expired (with the carry on, Python's own time between two calls counts
    # against the parked kernel's 300 us: a collection or a held-up thread there costs the odd one more; without it, none can)
    assert (1 if carry else 0) <= s["expired"] <= (3 if carry else 0), s
    assert np.percentile(ts, 90) < 1000.0 and ts.max() < 5000.0, ts
    n.close()


def test_patience_as_short_as_the_host_gap(rig):
    """Patience of a few microseconds: armed kernels keep running out of it while the host is between two passes, with the
    mail landing before, inside and after the skew of the workgroups' clocks.  Whatever happens per pass — served armed,
    found expired and relaunched, partially run and redone — the poses are the unarmed path's, every redone pass is counted,
    and the host never waits out a row limit (a workgroup that leaves reports it)."""
    ref = _reference(rig)
    seen = dict(fallbacks=0, expired=0, gave_up_seen=0)
    for patience in (2.0, 4.0, 6.0, 8.0, 12.0, 20.0):
        n = _ndt(rig)
        n.set_armed_carry(True)
        n.set_armed_patience(patience)
        t0 = time.perf_counter()
        for rep in range(4):
            for k in range(9):
                _, T = n.ScanMatch(rig["d_src"][k], rig["guesses"][k])
                assert np.array_equal(T, ref[k][0]), (patience, rep, k)
                assert n.last_stats["passes"] == ref[k][1]
        dt = time.perf_counter() - t0
        s = n.armed_stats()
        print("patience %4.1f us: %.1f us per ScanMatch, %s" % (patience, dt / 36 * 1e6, s))
        # no row limit was sat out: a workgroup that leaves says so and the wait ends at once (a thread the box holds up shows in the detector)
        assert s["late"] + s["rows_late"] == 0 or (s["host_gap_max_us"] >= HOST_STALL_US and s["late"] + s["rows_late"] <= 1), s
        assert _rows_in_time(s), s
        assert dt / 36 < 2e-3, (patience, dt)
        for key in seen:
            seen[key] += s[key]
        n.close()
    assert seen["expired"] + seen["fallbacks"] > 0, seen           # the give-up paths really ran
    with pytest.raises(capi.LsaError):
        n = _ndt(rig)
        try:
            n.set_armed_patience(0.0)
        finally:
            n.close()


def test_source_buffer_refilled_on_another_stream_while_a_kernel_is_parked(rig):
    """A parked kernel has passed nothing but its start when the caller rewrites the source buffer on a SECOND stream and
    orders the ScanMatch behind it by an event only (no wait of the NDT's stream, which would sit out the parked kernel):
    the kernel reads source, size and grid from its mail and the points behind it, so it must see the refilled buffer."""
    import torch
    n = _ndt(rig)
    n.set_armed_carry(True)
    # (what the caller does between two ScanMatches here — a copy on a fresh stream, a whole unarmed ScanMatch on another context —
    # is not bounded by the default 0.3 ms patience on a slow box: seen once, the kernel had left and the first pass was redone the
    # ordinary way, correctly counted.  This test is about WHAT a parked kernel reads, so it gets the patience to stay parked.)
    n.set_armed_patience(20000.0)
    # the comparison handle lives on a context of its OWN: any call on n's context would release n's parked kernel first
    ctx2 = capi.Context(0)
    plain = registration.NDTRegistrationHip(ctx2, **wl.NDT_PARAMS)
    plain.SetInputTarget(rig["d_map_f"])
    plain.set_armed(False)
    m = min(s.shape[0] for s in rig["d_src"][:4])
    buf = rig["d_src"][0][:m].clone()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    n.ScanMatch(buf, rig["guesses"][0])          # leaves a kernel parked
    n.armed_stats(reset=True)
    used = 0
    for k in (1, 2, 3, 0, 1, 2, 3, 0):
        parked = n.armed_stats()["parked"]
        with torch.cuda.stream(side):
            buf.copy_(rig["d_src"][k][:m], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        # the copy must be complete before the mail names the buffer: the host waits for the EVENT (not for the NDT's
        # stream); the parked kernel, already resident, then reads the new points
        ev.synchronize()
        _, Ta = n.ScanMatch(buf, rig["guesses"][k])
        _, Tb = plain.ScanMatch(rig["d_src"][k][:m].contiguous(), rig["guesses"][k])
        assert np.array_equal(Ta, Tb), k
        used += parked
    s = n.armed_stats()
    print("refill on a second stream: %d of 8 ScanMatches found their kernel parked; %s" % (used, s))
    assert _quiet(s, 1), s
    assert used >= 4 or s["expired"] >= 4      # the parked kernel was what ran (or had honestly expired on a slow box)
    plain.close(); n.close(); ctx2.close()


def test_host_upload_with_bad_intensity_offset_on_the_split_path(rig):
    """>= 65536 rows take the two-thread upload of lsa_voxel_filter_apply: a row layout that cannot hold the intensity is
    refused up front with LSA_ERR_INVALID, before either thread packs anything, and the handle stays usable."""
    import ctypes as C
    f = cloud_filter.VoxelFilterHip(rig["ctx"], 0.5, 0.5, 0.5)
    pts = np.random.default_rng(5).uniform(-30, 30, (70000, 4)).astype(np.float32)
    out = np.zeros((70000, 8), dtype=np.float32)
    n_out = C.c_size_t(0)
    L = capi.lib()
    for bad_off in (13, 16, 1000):
        rc = L.lsa_voxel_filter_apply(f.h, pts.ctypes.data, pts.shape[0], 16, bad_off, out.ctypes.data, out.shape[0], 32, 16, C.byref(n_out))
        assert rc == capi.LSA_ERR_INVALID and b"intensity" in L.lsa_last_error(rig["ctx"].h)
    rc = L.lsa_voxel_filter_apply(f.h, pts.ctypes.data, pts.shape[0], 16, 12, out.ctypes.data, out.shape[0], 32, 16, C.byref(n_out))
    assert rc == 0 and 0 < n_out.value <= 70000
    f.close()
