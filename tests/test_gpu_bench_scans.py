"""GPU: the bench workloads themselves under pytest (VERDICT r1 items 1, 2 and `configs_untested`).

* cfg2 (BASELINE.json configs[1], the headline): the nine bench scans at FULL size against the oracle, both NDT plugins
  ("NDTMANUAL" faithful mode and the PCL-semantics "NDT" mode), through the C-ABI.
    - on IDENTICAL voxel tables (the oracle's cells overwritten with the GPU's records) every derivative pass of the Newton
      iteration must agree: same iteration / pass counts and per-pass pose within 1e-9 — for every scan, including the ones
      the PCL-semantics iteration does not settle on.  What is left between the two sides there is the order of the f64
      sums only, and that is NOT amplified: the pose reaches the kernel through an f32 matrix, which absorbs 1e-15.
    - on each side's OWN table the faithful mode must land within 1e-4 m / 1e-5 rad with the same counts.
      The PCL-semantics mode is held to that bar on the scans where its iteration settles; where it does not (a
      non-contractive iteration: with PCL's zero-seeded covariances the 0.3 m-filtered map gives voxels 3 cm thin, an
      8 cm prediction error starts outside the Newton basin, the Hessian is indefinite along the path and every step is
      the clamped 0.1) the 1e-7 relative round-off gap between the two covariance formulas (the oracle restates PCL's
      cancellation-prone single-pass expression, the GPU evaluates the shift-invariant one) is amplified — the test pins
      that cause: the first Newton step already differs by that round-off, not by 1e-12, and with identical tables the
      very same scans agree to 1e-9.
* cfg4 (BASELINE.json configs[3]): the ~132 k-point raw scan against the 2 M-point submap — pair count, bitwise
  repeatability, additivity over the 8-way shard split, and ONE oracle derivative pass on a 1/16 subsample of the source
  against the full 2 M-point target.
"""
import numpy as np
import pytest

from lidar_slam_amd import capi, cloud_filter, registration, synth
from lidar_slam_amd import workloads as wl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def cfg2(ctx):
    """The bench's cfg2 inputs: local map filtered at 0.3 m and the nine frame-filtered test scans (GPU filters, as in
    bench.py), handed to both sides as identical host arrays."""
    import torch
    poses, scans = wl.cfg2_dataset(workers=1)
    map_filter = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
    frame_filter = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    d_map = torch.from_numpy(wl.local_map(poses, scans)).cuda()
    map_f = map_filter.filter_device(d_map).cpu().numpy().copy()
    srcs = [frame_filter.filter_device(torch.from_numpy(np.ascontiguousarray(scans[k])).cuda()).cpu().numpy().copy() for k in wl.TEST_SCANS]
    guesses = [wl.guess(poses, k) for k in wl.TEST_SCANS]
    map_filter.close(); frame_filter.close()
    return dict(poses=poses, scans=scans, map_f=map_f, srcs=srcs, guesses=guesses)


def _trace_gap(a, b):
    n = min(len(a), len(b))
    return np.abs(a[:n, :6] - b[:n, :6]).max(axis=1) if n else np.zeros(0)


@pytest.mark.parametrize("pcl", [False, True], ids=["ndtmanual", "pcl_semantics"])
def test_cfg2_bench_scans_identical_tables(ctx, oracle, cfg2, pcl):
    import oracle_py
    tgt8 = synth.to_xyzi32(cfg2["map_f"])
    n = registration.NDTRegistrationHip(ctx, pcl_mode=pcl, **wl.NDT_PARAMS)
    n.SetInputTarget(tgt8)
    gi, gr = n.grid_dump()
    o = oracle_py.OracleNDT(pcl_mode=pcl, **wl.NDT_PARAMS)
    o.set_input_target(tgt8)
    oi, _ = o.grid_dump()
    assert np.array_equal(gi[:, :4], oi[:, :4])          # same cells, same order, same point counts
    assert np.array_equal(gi[:, 4] == 1, oi[:, 4] >= 6)  # same neighbour set
    o.override_cells(gi, gr)
    worst = 0.0
    for src, guess in zip(cfg2["srcs"], cfg2["guesses"]):
        s8 = synth.to_xyzi32(src)
        Tr, st_r = o.scan_match(s8, guess, trace=True)
        _, T = n.ScanMatch(s8, guess, trace=True)
        st = n.last_stats
        assert (st["iterations"], st["passes"], st["ls_iterations"], st["converged"]) == \
               (st_r["iterations"], st_r["passes"], st_r["ls_iterations"], st_r["converged"])
        assert len(st["trace"]) == len(st_r["trace"]) == st["passes"]
        gap = _trace_gap(st["trace"], st_r["trace"])
        worst = max(worst, float(gap.max()))
        assert gap.max() <= 1e-9, gap                                            # every pass, capped scans included
        assert np.array_equal(st["trace"][:, 8], st_r["trace"][:, 8])            # same pass kinds in the same order
        assert np.max(np.abs(st["trace"][:, 7] - st_r["trace"][:, 7])) <= 1e-9   # same step lengths
        dt, dr = wl.pose_err(T, Tr)
        assert dt <= 1e-6 and dr <= 1e-7, (dt, dr)
    print("identical tables, %s: worst per-pass pose gap %.2e" % ("pcl" if pcl else "manual", worst))
    n.close()


def test_cfg2_bench_scans_match_oracle(ctx, oracle, cfg2):
    """Faithful mode, each side on its own voxel table: the SURVEY.md 8d bar on the headline's own scans at full size."""
    import oracle_py
    tgt8 = synth.to_xyzi32(cfg2["map_f"])
    n = registration.NDTRegistrationHip(ctx, **wl.NDT_PARAMS)
    n.SetInputTarget(tgt8)
    o = oracle_py.OracleNDT(**wl.NDT_PARAMS)
    o.set_input_target(tgt8)
    errs = []
    for k, src, guess in zip(wl.TEST_SCANS, cfg2["srcs"], cfg2["guesses"]):
        s8 = synth.to_xyzi32(src)
        # the frame filter itself: same voxel count as the oracle's on the raw scan (values are covered by test_gpu_parity)
        ref_f, _ = oracle.voxel_filter(synth.to_xyzi32(cfg2["scans"][k]), 0.5)
        assert ref_f.shape[0] == src.shape[0]
        Tr, st_r = o.scan_match(s8, guess)
        _, T = n.ScanMatch(s8, guess)
        st = n.last_stats
        assert (st["iterations"], st["passes"], st["converged"]) == (st_r["iterations"], st_r["passes"], st_r["converged"])
        assert st["iterations"] <= wl.NDT_PARAMS["max_iter"] and st["converged"] == 1
        dt, dr = wl.pose_err(T, Tr)
        errs.append((dt, dr))
        assert dt <= 1e-4 and dr <= 1e-5, (k, dt, dr)
        assert abs(st["pairs_per_pass"] - st_r["pairs_per_pass"]) <= 1e-9 * st_r["pairs_per_pass"]
        # and it is a registration: within 5 cm of the true pose from an 8 cm / 0.5 deg prediction error
        assert wl.pose_err(T, wl.truth(cfg2["poses"], k))[0] < 0.05
    e = np.array(errs)
    print("cfg2 faithful mode vs oracle: RMSE %.2e m / %.2e rad" % (np.sqrt(np.mean(e[:, 0] ** 2)), np.sqrt(np.mean(e[:, 1] ** 2))))
    n.close()


def test_cfg2_pcl_mode_own_tables_cause_of_the_gap(ctx, oracle, cfg2):
    """PCL-semantics mode, own tables (the comparison BENCH_r01's pose_rmse_pcl_mode_vs_oracle made).  Settled scans meet
    the 1e-4 m / 1e-5 rad bar; on the others the gap is there from the FIRST Newton step at the size of the covariance
    round-off (>= 1e-12, <= 1e-5) and grows along a non-contractive iteration — see the module docstring."""
    import oracle_py
    tgt8 = synth.to_xyzi32(cfg2["map_f"])
    n = registration.NDTRegistrationHip(ctx, pcl_mode=True, **wl.NDT_PARAMS)
    n.SetInputTarget(tgt8)
    o = oracle_py.OracleNDT(pcl_mode=True, **wl.NDT_PARAMS)
    o.set_input_target(tgt8)
    # the voxels of this map under PCL's statistics: thin (the eigenvalue floor is 1 % of the largest eigenvalue)
    gi, gr = n.grid_dump()
    ic = gr[gi[:, 4] == 1][:, 3:9]
    M = np.stack([np.stack([ic[:, 0], ic[:, 1], ic[:, 2]], 1), np.stack([ic[:, 1], ic[:, 3], ic[:, 4]], 1), np.stack([ic[:, 2], ic[:, 4], ic[:, 5]], 1)], 1)
    lam = np.linalg.eigvalsh(M)
    lam = lam[lam[:, 2] > 0]
    sigma_min = float(np.median(1.0 / np.sqrt(lam[:, 2])))
    assert sigma_min < 0.05          # ~3 cm: an 8 cm prediction error is well outside one sigma
    settled, unsettled, first_step_gap = 0, 0, []
    for src, guess in zip(cfg2["srcs"], cfg2["guesses"]):
        s8 = synth.to_xyzi32(src)
        Tr, st_r = o.scan_match(s8, guess, trace=True)
        _, T = n.ScanMatch(s8, guess, trace=True)
        st = n.last_stats
        gap = _trace_gap(st["trace"], st_r["trace"])
        assert gap[0] == 0.0                       # the pass at the guess: same pose on both sides
        if len(gap) > 1:
            first_step_gap.append(float(gap[1]))
            assert gap[1] <= 1e-5, gap[:4]         # table round-off through one Newton solve, not a different algorithm
        cap = wl.NDT_PARAMS["max_iter"]
        same_path = st["iterations"] == st_r["iterations"] and st["iterations"] <= cap and float(gap.max()) < 1e-5
        if same_path:
            settled += 1
            dt, dr = wl.pose_err(T, Tr)
            assert dt <= 1e-4 and dr <= 1e-5, (dt, dr)
        else:
            unsettled += 1
            # both sides are still doing the same thing: every step is a clamped Newton step in [eps / 2, step_size]
            lo, hi = float(np.float32(wl.NDT_PARAMS["trans_eps"])) / 2, float(np.float32(wl.NDT_PARAMS["step_size"]))  # the plugin holds floats
            for tr in (st["trace"], st_r["trace"]):
                assert np.all(tr[1:, 7] >= lo) and np.all(tr[1:, 7] <= hi)
    print("pcl semantics, own tables: %d settled / %d not; pose gap after the first Newton step: median %.1e, max %.1e; sigma_min %.3f m"
          % (settled, unsettled, np.median(first_step_gap), np.max(first_step_gap), sigma_min))
    assert settled >= 1
    n.close()


# --------------------------------------------------------------------------------------------- the KITTI-shaped scene
def test_kitti_shaped_scans_match_oracle(ctx, oracle):
    """bench.py's `kitti_shaped` workload (synth.TerrainScene: rolling terrain, vegetation, distant blocks; the HDL-64E's beam
    table; the reference's full window of 20 key-frames) under the same bars as the headline's: NDTMANUAL on each side's own
    table within 1e-4 m / 1e-5 rad with the same iteration and pass counts, and the PCL-semantics mode on IDENTICAL tables
    within 1e-9 per pass on all nine scans — the ones its iteration does not settle on included."""
    import oracle_py
    import torch
    poses, scans = wl.kitti_dataset(workers=8)
    map_filter = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
    frame_filter = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
    map_f = map_filter.filter_device(torch.from_numpy(wl.local_map(poses, scans, wl.KITTI_MAP_KEYFRAMES)).cuda()).cpu().numpy().copy()
    srcs = [frame_filter.filter_device(torch.from_numpy(np.ascontiguousarray(scans[k])).cuda()).cpu().numpy().copy() for k in wl.KITTI_TEST_SCANS]
    guesses = [wl.guess(poses, k) for k in wl.KITTI_TEST_SCANS]
    map_filter.close(); frame_filter.close()
    tgt8 = synth.to_xyzi32(map_f)
    n = registration.NDTRegistrationHip(ctx, **wl.NDT_PARAMS)
    n.SetInputTarget(tgt8)
    o = oracle_py.OracleNDT(fast=True, **wl.NDT_PARAMS)
    o.set_input_target(tgt8)
    info = n.grid_info()
    assert (info["occupied"], info["valid"]) == o.grid_counts() and info["valid"] > 5000
    worst = [0.0, 0.0]
    kbar = []
    for src, guess in zip(srcs, guesses):
        s8 = synth.to_xyzi32(src)
        Tr, st_r = o.scan_match(s8, guess)
        _, T = n.ScanMatch(s8, guess)
        st = n.last_stats
        dt, dr = wl.pose_err(T, Tr)
        assert dt <= 1e-4 and dr <= 1e-5 and st["iterations"] == st_r["iterations"] and st["passes"] == st_r["passes"], (dt, dr, st, st_r)
        worst = [max(worst[0], dt), max(worst[1], dr)]
        kbar.append(st["pairs_per_pass"] / src.shape[0])
    print("kitti_shaped: %d source points, %d target points, %d valid voxels, K-bar %.2f; NDTMANUAL worst %.1e m / %.1e rad vs the oracle"
          % (int(np.mean([s.shape[0] for s in srcs])), map_f.shape[0], info["valid"], float(np.mean(kbar)), worst[0], worst[1]))
    n.close()
    npcl = registration.NDTRegistrationHip(ctx, pcl_mode=True, **wl.NDT_PARAMS)
    npcl.SetInputTarget(tgt8)
    gi, gr = npcl.grid_dump()
    op = oracle_py.OracleNDT(fast=True, pcl_mode=True, **wl.NDT_PARAMS)
    op.set_input_target(tgt8)
    op.override_cells(gi, gr)
    capped = 0
    for src, guess in zip(srcs, guesses):
        s8 = synth.to_xyzi32(src)
        Tr, st_r = op.scan_match(s8, guess, trace=True)
        _, T = npcl.ScanMatch(s8, guess, trace=True)
        st = npcl.last_stats
        assert st["iterations"] == st_r["iterations"] and st["passes"] == st_r["passes"]
        gap = _trace_gap(st["trace"], st_r["trace"])
        assert len(gap) and gap.max() <= 1e-9, gap.max()
        capped += int(st["iterations"] > wl.NDT_PARAMS["max_iter"])
    print("kitti_shaped, PCL semantics on identical tables: all nine scans agree per pass to 1e-9; %d of 9 run into max_iter" % capped)
    npcl.close()


# --------------------------------------------------------------------------------------------- cfg3 (BASELINE.json configs[2])
def test_cfg3_bench_scans_match_oracle(ctx, oracle, cfg2):
    """BASELINE.json configs[2] says "on same KITTI pairs": both ICP formulations on the headline's own nine scans against the
    0.3 m-filtered local map, 30 iterations, max correspondence distance 2 m — `ICPMANUAL` (icp_registration_manual.cpp:125-178,
    point-to-point Gauss-Newton, the parity mode) against oracle_py.icp_manual, and the point-to-plane build extension against
    oracle_py.icp_plane (same correspondence count in the last iteration).  2e-4 m / 2e-5 rad: the GPU accumulates the
    normal equations in f64 and rounds once, the reference (and its restatement) iterate in f32."""
    import oracle_py
    tgt8 = synth.to_xyzi32(cfg2["map_f"])
    icp = registration.ICPRegistrationHip(ctx, 2.0, 30)
    icp.SetInputTarget(tgt8)
    pl = registration.ICPRegistrationHip(ctx, 2.0, 30, mode=capi.ICP_MODE_PLANE)
    pl.SetInputTarget(tgt8)
    worst = [0.0, 0.0, 0.0, 0.0]
    for k, (src, guess) in enumerate(zip(cfg2["srcs"], cfg2["guesses"])):
        s8 = synth.to_xyzi32(src)
        T_ref = oracle_py.icp_manual(tgt8, s8, guess, 2.0, 30)
        _, T = icp.ScanMatch(s8, guess)
        dt, dr = wl.pose_err(T, T_ref)
        assert dt <= 2e-4 and dr <= 2e-5, ("icpmanual", k, dt, dr)
        worst[0], worst[1] = max(worst[0], dt), max(worst[1], dr)
        Tp_ref, ncorr_ref = oracle_py.icp_plane(tgt8, s8, guess, 2.0, 30, 1.0)
        _, Tp = pl.ScanMatch(s8, guess)
        dt, dr = wl.pose_err(Tp, Tp_ref)
        assert dt <= 2e-4 and dr <= 2e-5, ("plane", k, dt, dr)
        assert int(pl.last_stats["correspondences"]) == ncorr_ref, (k, pl.last_stats, ncorr_ref)
        worst[2], worst[3] = max(worst[2], dt), max(worst[3], dr)
    print("cfg3 on the nine bench scans: ICPMANUAL worst %.1e m / %.1e rad, point-to-plane worst %.1e m / %.1e rad vs the oracle" % tuple(worst))
    icp.close(); pl.close()


def test_icp_skipped_searches_change_nothing(ctx, cfg2, monkeypatch):
    """From the second ICP iteration on a query whose old neighbour is PROVABLY still the nearest (proof radius left by its last
    search, csrc/icp.hip icp_search) does not search.  The bar is BITS against the build that runs every search
    (LSA_ICP_NO_SKIP=1): final poses and correspondence counts of ICPMANUAL and point-to-plane on the nine bench scans — from the
    bench's guess and from a guess 0.6 m / 2 deg off, where the pose keeps moving for many iterations and queries change
    neighbours — and the normal equations after 1, 2, 3 and 7 iterations (a wrong neighbour anywhere would change the sums)."""
    tgt8 = synth.to_xyzi32(cfg2["map_f"])
    off = synth.pose_matrix(0.5, -0.3, 0.05, 0.004, -0.003, 0.035).astype(np.float32)
    runs = {}
    for label in ("skip", "full"):
        if label == "full":
            monkeypatch.setenv("LSA_ICP_NO_SKIP", "1")
            monkeypatch.setenv("LSA_ICP_NO_ARMED", "1")   # ... and one launch per iteration instead of the armed passes
        out = []
        for mode in (capi.ICP_MODE_MANUAL, capi.ICP_MODE_PLANE):
            for iters in (30, 1, 2, 3, 7):
                icp = registration.ICPRegistrationHip(ctx, 2.0, iters, mode=mode)
                icp.SetInputTarget(tgt8)
                for k, (src, guess) in enumerate(zip(cfg2["srcs"], cfg2["guesses"])):
                    if iters != 30 and k > 2:
                        break
                    s8 = synth.to_xyzi32(src)
                    for g in (guess, (guess @ off).astype(np.float32)):
                        _, T = icp.ScanMatch(s8, g)
                        out.append((T.copy(), int(icp.last_stats["correspondences"])))
                armed_passes, fallbacks = icp.armed_counts()
                # (a fallback needs the host to stall for 0.3 ms between two iterations: possible on a loaded box, counted, harmless)
                assert fallbacks <= 1 + armed_passes // 100 and (armed_passes > 0) == (label == "skip" and iters > 1), (label, iters, armed_passes, fallbacks)
                icp.close()
        runs[label] = out
    assert len(runs["skip"]) == len(runs["full"]) > 40
    for i, ((Ta, ca), (Tb, cb)) in enumerate(zip(runs["skip"], runs["full"])):
        assert ca == cb and np.array_equal(Ta, Tb), (i, ca, cb, Ta - Tb)
    # ... and searches ARE skipped: every query searches in the first iteration, a few per cent of them once the pose has settled
    # (lsa_icp_search_counts); with LSA_ICP_NO_SKIP every query searches in every iteration
    monkeypatch.delenv("LSA_ICP_NO_SKIP")
    monkeypatch.delenv("LSA_ICP_NO_ARMED")
    s8 = synth.to_xyzi32(cfg2["srcs"][0])
    for mode, settled in ((capi.ICP_MODE_MANUAL, 0.05), (capi.ICP_MODE_PLANE, 0.01)):
        icp = registration.ICPRegistrationHip(ctx, 2.0, 30, mode=mode)
        icp.SetInputTarget(tgt8)
        icp.ScanMatch(s8, cfg2["guesses"][0])
        searches, queries, per = icp.search_counts()
        assert len(per) == 30 and per[0] * 30 == queries and sum(per) == searches, (per, queries)
        assert per[-1] <= settled * per[0] and searches < 0.3 * queries, (mode, per)
        icp.close()
    # a pre-launched kernel that runs out of patience before its pose is posted leaves without rows; the iteration is then run the
    # ordinary way, counted, and the result is the same: patience of 1 us (the host needs ~5 to solve and post) and of 8 us (some
    # workgroups take the mail, others have just left)
    want = [r[0] for r in runs["skip"][:2]]
    for patience in (1.0, 3.0, 4.0, 5.0, 8.0, 0.0):
        icp = registration.ICPRegistrationHip(ctx, 2.0, 30)
        icp.set_armed_patience(patience)
        icp.SetInputTarget(tgt8)
        g0 = cfg2["guesses"][0]
        for rep in range(3):
            _, Ta = icp.ScanMatch(s8, g0)
            _, Tb = icp.ScanMatch(s8, (g0 @ off).astype(np.float32))
            assert np.array_equal(Ta, want[0]) and np.array_equal(Tb, want[1]), (patience, rep)
        ap, fb = icp.armed_counts()
        print("ICP armed patience %.0f us: %d armed iterations, %d fallbacks in 6 ScanMatches" % (patience, ap, fb))
        assert fb >= 1 if patience <= 1.0 else fb >= 0
        icp.close()
    monkeypatch.setenv("LSA_ICP_NO_SKIP", "1")
    icp = registration.ICPRegistrationHip(ctx, 2.0, 5)
    icp.SetInputTarget(tgt8)
    icp.ScanMatch(s8, cfg2["guesses"][0])
    searches, queries, per = icp.search_counts()
    assert searches == queries and len(set(per)) == 1
    icp.close()


# --------------------------------------------------------------------------------------------- cfg4 (BASELINE.json configs[3])
@pytest.fixture(scope="module")
def cfg4(cfg2):
    poses, scans = cfg2["poses"], cfg2["scans"]
    target = wl.submap_2m(poses, scans)
    src, guess = wl.cfg4_source(poses)
    assert target.shape[0] == 2000000 and 120000 < src.shape[0] < 140000
    return target, src, guess


def test_cfg4_raw_scan_against_2m_submap(ctx, oracle, cfg4):
    import oracle_py
    import torch
    target, src, guess = cfg4
    d_t = torch.from_numpy(target).cuda()
    d_s = torch.from_numpy(src).cuda()
    n = registration.NDTRegistrationHip(ctx, **wl.NDT_PARAMS)
    n.SetInputTarget(d_t)
    p = oracle.matrix_to_pose(guess)
    n.set_source(d_s)
    a = n.derivatives(guess, p, 0)
    b = n.derivatives(guess, p, 0)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])      # bitwise repeatable
    M = src.shape[0]
    assert int(a[4]) == M and a[3] > M                                                      # every point counted, K-bar > 1
    # the configs[3] split: 8 contiguous shards of the source, target replicated; shard sums == whole
    tot_s, tot_g, tot_H, tot_pairs = 0.0, np.zeros(6), np.zeros((6, 6)), 0
    for r in range(8):
        lo, hi = (M * r) // 8, (M * (r + 1)) // 8
        n.set_source(d_s[lo:hi])
        s, g, H, pairs, npts = n.derivatives(guess, p, 0)
        assert int(npts) == hi - lo
        tot_s += s; tot_g += g; tot_H += H; tot_pairs += int(pairs)
    assert tot_pairs == int(a[3])
    assert abs(tot_s - a[0]) <= 1e-11 * abs(a[0])
    assert np.max(np.abs(tot_g - a[1])) <= 1e-10 * np.max(np.abs(a[1]))
    assert np.max(np.abs(tot_H - a[2])) <= 1e-10 * np.max(np.abs(a[2]))
    # ONE oracle derivative pass: a 1/16 subsample of the source against the FULL 2 M-point target (seconds on the CPU)
    sub = np.ascontiguousarray(src[::16])
    o = oracle_py.OracleNDT(fast=True, **wl.NDT_PARAMS)
    o.set_input_target(synth.to_xyzi32(target))
    occ, valid = o.grid_counts()
    info = n.grid_info()
    assert (info["occupied"], info["valid"]) == (occ, valid)
    s_r, g_r, H_r, pairs_r = o.derivatives(synth.to_xyzi32(sub), guess, p)
    n.set_source(torch.from_numpy(sub).cuda())
    s, g, H, pairs, npts = n.derivatives(guess, p, 0)
    assert int(pairs) == pairs_r and int(npts) == sub.shape[0]
    assert abs(s - s_r) <= 1e-7 * abs(s_r)
    assert np.max(np.abs(g - g_r)) <= 1e-7 * np.max(np.abs(g_r))
    assert np.max(np.abs(H - 0.5 * (H_r + H_r.T))) <= 1e-7 * np.max(np.abs(H_r))
    # the align itself on the raw scan: the guess is the true pose, NDT stays there; same result twice
    _, T1 = n.ScanMatch(d_s, guess)
    st1 = dict(n.last_stats)
    _, T2 = n.ScanMatch(d_s, guess)
    assert np.array_equal(T1, T2) and st1["passes"] == n.last_stats["passes"]
    assert wl.pose_err(T1, guess)[0] < 0.1
    n.close()
