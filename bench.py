#!/usr/bin/env python3
"""bench.py — headline benchmark of the scan-matching front end on MI355X.

Metric (BASELINE.json): KITTI HDL-64E scans/sec at converged NDT (+ final-pose error against the CPU oracle).
A "step" is one scan through the hot path:  VoxelFilter(leaf 0.5) -> NDT ScanMatch against the resident local-map
voxel table (res 1.0, step 0.1, eps 0.01, max_iter 30, outlier 0.55 — config/mapping/front_end.yaml), i.e.
BASELINE.json configs[1] ("KITTI HDL-64E (~120k pts) NDT scan-to-local-map, voxel 1.0 m, 1xMI355X") restated on the
deterministic synthetic generator (no KITTI data in the reference tree or the image).
Inputs (raw scans, local map) are resident in HBM before the timed region; SetInputTarget is per key-frame work and
is reported separately (`config.set_target_ms`).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Two ways to use the GPUs, chosen by the measured break-even of dist.shard_pays() on the
size of the filtered source (same rule on every rank; LSA_BENCH_SHARD=1 / 0 forces one):
  * sharded (BASELINE.json's north_star; pays from a few hundred thousand source points per pass, i.e. cfg4 / cfg5-sized
    clouds): every rank filters the scan, takes its contiguous shard of the filtered source, and the 32-double
    normal-equation block is summed with ONE RCCL all-reduce per derivative pass; value = steps / time, "strong";
  * independent scan streams (KITTI-sized scans, where a 28 us collective per 8 us pass would make N GPUs SLOWER than
    one): rank r registers its own stream of scans against its own copy of the local map, no data-path collective;
    value = N * steps / time, "weak".
The mode that was NOT chosen is timed too (a quarter of the steps, after the headline run) and reported under
`multi_gpu_alt`, and `cfg5_sharded` times the exchange step at the size it is meant for (2,097,152-point source sharded
over the ranks), so the scaling of the exchange step is on the line either way.  Barriers and the max over ranks run on
a gloo group; the RCCL communicator is created only when a mode with an exchange step is about to run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s
F64_VECTOR_PEAK_TFLOPS = 78.6              # MI355X f64 vector peak

# f64 flops per (source point, target voxel) pair, counted from pair_update<KIND> (csrc/ndt_pass_dev.h), a multiply and an
# add each counting 1 (an FMA 2), exp() priced at 30:
#   common  d = x' - mu 3, radius test 5, u = C d 15, q = d.u 5, exponent 2 (+ exp), f 2, u.J3 / u.J4 / u.J5 13      = 45 + exp
#   kind 0  + score 3 + gradient 12 + Hessian: 6 translation entries 30, column 3 64, column 4 62, column 5 46, -d2 1  = 263 + exp
#   kind 1  + score 3 + gradient 12                                                                                    = 60 + exp
#   kind 2  kind 0 without score and gradient                                                                          = 248 + exp
# and once per source POINT (phase 0): f32 transform 18 + 23 dot products of computePointDerivatives 115 = 133.
# SURVEY.md 8d's own estimate is 520 per pair for a formulation that does not share C J_i between Hessian entries.
FLOPS_PER_PAIR = {"kind0": 293, "kind1": 90, "kind2": 278, "per_point": 133, "exp_priced_at": 30, "survey_estimate": 520}


def _load_pkg():
    import importlib.util
    if "lidar_slam_amd" in sys.modules:
        return
    pkg_dir = os.path.join(ROOT, "lidar-slam_amd")
    spec = importlib.util.spec_from_file_location("lidar_slam_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["lidar_slam_amd"] = mod
    spec.loader.exec_module(mod)


def _under_profiler():
    """rocprofv3's preloaded tool library has initialised the GPU before this program starts: no worker pool then."""
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def pose_err(Ta, Tb):
    _load_pkg()
    from lidar_slam_amd import workloads as wl
    return wl.pose_err(Ta, Tb)


def vector_f64(pairs, points, ms, kind="kind0"):
    """f64-vector roofline of one pass launch: (pairs x flops per pair + points x per-point flops) / time against 78.6 TF/s."""
    fl = pairs * FLOPS_PER_PAIR[kind] + points * FLOPS_PER_PAIR["per_point"]
    tf = fl / (ms * 1e-3) / 1e12
    return {"achieved": round(tf, 3), "peak": F64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / F64_VECTOR_PEAK_TFLOPS, 4),
            "flops_per_launch": int(fl)}


def spread(ts_ms):
    ts_ms = np.asarray(ts_ms, dtype=np.float64)
    return {"min": round(float(ts_ms.min()), 4), "median": round(float(np.median(ts_ms)), 4), "max": round(float(ts_ms.max()), 4), "n": int(ts_ms.size)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--ramp-seconds", type=float, default=1.5, help="untimed clock-ramp spin before the W warm-up steps")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bound on the CPU-baseline sample (rank 0, N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large", action="store_true", help="skip the cfg5-sized roofline probe of the pass kernel")
    ap.add_argument("--streams", type=int, default=4, help="extra key `concurrent_streams`: S independent scan streams on ONE GPU (0: skip)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra keys that run other workloads through the same kernels (dense / kitti_shaped scenes, front-end loop, ICP): "
                    "a kernel trace of the run then holds the headline's dispatches only")
    args = ap.parse_args()
    if os.environ.get("LSA_BENCH_DUMP_AFTER"):   # diagnostics: where every thread stands if the run is still going after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LSA_BENCH_DUMP_AFTER"]), exit=False)

    # stdout carries exactly ONE line, the JSON result: RCCL prints a version banner on stdout when its first communicator
    # comes up, gloo a "[Gloo] Rank ..." line, so everything any library writes to fd 1 is sent to stderr instead and the
    # result goes to a private duplicate of the original stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # LSA_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, RCCL all-reduce per pass, barrier, max over
    # ranks) even for a world of one, so a one-GPU box rehearses exactly what `--gpus N` under torch.distributed.run does
    dist_mode = args.gpus > 1 or os.environ.get("LSA_BENCH_FORCE_DIST") == "1"
    world = int(os.environ.get("WORLD_SIZE", "1")) if dist_mode else 1
    rank = int(os.environ.get("RANK", "0")) if dist_mode else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if dist_mode else 0
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))

    def progress(msg):
        """one line per phase on stderr (rank 0): a long run must keep writing or the GPU box's watchdog takes it for hung"""
        if rank == 0:
            print("[bench %6.1fs] %s" % (time.time() - t_start, msg), file=sys.stderr, flush=True)

    extras_failed = {}

    class guarded:
        """An extra key's section: a failure in it is recorded under "extras_failed" in the JSON line and on stderr — the headline
        line is printed whatever happens to an extra.  (Single-process sections only: a rank that skipped a collective would
        hang the others.)"""
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            return self

        def __exit__(self, et, ev, tb):
            if et is not None and issubclass(et, Exception):
                import traceback
                extras_failed[self.name] = "%s: %s" % (et.__name__, ev)
                print("[bench] section '%s' failed: %s: %s" % (self.name, et.__name__, ev), file=sys.stderr, flush=True)
                traceback.print_tb(tb, file=sys.stderr)
                return True
            return False

    t_start = time.time()
    _load_pkg()
    from lidar_slam_amd import workloads as wl
    NDT_PARAMS, MAP_KEYFRAMES, TEST_SCANS = wl.NDT_PARAMS, wl.MAP_KEYFRAMES, wl.TEST_SCANS
    workers = 1 if _under_profiler() else max(1, min(8, (os.cpu_count() or 8) // max(1, world)))
    poses, scans = wl.cfg2_dataset(workers, leader=(local_rank == 0))

    import torch
    import torch.distributed as dist
    _load_pkg()
    from lidar_slam_amd import capi, cloud_filter, registration, synth
    from lidar_slam_amd import dist as lsa_dist

    # rehearsal on a one-GPU box: LSA_BENCH_ONE_GPU=1 maps every rank to cuda:0 and swaps the collective for gloo
    # (RCCL refuses two ranks on one device); the production launch is one rank per GPU over RCCL
    one_gpu = os.environ.get("LSA_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    if dist_mode:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # Control plane (barriers, the max over ranks of the elapsed time) on gloo; the data-path collective is brought up
        # only where the path has an exchange step (ensure_exchange() below).  Measured on one GPU: with torch's "nccl"
        # process group initialised, the SAME collective-free pipeline runs 25 % slower (10.6 k vs 14 k scans/s; its
        # watchdog / heartbeat machinery, not RCCL — the library's own RCCL communicator costs nothing), which would be
        # booked as scaling loss of the independent scan streams.
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", dev_index)
    # one non-default stream shared by torch (uploads, RCCL collectives) and the library (kernels, copies):
    # everything the hot path does is ordered on it, no cross-stream events needed
    tstream = torch.cuda.Stream(device=dev, priority=int(os.environ["LSA_STREAM_PRIORITY"])) if os.environ.get("LSA_STREAM_PRIORITY") else torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    ctx = capi.Context(dev_index, tstream.cuda_stream)

    # the frame filter gets its own context / HIP stream: the filter of scan k+1 is enqueued before the registration of
    # scan k and overlaps it (two-stage software pipeline over the stream of scans; every scan still goes through both
    # stages, and the timed region contains exactly K filters and K registrations)
    fstream = torch.cuda.Stream(device=dev)
    ctx_f = capi.Context(dev_index, fstream.cuda_stream)
    frame_filter = cloud_filter.VoxelFilterHip(ctx_f, 0.5, 0.5, 0.5)     # front_end.yaml:46-47
    map_filter = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)         # front_end.yaml:44-45
    ndt = registration.NDTRegistrationHip(ctx, **NDT_PARAMS)

    progress("dataset ready")
    # ---- local map (front_end.cpp:348-424): key-frames transformed into the map frame, concatenated, filtered at 0.3
    P0inv = np.linalg.inv(poses[0])
    local_map = wl.local_map(poses, scans)
    d_map = torch.from_numpy(local_map).to(dev)
    torch.cuda.synchronize()
    # per key-frame work: 0.3 m map filter (1.4 M points) + voxel table; the first call sizes the buffers, the second is
    # the steady-state cost a key-frame pays
    # (median of 5 after 12 untimed repetitions: the section starts behind seconds of CPU-only dataset generation, i.e. on idle GPU clocks)
    st_ms = []
    for it in range(17):
        t0 = time.perf_counter()
        d_map_f = map_filter.filter_device(d_map)
        ndt.SetInputTarget(d_map_f.contiguous())
        torch.cuda.synchronize()
        if it >= 12:
            st_ms.append((time.perf_counter() - t0) * 1e3)
    set_target_ms = float(np.median(st_ms))
    ginfo = ndt.grid_info()

    # ---- test scans resident in HBM + constant-velocity guesses (front_end.cpp:238-240) from the true trajectory
    d_scans, guesses, truths = [], [], []
    for k in TEST_SCANS:
        d_scans.append(torch.from_numpy(np.ascontiguousarray(scans[k])).to(dev))
        guesses.append(wl.guess(poses, k))     # constant-velocity prediction + a seeded 8 cm / 0.5 deg prediction error
        truths.append(wl.truth(poses, k))
    d_filt2 = [torch.empty((max(s.shape[0] for s in d_scans), 4), dtype=torch.float32, device=dev) for _ in range(2)]
    d_filt = d_filt2[0]
    # N > 1: shard the source over the ranks (one all-reduce per pass) only where that pays; a KITTI-sized filtered scan
    # is faster replicated (dist.shard_pays).  LSA_BENCH_SHARD=1 / 0 forces either mode.
    exchange = {"cb": None, "native": None, "ready": False}

    def ensure_exchange():
        """Bring up the data-path collective (collective call: every rank, same point of the program)."""
        if exchange["ready"] or not dist_mode:
            return
        if one_gpu:
            exchange["cb"] = lsa_dist.make_gloo_staged_allreduce()
        else:
            # one rank per GPU over RCCL, issued by the library itself on its own stream (lsa_rccl_*: ncclAllReduce behind the
            # C-ABI, no Python per pass; the 128-byte unique id travels over the gloo group).  LSA_BENCH_NATIVE_RCCL=0 takes
            # torch.distributed's "nccl" backend on a process group of its own instead: measured in the world-of-one
            # rehearsal 154 us per sharded cfg2 step against 99 us.
            # Bring-up is agreed on by ALL ranks (a min over the gloo group): if the library's own communicator fails anywhere
            # (librccl not loadable, ncclCommInitRank error), every rank falls back to torch.distributed's "nccl" group together
            # instead of one rank raising while the others wait in a collective; the line says which backend carried the sums.
            want_native = os.environ.get("LSA_BENCH_NATIVE_RCCL", "1") != "0"
            ok = 1
            if want_native:
                try:
                    exchange["native"] = lsa_dist.NativeRccl(ctx, rank, world)
                except Exception as e:  # noqa: BLE001
                    ok = 0
                    print("[bench] native RCCL bring-up failed on rank %d: %s" % (rank, e), file=sys.stderr, flush=True)
                flag = torch.tensor([ok], dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    if exchange["native"] is not None:
                        exchange["native"].close()
                        exchange["native"] = None
                    exchange["fallback"] = "library RCCL communicator failed on at least one rank"
            if exchange["native"] is None:
                exchange["cb"] = lsa_dist.make_rccl_allreduce(group=dist.new_group(backend="nccl"))
        exchange["ready"] = True

    m_probe = int(frame_filter.filter_device(d_scans[0], d_filt).shape[0])
    force = os.environ.get("LSA_BENCH_SHARD")
    shard_policy = dist_mode and (force == "1" or (force != "0" and lsa_dist.shard_pays(m_probe, world)))
    shard_now = [False]

    def set_mode(shard):
        shard_now[0] = bool(shard)
        if dist_mode:
            if shard:
                ensure_exchange()
            if exchange["native"] is not None and shard:
                exchange["native"].attach(ndt)
            else:
                ndt.set_allreduce(exchange["cb"] if shard else None)

    results = {}

    # The timed region is ONE call into the C-ABI per run: lsa_pipeline_scan_match walks `count` scans through
    # Filter -> ScanMatch with the filter of scan k+1 enqueued (own context / HIP stream) before the registration of scan
    # k, i.e. the loop front_end_node's C++ would run; every scan is filtered once and registered once.
    import ctypes as C
    L = capi.lib()
    L.lsa_pipeline_scan_match.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_float),
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(capi.NdtStats), C.POINTER(C.c_size_t)]
    ns = len(d_scans)
    guess_cm = np.stack([capi.colmajor(g) for g in guesses]).astype(np.float32)   # (ns, 16) column-major poses
    hf, hn = frame_filter.h, ndt.h
    last_run = {}

    def run_scans(count):
        """count scans through the pipeline; scan i of the run is test scan (i + off) % ns."""
        # independent-streams mode: every rank walks its own scan stream (offset by its rank); sharded mode: all ranks walk
        # the same stream and each takes its shard of every scan
        off = 0 if shard_now[0] else rank
        idx = [(i + off) % ns for i in range(count)]
        ptrs = (C.c_void_p * count)(*[d_scans[k].data_ptr() for k in idx])
        nn = (C.c_size_t * count)(*[d_scans[k].shape[0] for k in idx])
        pred = np.ascontiguousarray(guess_cm[idx])
        out = np.zeros((count, 16), dtype=np.float32)
        st = (capi.NdtStats * count)()
        nf = (C.c_size_t * count)()
        r, w = (rank, world) if shard_now[0] else (0, 1)

        def call():
            rc = L.lsa_pipeline_scan_match(hf, hn, count, ptrs, nn, capi.fptr(pred), d_filt2[0].data_ptr(), d_filt2[1].data_ptr(), r, w,
                                           capi.fptr(out), st, nf)
            if rc:
                ctx_f.check(rc)
        last_run.update(idx=idx, out=out, st=st, nf=nf)
        return call

    def collect():
        idx, out, st, nf = last_run["idx"], last_run["out"], last_run["st"], last_run["nf"]
        for i in range(max(0, len(idx) - ns), len(idx)):   # the last visit of every test scan
            results[idx[i]] = (capi.from_colmajor(out[i]), st[i].as_dict(), int(nf[i]))

    def sync_all(local=False):
        # through the library first: lsa_context_synchronize releases anything a handle has left parked on the context's stream
        # before it waits (a bare device-wide wait would sit out a parked kernel's patience)
        # local: this rank only — the sections only rank 0 runs must not enter a collective the other ranks never reach
        ctx.synchronize(); ctx_f.synchronize()
        torch.cuda.synchronize()
        if dist_mode and not local:
            dist.barrier()
            torch.cuda.synchronize()

    def run_timed(shard, steps, warmup, ramp_seconds):
        """W untimed warm-ups (after a clock ramp), then exactly `steps` timed steps between barriers; max over ranks."""
        set_mode(shard)
        # clock ramp: a step is ~0.1 ms of latency-bound kernels, so W steps alone end before the GPU has left its idle
        # DVFS state; spin the same work (untimed, results discarded) first, then do the W warm-ups.  Steps may contain
        # collectives, so all ranks run the SAME number of ramp steps: count-based, not timed.
        n_ramp = int(ramp_seconds * 8000)
        if n_ramp:
            run_scans(n_ramp)()
        if warmup:
            run_scans(warmup)()
        timed = run_scans(steps)
        sync_all()
        ndt.armed_timing(reset=True)
        t0 = time.perf_counter()
        timed()
        # (the batch returns with nothing parked on either stream — tests/test_gpu_armed_stalls.py — so the device-wide wait
        # is all that is left to do here)
        torch.cuda.synchronize()
        if dist_mode:
            dist.barrier()
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        last_run["armed_timing"] = ndt.armed_timing(reset=True)   # device-side wait / body of the armed passes of the timed region
        if dist_mode:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    progress("inputs resident, headline run")
    # the headline run, in the mode the policy chose (the exchange step is brought up first only if that mode needs it)
    elapsed = run_timed(shard_policy, args.steps, args.warmup, args.ramp_seconds)
    collect()
    armed_after_headline = ndt.armed_stats()   # passes served by a kernel enqueued ahead / cancelled / redone / late / expired ...
    armed_timing_headline = last_run.get("armed_timing")
    # N > 1: also time the mode the policy did NOT choose (a quarter of the steps) so the line shows both
    alt = None
    if dist_mode and world > 1:
        k_alt = max(50, args.steps // 4)
        dt_alt = run_timed(not shard_policy, k_alt, min(args.warmup, 50), min(args.ramp_seconds, 0.25))
        units_alt = k_alt * (1 if not shard_policy else world)
        alt = {"mode": "every scan sharded over the ranks, one all-reduce per pass (strong scaling)" if not shard_policy
               else "one independent scan stream per rank, no collective (weak scaling)", "steps": k_alt,
               "value": round(units_alt / dt_alt, 2), "ms_per_step": round(dt_alt / k_alt * 1e3, 4)}

    progress("headline done, component split")
    # ---- component split + roofline of the dominant kernel (rank 0; HIP events on the library's stream)
    roofline = None
    split = {}
    if rank == 0:
        set_mode(False)
        k = 0
        reps = 50
        sync_all(local=True)

        def per_call(fn):
            """min / median / max of `reps` synchronous calls (each returns with its result on the host): one outlier — a page
            fault of a cold box, a collection of the interpreter — shows up as `max` instead of poisoning a mean"""
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                r = fn()
                ts.append(time.perf_counter() - t0)
            ts = np.array(ts) * 1e3
            return r, float(np.median(ts)), {"min": round(float(ts.min()), 4), "median": round(float(np.median(ts)), 4), "max": round(float(ts.max()), 4)}
        f, split["filter_ms"], split["filter_ms_spread"] = per_call(lambda: frame_filter.filter_device(d_scans[k], d_filt))
        f = f.contiguous()
        ndt.ScanMatch(f, guesses[k])
        _, split["align_ms"], split["align_ms_spread"] = per_call(lambda: ndt.ScanMatch(f, guesses[k]))
        sync_all(local=True)
        split["armed_after_split"] = ndt.armed_stats()
        # PCIe-inclusive rate through the host-pointer boundary (what a PCL caller sees): pack 32-B PointXYZI -> 16 B,
        # H2D, filter, D2H, H2D, align.  Reported for DESIGN.md; it is never the headline `value`.
        raw8 = synth.to_xyzi32(scans[TEST_SCANS[k]])
        # (filter and registration on ONE context, as INTEGRATION.md's adaptor creates them)
        hb_filter = cloud_filter.VoxelFilterHip(ctx, 0.5, 0.5, 0.5)
        f8_cloud = np.zeros((raw8.shape[0], 8), dtype=np.float32)   # the caller's filtered_cloud_ptr, reused scan after scan (front_end.cpp:106-107)
        for _ in range(3):                     # untimed: first use of the host path allocates its pinned staging buffers
            _, f8 = hb_filter.Filter(raw8, f8_cloud)
            ndt.ScanMatch(f8, guesses[k])
        tf_, ta_ = [], []
        for _ in range(30):
            t0 = time.perf_counter()
            _, f8 = hb_filter.Filter(raw8, f8_cloud)
            t1 = time.perf_counter()
            ndt.ScanMatch(f8, guesses[k])
            tf_.append(t1 - t0)
            ta_.append(time.perf_counter() - t1)
        hb_filter.close()
        # medians: the first calls through the host path still pay one-off runtime set-up (copy queues, staging buffers)
        split["host_boundary_filter_ms"] = float(np.median(tf_)) * 1e3
        split["host_boundary_align_ms"] = float(np.median(ta_)) * 1e3
        split["host_boundary_ms_per_scan"] = split["host_boundary_filter_ms"] + split["host_boundary_align_ms"]
        # ... and the same boundary PIPELINED (lsa_pipeline_scan_match_host: a run of host clouds, the next two packed and crossing the
        # link while the current one is registered): the PCIe-inclusive rate of the hot path when the caller can hand over a queue
        host_pipe = None
        try:
            if args.no_extra:
                raise RuntimeError("skipped (--no-extra)")
            n_hf = int(os.environ.get("LSA_BENCH_HOST_FILTERS", "2"))   # (measured: 2 -> 12.7 k, 3 -> 11.4 k, 4 -> 11.8 k scans/s: not latency but host memory traffic bounds it)
            chs = [capi.Context(dev_index, None) for _ in range(n_hf)]
            fhs = [cloud_filter.VoxelFilterHip(c_, 0.5, 0.5, 0.5) for c_ in chs]
            fh_arr = (C.c_void_p * n_hf)(*[f_.h for f_ in fhs])
            raw_all = [synth.to_xyzi32(scans[kk]) for kk in TEST_SCANS]
            Lh = capi.lib()
            Lh.lsa_pipeline_scan_match_host.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t,
                                                        C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(capi.NdtStats), C.POINTER(C.c_size_t)]

            def host_run(count):
                idx_h = [i % ns for i in range(count)]
                ptrs_h = (C.c_void_p * count)(*[raw_all[i].ctypes.data for i in idx_h])
                nn_h = (C.c_size_t * count)(*[raw_all[i].shape[0] for i in idx_h])
                pred_h = np.ascontiguousarray(guess_cm[idx_h])
                out_h = np.zeros((count, 16), dtype=np.float32)
                st_h = (capi.NdtStats * count)()
                nf_h = (C.c_size_t * count)()
                t0 = time.perf_counter()
                rc_h = Lh.lsa_pipeline_scan_match_host(fh_arr, n_hf, ndt.h, count, ptrs_h, nn_h, 32, 16, capi.fptr(pred_h), capi.fptr(out_h), st_h, nf_h)
                sync_all(local=True)
                dt_h = time.perf_counter() - t0
                if rc_h:
                    ctx.check(rc_h)
                return dt_h, out_h, idx_h
            host_run(300)
            dts_h = []
            for _ in range(3):
                dt_h, out_h, idx_h = host_run(1000)
                dts_h.append(dt_h / 1000 * 1e3)
            same_h = all(np.array_equal(capi.from_colmajor(out_h[i]), results[idx_h[i]][0]) for i in range(len(idx_h) - ns, len(idx_h)) if idx_h[i] in results)
            host_pipe = {"value": round(1e3 / float(np.median(dts_h)), 1), "unit": "scans/s", "ms_per_scan": round(float(np.median(dts_h)), 4), "ms_per_scan_spread": spread(dts_h),
                         "scans_pulled_by_the_device": int(sum(f_.counters()["pulled"] for f_ in fhs)), "poses_identical_to_the_device_path": bool(same_h), "filters": n_hf,
                         "what": "lsa_pipeline_scan_match_host: 1000 host PointXYZI scans (32-byte rows) per call, pack + PCIe + filter + ScanMatch, ingest %d scans deep; PCIe-inclusive, never `value`" % n_hf}
            for f_ in fhs:
                f_.close()
            for c_ in chs:
                c_.close()
        except Exception as exc:   # an extra key must not take the line down
            host_pipe = {"error": repr(exc)}
        split["host_pipeline"] = host_pipe
        ndt.set_source(f)
        L = capi.lib()
        p = np.zeros(6)
        Tc = capi.colmajor(guesses[k])
        L.lsa_matrix_to_pose(capi.fptr(Tc), capi.dptr(p))
        (score, g, H, pairs, npts), ms = ndt.derivatives_timed(guesses[k], p, kind=0, repeat=200)
        M = int(npts)
        kbar = pairs / max(1, M)
        # SURVEY.md 8d formula C: algorithmic bytes per derivative pass = M * (16 + 4*27 + 48*Kbar)
        alg_bytes = M * (16.0 + 4.0 * 27.0 + 48.0 * kbar)
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        traffic = None
        for tname in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):   # PMC passes of this command, see profiles/README.md
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    traffic = tj.get("k_ndt_pass_armed_bytes_per_launch", tj.get("k_ndt_pass_kind0_bytes_per_launch"))
                    break
                except Exception:
                    traffic = None
        # The pipeline runs the ARMED form of the kernel (k_ndt_pass_armed: the same body behind a wait for its parameters), so
        # that is the dominant kernel of the timed region.  Its launch duration = the device-side wait for the mail + the body,
        # both stamped by the kernel itself with the 100 MHz wall clock and averaged over the workgroups and over every armed
        # pass of the TIMED region (lsa_ndt_armed_timing); rocprofv3's kernel trace of the same command reports the same span
        # per dispatch (profiles/).  `isolated` is the unarmed k_ndt_pass<0> of the same pass, back to back between HIP events.
        at = armed_timing_headline or {}
        us_armed = (at.get("dev_wait_us", 0.0) + at.get("dev_body_us", 0.0)) if at.get("passes") else None
        us_roof = us_armed if us_armed else ms * 1e3
        achieved_roof = alg_bytes / (us_roof * 1e-6) / 1e9
        # what rocprofv3's kernel trace of the same command says for that kernel (committed summary under profiles/; the traced run
        # is slower on the host side — the tracer sits in every API call — so its kernels wait longer for their mail than here)
        rocprof_avg = None
        # (r05_final_headline_*: the trace of `bench.py --steps 2000 --warmup 20 --ramp-seconds 0.5 --no-cpu-baseline --no-large --streams 0 --no-extra`,
        # i.e. of the headline's own dispatches; the *_bench_kernel_stats.csv beside it averages over every section of the full command —
        # the front-end loops' armed kernels wait through a frame filter for their mail)
        for pname in ("r05_final_headline_kernel_stats.csv", "r05_final_bench_kernel_stats.csv", "r04_final_headline_kernel_stats.csv", "r04_final_bench_kernel_stats.csv", "r03_final_bench_kernel_stats.csv", "r03_a_bench_kernel_stats.csv"):
            ppath = os.path.join(ROOT, "profiles", pname)
            if os.path.exists(ppath):
                try:
                    import csv
                    for row in csv.DictReader(open(ppath)):
                        if "k_ndt_pass_armed<true>" in row["Name"]:
                            rocprof_avg = {"file": "profiles/" + pname, "dispatches": int(row["Calls"]), "avg_us": round(float(row["AverageNs"]) / 1e3, 3)}
                            break
                except Exception:
                    rocprof_avg = None
                if rocprof_avg:
                    break
        if rocprof_avg:
            rocprof_avg["frac_at_this_duration"] = round(alg_bytes / (rocprof_avg["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
            rocprof_avg["why_longer_than_us_per_launch"] = ("a dispatch's begin -> end also holds the launch ramp / drain (~2 us), the dispatches that end by a cancel or "
                                                            "run out of patience (up to 300 us each) count too, and in the traced run the host — hence every mail — is slower: "
                                                            "that run's own device stamps are in profiles/*_bench_traced_run.json (roofline.armed)")
        roofline = {"bound": "hbm", "kernel": "k_ndt_pass_armed<true>" if us_armed else "k_ndt_pass<0, true>",
                    "rocprof_kernel_trace": rocprof_avg,
                    "achieved": round(achieved_roof, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved_roof / HBM_PEAK_GBS, 5), "traffic": traffic, "us_per_launch": round(us_roof, 3),
                    "armed": {"passes_timed": at.get("passes"), "dev_wait_us": round(at.get("dev_wait_us", 0.0), 3), "dev_body_us": round(at.get("dev_body_us", 0.0), 3),
                              "host_post_to_rows_us": round(at.get("post_to_rows_us", 0.0), 3), "host_rows_to_post_us": round(at.get("rows_to_post_us", 0.0), 3),
                              "frac_body_only": round(alg_bytes / (max(at.get("dev_body_us", 0.0), 1e-9) * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if us_armed else None},
                    "isolated": {"kernel": "k_ndt_pass<0, true>", "us_per_launch": round(ms * 1e3, 3), "achieved": round(achieved, 2),
                                 "frac": round(achieved / HBM_PEAK_GBS, 5), "how": "200 back-to-back launches between two HIP events on the library's stream"},
                    "alg_bytes_per_launch": int(alg_bytes), "source_points": M, "kbar": round(kbar, 3),
                    "flops_per_pair": FLOPS_PER_PAIR, "vector_f64": vector_f64(pairs, M, us_roof * 1e-3),
                    "note": "at this size one launch is latency bound (7.5 k points: 1.8 MB, ~15 MFLOP); both fractions are small by construction, "
                            "roofline_cfg5 / wide_map give the same kernel where the axes mean something (DESIGN.md)"}

    # ---- the same kernel at BASELINE.json configs[4] size (2,097,152-point source against a 2 M-point target): the size at
    # which the HBM roofline is a meaningful yardstick for it (at cfg2 size one launch moves < 2 MB).  Extra key, rank 0.
    def make_cfg5():
        """2 M-point target (17 scans in the map frame) and the 2,097,152-point 128-beam source of BASELINE.json configs[4]."""
        tgt5 = wl.submap_2m(poses, scans)
        src5, g5 = wl.cfg5_source(poses)
        p5 = np.zeros(6)
        capi.lib().lsa_matrix_to_pose(capi.fptr(capi.colmajor(g5)), capi.dptr(p5))
        return torch.from_numpy(tgt5).to(dev), torch.from_numpy(src5).to(dev), g5, p5

    with guarded("cfg5 probe"):
        progress("cfg5 probe")
        roofline_cfg5 = None
        if rank == 0 and world == 1 and not args.no_large:
            d_t5, d_s5, g5, p5 = make_cfg5()
            torch.cuda.synchronize()
            ndt5 = registration.NDTRegistrationHip(ctx, 1.0, 0.1, 0.0, 30, 0.55)
            ndt5.SetInputTarget(d_t5)
            ndt5.set_source(d_s5)
            (_, _, _, pairs5, npts5), ms5 = ndt5.derivatives_timed(g5, p5, kind=0, repeat=20)
            M5, kbar5 = int(npts5), pairs5 / max(1.0, float(npts5))
            alg5 = M5 * (16.0 + 4.0 * 27.0 + 48.0 * kbar5)
            roofline_cfg5 = {"workload": "cfg5: k_ndt_pass<0>, 2,097,152-point source vs 2 M-point target (res 1.0 m)", "bound": "hbm",
                             "achieved": round(alg5 / (ms5 * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(alg5 / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "us_per_launch": round(ms5 * 1e3, 1),
                             "alg_bytes_per_launch": int(alg5), "source_points": M5, "kbar": round(kbar5, 3),
                             "pairs_per_s": round(pairs5 / (ms5 * 1e-3), 0), "vector_f64": vector_f64(pairs5, M5, ms5)}
            # BASELINE.json configs[4] in full: the 30-iteration align of the 2 M-point scan (eps 0 -> runs to max_iter)
            ndt5.ScanMatch(d_s5, g5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ndt5.ScanMatch(d_s5, g5)
            torch.cuda.synchronize()
            roofline_cfg5["align_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
            roofline_cfg5["align_iterations"] = int(ndt5.last_stats["iterations"])
            roofline_cfg5["align_passes"] = int(ndt5.last_stats["passes"])
            ndt5.close()
            # ---- the two stages SURVEY.md 8d expects at the HBM roofline, at the same size: VoxelFilter (formula A: 16 N + 16 V bytes)
            # and the target voxelisation (formula B: 16 N_t + 48 V) of the 2 M-point cloud.  Whole calls (every kernel of the build +
            # the one host wait), median of 30 behind 10 warm-up calls; DESIGN.md has the per-kernel split (profiles/r05_*).
            with guarded("2 M-point build"):
                def med_ms(fn, reps=30, warm=10):
                    for _ in range(warm):
                        fn()
                    torch.cuda.synchronize()
                    ts = []
                    for _ in range(reps):
                        t0 = time.perf_counter()
                        fn()
                        torch.cuda.synchronize()
                        ts.append((time.perf_counter() - t0) * 1e3)
                    return float(np.median(ts)), float(np.min(ts)), float(np.max(ts))
                f5 = cloud_filter.VoxelFilterHip(ctx, 0.3, 0.3, 0.3)
                out5 = torch.empty((d_t5.shape[0], 4), dtype=torch.float32, device=dev)
                V5 = int(f5.filter_device(d_t5, out5).shape[0])
                fm, fmin, fmax = med_ms(lambda: f5.filter_device(d_t5, out5))
                ndt6 = registration.NDTRegistrationHip(ctx, 1.0, 0.1, 0.0, 30, 0.55)
                tm, tmin, tmax = med_ms(lambda: ndt6.SetInputTarget(d_t5))
                gi6 = ndt6.grid_info()
                N5 = int(d_t5.shape[0])
                algA, algB = 16.0 * N5 + 16.0 * V5, 16.0 * N5 + 48.0 * int(gi6["occupied"])
                roofline_cfg5["build"] = {
                    "what": "the 2 M-point target through VoxelFilter 0.3 m (SURVEY 8d formula A) and through SetInputTarget at 1.0 m (formula B): whole calls, "
                            "median of 30; bound hbm, peak %.0f GB/s" % HBM_PEAK_GBS,
                    "filter": {"ms": round(fm, 4), "min_ms": round(fmin, 4), "max_ms": round(fmax, 4), "points": N5, "voxels": V5, "alg_bytes": int(algA),
                               "achieved": round(algA / (fm * 1e-3) / 1e9, 1), "unit": "GB/s", "frac": round(algA / (fm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                    "set_input_target": {"ms": round(tm, 4), "min_ms": round(tmin, 4), "max_ms": round(tmax, 4), "points": N5, "voxels_occupied": int(gi6["occupied"]),
                                         "alg_bytes": int(algB), "achieved": round(algB / (tm * 1e-3) / 1e9, 1), "unit": "GB/s",
                                         "frac": round(algB / (tm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
                ndt6.close(); f5.close()
                del out5
            del d_t5, d_s5

    # ---- the same kernel against a target that cannot live in a 4 MiB per-XCD L2 (wide_map: >= 100 k valid voxels over
    # +-160 m, a 26 MB dense probe table + ~10 MB of records), cfg5-sized source in RANDOM order (adversarial) and sorted by
    # cell (what a sweeping sensor delivers): where the "table is cache resident" claim can fail.  Extra key, rank 0.
    with guarded("wide-map probe"):
        progress("wide-map probe")
        wide_probe = None
        if rank == 0 and world == 1 and not args.no_large:
            tgt_w, src_w, g_w = wl.wide_map()
            p_w = np.zeros(6)
            capi.lib().lsa_matrix_to_pose(capi.fptr(capi.colmajor(g_w)), capi.dptr(p_w))
            ndt_w = registration.NDTRegistrationHip(ctx, 1.0, 0.1, 0.0, 30, 0.55)
            ndt_w.SetInputTarget(torch.from_numpy(tgt_w).to(dev))
            gi_w = ndt_w.grid_info()
            wide_probe = {"workload": "wide_map: %d-point target over +-160 m, dense probe table %d x %d x %d cells (%.1f MB) + %.1f MB of records; 2,097,152-point source"
                                      % (tgt_w.shape[0], gi_w["dim"][0], gi_w["dim"][1], gi_w["dim"][2], float(np.prod(gi_w["dim"])) * 16 / 1e6, gi_w["occupied"] * 80 / 1e6),
                          "target_voxels_valid": int(gi_w["valid"])}
            order = np.lexsort((np.floor(src_w[:, 0]), np.floor(src_w[:, 1]), np.floor(src_w[:, 2])))
            for name, arr in (("random_order_unsorted", src_w), ("random_order", src_w), ("cell_sorted", np.ascontiguousarray(src_w[order]))):
                d_sw = torch.from_numpy(arr).to(dev)
                torch.cuda.synchronize()
                # "random_order": the library's default — it samples the cloud, finds it incoherent and walks a spatially ordered copy
                # (csrc/source_sort.hip); "random_order_unsorted": that switched off, the raw cost of an incoherent walk
                ndt_w.set_source_sort(0 if name == "random_order_unsorted" else 1)
                ndt_w.set_source(d_sw)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ndt_w.set_source(d_sw)
                torch.cuda.synchronize()
                order_ms = (time.perf_counter() - t0) * 1e3
                (_, _, _, pairs_w, npts_w), ms_w = ndt_w.derivatives_timed(g_w, p_w, kind=0, repeat=20)
                Mw, kbar_w = int(npts_w), pairs_w / max(1.0, float(npts_w))
                alg_w = Mw * (16.0 + 4.0 * 27.0 + 48.0 * kbar_w)
                wide_probe[name] = {"us_per_launch": round(ms_w * 1e3, 1), "kbar": round(kbar_w, 3), "pairs_per_s": round(pairs_w / (ms_w * 1e-3), 0),
                                    "hbm_alg": {"achieved": round(alg_w / (ms_w * 1e-3) / 1e9, 1), "unit": "GB/s", "frac": round(alg_w / (ms_w * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                                    "vector_f64": vector_f64(pairs_w, Mw, ms_w)}
                if name == "random_order":
                    wide_probe[name]["walked_in_spatial_order"] = bool(ndt_w.source_sort_count()[1])
                    wide_probe[name]["order_ms_once_per_scan_match"] = round(order_ms, 3)
                    wide_probe[name]["us_per_launch_incl_ordering_over_31_passes"] = round(ms_w * 1e3 + order_ms * 1e3 / 31.0, 1)
                del d_sw
            ndt_w.close()

    # ---- a second, KITTI-residential-shaped workload through the same pipeline (workloads.dense_dataset: front gardens, hedges,
    # trees with volumetric crowns, second-row buildings; 100 m range): brackets the headline's bare street.  Extra key, rank 0.
    def front_end_drive(lsa_fe, ctx, d_all, poses_drive, reps=5):
        """The drive through FrontEnd::Update twice over: scan by scan (one library call per scan; nothing is waited for between two
        scans, as front_end_node calls it), and as ONE run of lsa_front_end_update_batch_device (next scan's frame filter on a second
        stream).  Same poses, bit for bit, asserted here.  Every figure is the MEDIAN of `reps` whole drives behind a clock ramp (the
        section may start seconds of CPU work after the GPU's last kernel), with min / max and the registration's armed-pass counters
        beside it, so that a slow run explains itself."""
        import torch
        n = len(d_all)

        def one_by_one(fe):
            n_kf = 0
            for d_one in d_all:
                _, T = fe.Update(d_one)
                n_kf += fe.last_key_frame
            return T, n_kf

        def fresh():
            fe = lsa_fe.FrontEndHip(ctx)
            fe.SetInitPose(poses_drive[0].astype(np.float32))
            return fe
        # clock ramp + buffer sizing: whole drives, untimed, for ~0.4 s
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < min(args.ramp_seconds, 0.4):
            fe = fresh()
            fe.UpdateBatch(d_all, stats="key_frame")
            torch.cuda.synchronize()
            fe.close()
        fe = fresh(); one_by_one(fe); torch.cuda.synchronize(); fe.close()
        t_fe, t_b, armed_fe, armed_b = [], [], None, None
        for _rep in range(reps):
            fe = fresh()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            T_fe, n_kf = one_by_one(fe)
            torch.cuda.synchronize()
            t_fe.append((time.perf_counter() - t0) * 1e3 / n)
            cnt_fe, armed_fe = fe.map_counters(), fe.armed_stats()
            fe.close()
        for _rep in range(reps):
            fe = fresh()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            T_b, kf_b = fe.UpdateBatch(d_all, stats="key_frame")
            torch.cuda.synchronize()
            t_b.append((time.perf_counter() - t0) * 1e3 / n)
            armed_b = fe.armed_stats()
            fe.close()
        if not np.array_equal(T_b[-1], T_fe) or sum(kf_b) != n_kf:
            raise RuntimeError("front end: the batch run and the scan-by-scan run disagree")
        return {"scans": n, "key_frames": n_kf, "value": round(1e3 / float(np.median(t_fe)), 1), "unit": "scans/s",
                "ms_per_scan": round(float(np.median(t_fe)), 4), "ms_per_scan_spread": spread(t_fe), "armed_passes_last_drive": armed_fe,
                "one_run": {"value": round(1e3 / float(np.median(t_b)), 1), "unit": "scans/s", "ms_per_scan": round(float(np.median(t_b)), 4),
                            "ms_per_scan_spread": spread(t_b), "armed_passes_last_drive": armed_b,
                            "what": "the same drive as one lsa_front_end_update_batch_device call; final pose identical to the scan-by-scan run"},
                "how": "median of %d whole drives each way, after ~0.4 s of untimed drives (clock ramp, buffers sized)" % reps,
                "end_drift_m": round(float(np.linalg.norm(T_fe[:3, 3] - poses_drive[n - 1][:3, 3])), 4), "local_map": cnt_fe}

    def second_workload(label, poses_d, scans_d, keyframes, tests, with_oracle=False, with_front_end=False):
        """Another scene through the same Filter -> ScanMatch pipeline: local map of `keyframes` (filtered at 0.3 m), the `tests`
        scans with the constant-velocity guess + prediction error of the headline."""
        nt = len(tests)
        d_map_d = map_filter.filter_device(torch.from_numpy(wl.local_map(poses_d, scans_d, keyframes)).to(dev)).clone()
        ndt_d = registration.NDTRegistrationHip(ctx, **NDT_PARAMS)
        ndt_d.SetInputTarget(d_map_d)
        gi_d = ndt_d.grid_info()
        d_sc = [torch.from_numpy(np.ascontiguousarray(scans_d[k])).to(dev) for k in tests]
        g_list = [wl.guess(poses_d, k) for k in tests]
        g_d = np.stack([capi.colmajor(g) for g in g_list]).astype(np.float32)
        buf_d = [torch.empty((max(x.shape[0] for x in d_sc), 4), dtype=torch.float32, device=dev) for _ in range(2)]

        def run_dense(count):
            idx = [i % nt for i in range(count)]
            ptrs = (C.c_void_p * count)(*[d_sc[k].data_ptr() for k in idx])
            nn = (C.c_size_t * count)(*[d_sc[k].shape[0] for k in idx])
            pred = np.ascontiguousarray(g_d[idx])
            out = np.zeros((count, 16), dtype=np.float32)
            st = (capi.NdtStats * count)()
            nf = (C.c_size_t * count)()
            rc = L.lsa_pipeline_scan_match(hf, ndt_d.h, count, ptrs, nn, capi.fptr(pred), buf_d[0].data_ptr(), buf_d[1].data_ptr(), 0, 1, capi.fptr(out), st, nf)
            if rc:
                ctx_f.check(rc)
            return idx, out, st, nf
        # The headline's treatment (the driver's kitti_shaped figure was 40 % under the builder's with nothing on the line to
        # say why): a clock ramp first — this section starts after seconds of CPU-only dataset generation and oracle work, during which the
        # GPU falls back to its idle clocks —, then FIVE timed batches instead of one, the value their median, min / max beside it, and
        # the registration's counters (late rows cost 2 ms each, redone passes a launch) sampled over exactly the timed batches.
        k_d = max(200, args.steps // 4)
        n_ramp_d = int(min(args.ramp_seconds, 0.75) * 8000)
        if n_ramp_d:
            run_dense(n_ramp_d)
        run_dense(min(2000, k_d))
        ndt_d.armed_stats(reset=True)
        ndt_d.armed_timing(reset=True)
        dts = []
        for _b in range(5):
            ctx.synchronize(); ctx_f.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            idx, out_d, st_d, nf_d = run_dense(k_d)
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        armed_d, timing_d = ndt_d.armed_stats(), ndt_d.armed_timing()
        dt_d = float(np.median(dts))
        last = range(k_d - nt, k_d)
        e_d = np.array([pose_err(capi.from_colmajor(out_d[i]), wl.truth(poses_d, tests[idx[i]])) for i in last])
        key = {"workload": label, "value": round(k_d / dt_d, 1), "unit": "scans/s", "ms_per_step": round(dt_d / k_d * 1e3, 4), "steps": k_d,
               "batches": {"timed": len(dts), "steps_each": k_d, "ramp_steps": n_ramp_d, "ms_per_step": spread([d / k_d * 1e3 for d in dts]),
                           "ms_per_step_each": [round(d / k_d * 1e3, 4) for d in dts]},
               "armed_passes": armed_d,
               "armed_timing_us": {k_: round(v_, 3) for k_, v_ in timing_d.items()},
               "key_frames_in_local_map": len(keyframes),
               "raw_points": int(np.mean([x.shape[0] for x in d_sc])), "source_points": int(np.mean([nf_d[i] for i in last])),
               "target_points": int(gi_d["n_target"]), "target_voxels_occupied": int(gi_d["occupied"]), "target_voxels_valid": int(gi_d["valid"]),
               "kbar": round(float(np.mean([st_d[i].pairs_per_pass / max(1, nf_d[i]) for i in last])), 3),
               "iterations": float(np.mean([st_d[i].iterations for i in last])), "passes": float(np.mean([st_d[i].passes for i in last])),
               "pose_rmse_vs_truth": {"t_m": float(np.sqrt(np.mean(e_d[:, 0] ** 2))), "r_rad": float(np.sqrt(np.mean(e_d[:, 1] ** 2)))}}
        if with_oracle and not args.no_cpu_baseline:
            # the same nine scans through the CPU oracle (the checker): NDTMANUAL on each side's own table, and the PCL-semantics
            # mode with the oracle's table overwritten by the GPU's records (identical tables: states the iteration's parity on
            # ALL nine scans, the ones its non-contractive iteration does not settle on included)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py
            if not (_under_profiler() and not os.path.exists(os.path.join(ROOT, "oracle", "liblsa_oracle_fast.so"))):
                oracle_py.build(ref=False)
                o_ = oracle_py.Oracle(fast=True)
                map8 = synth.to_xyzi32(d_map_d.cpu().numpy())
                on = oracle_py.OracleNDT(fast=True, **NDT_PARAMS)
                on.set_input_target(map8)
                ndt_p = registration.NDTRegistrationHip(ctx, pcl_mode=True, **NDT_PARAMS)
                ndt_p.SetInputTarget(d_map_d)
                gi_p, gr_p = ndt_p.grid_dump()
                op = oracle_py.OracleNDT(fast=True, pcl_mode=True, **NDT_PARAMS)
                op.set_input_target(map8)
                op.override_cells(gi_p, gr_p)
                e_n, e_p, capped = [], [], 0
                res_last = {idx[i]: capi.from_colmajor(out_d[i]) for i in last}
                for j, k in enumerate(tests):
                    filt, _ = o_.voxel_filter(synth.to_xyzi32(scans_d[k]), 0.5)
                    f8 = synth.to_xyzi32(filt)
                    Tref, _st = on.scan_match(f8, g_list[j])
                    e_n.append(pose_err(res_last[j], Tref))
                    Tp_ref, st_p = op.scan_match(f8, g_list[j])
                    _, Tp = ndt_p.ScanMatch(f8, g_list[j])
                    capped += int(st_p["iterations"] > NDT_PARAMS["max_iter"])
                    e_p.append(pose_err(Tp, Tp_ref))
                e_n, e_p = np.array(e_n), np.array(e_p)
                key["pose_rmse_vs_oracle"] = {"t_m": float(np.sqrt(np.mean(e_n[:, 0] ** 2))), "r_rad": float(np.sqrt(np.mean(e_n[:, 1] ** 2))), "scans": len(e_n)}
                key["pose_rmse_pcl_mode_vs_oracle_identical_tables"] = {"t_m": float(np.sqrt(np.mean(e_p[:, 0] ** 2))), "r_rad": float(np.sqrt(np.mean(e_p[:, 1] ** 2))),
                                                                        "scans": len(e_p), "scans_at_iteration_cap": capped}
                ndt_p.close()
        if with_front_end:
            from lidar_slam_amd import front_end as lsa_fe
            d_all = [torch.from_numpy(np.ascontiguousarray(sc)).to(dev) for sc in scans_d]
            torch.cuda.synchronize()
            key["front_end_loop"] = front_end_drive(lsa_fe, ctx, d_all, poses_d)
            del d_all
        ndt_d.close()
        del d_sc, d_map_d
        return key

    with guarded("dense workload"):
        progress("dense workload")
        dense_key = None
        kitti_key = None
        if rank == 0 and world == 1 and not dist_mode and not args.no_extra:
            poses_d, scans_d = wl.dense_dataset(workers)
            dense_key = second_workload("dense: residential street (gardens, hedges, trees, second-row buildings; 100 m range) through the same Filter -> ScanMatch pipeline",
                                        poses_d, scans_d, MAP_KEYFRAMES, TEST_SCANS)
            del poses_d, scans_d
            progress("KITTI-shaped workload")
            poses_k, scans_k = wl.kitti_dataset(workers)
            kitti_key = second_workload("kitti_shaped: residential street on rolling terrain with vegetation and distant blocks, HDL-64E beam table, 120 m range, 50-pose drive; "
                                        "local map = the reference's full window of 20 key-frames (front_end.yaml local_frame_num), filtered at 0.3 m",
                                        poses_k, scans_k, wl.KITTI_MAP_KEYFRAMES, wl.KITTI_TEST_SCANS, with_oracle=True, with_front_end=True)
            del poses_k, scans_k

    # ---- N > 1: the exchange step at the size it is meant for (BASELINE.json configs[3] / [4]): the 2,097,152-point source
    # sharded over the ranks against the replicated 2 M-point target, ONE all-reduce of the 32-double block per pass.
    # Whole passes (kernel + second stage + collective + result on the host), wall clock between barriers, max over ranks.
    # Every rank executes exactly the same sequence of collectives.  Extra key; the headline stays the cfg2 metric.
    progress("sharded cfg5 / cfg4")
    cfg5_sharded = None
    cfg4_sharded = None
    cfg3_sharded = None
    exchange_info = None
    if dist_mode and world > 1 and not args.no_large:
        tgt5 = wl.submap_2m(poses, scans)
        d_t5 = torch.from_numpy(tgt5).to(dev)
        ensure_exchange()
        if exchange["native"] is not None:
            exchange_info = {"backend": "RCCL called by the library (lsa_rccl_*: ncclAllReduce via dlopen)", "comm_ranks": exchange["native"].comm_count()}
        elif one_gpu:
            exchange_info = {"backend": "gloo staged through the host (one-GPU rehearsal: RCCL refuses two ranks on one device)", "comm_ranks": world}
        else:
            exchange_info = {"backend": "torch.distributed nccl (= RCCL) process group", "comm_ranks": dist.get_world_size()}
            if exchange.get("fallback"):
                exchange_info["note"] = exchange["fallback"]

        def sharded(src_np, guess, label):
            """BASELINE.json configs[3] / [4]: the source sharded over the ranks against the replicated 2 M-point target, one
            all-reduce per pass: wall-clock per whole pass and for the whole align (max over ranks, between barriers)."""
            n_src = int(src_np.shape[0])
            lo, hi = lsa_dist.shard_bounds(n_src, rank, world)
            d_sh = torch.from_numpy(np.ascontiguousarray(src_np[lo:hi])).to(dev)
            p_s = np.zeros(6)
            capi.lib().lsa_matrix_to_pose(capi.fptr(capi.colmajor(guess)), capi.dptr(p_s))
            nd = registration.NDTRegistrationHip(ctx, 1.0, 0.1, 0.0, 30, 0.55)   # max_iter 30, eps 0: runs to the iteration cap
            nd.SetInputTarget(d_t5)
            nd.set_source(d_sh)
            if exchange["native"] is not None:
                exchange["native"].attach(nd)
            else:
                nd.set_allreduce(exchange["cb"])
            kp = 20
            for _ in range(3):
                o5 = nd.derivatives(guess, p_s, 0)
            sync_all()
            t0 = time.perf_counter()
            for _ in range(kp):
                o5 = nd.derivatives(guess, p_s, 0)
            sync_all()
            t_pass = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(t_pass, op=dist.ReduceOp.MAX)
            nd.ScanMatch(d_sh, guess)
            sync_all()
            t0 = time.perf_counter()
            nd.ScanMatch(d_sh, guess)
            sync_all()
            t_al = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(t_al, op=dist.ReduceOp.MAX)
            st_al = dict(nd.last_stats)
            nd.close()
            if rank != 0:
                return None
            dtp = float(t_pass.item()) / kp
            return {"workload": label, "ranks": world, "source_points": float(o5[4]), "source_points_per_rank": int(hi - lo), "pairs": float(o5[3]),
                    "us_per_pass": round(dtp * 1e6, 1), "pairs_per_s": round(float(o5[3]) / dtp, 0),
                    "align_ms": round(float(t_al.item()) * 1e3, 3), "align_passes": int(st_al["passes"]), "align_iterations": int(st_al["iterations"])}

        src5, g5 = wl.cfg5_source(poses)
        cfg5_sharded = sharded(src5, g5, "cfg5 (BASELINE.json configs[4]): 2,097,152-point source sharded over the ranks, 2 M-point target replicated, one all-reduce per pass, 30 iterations")
        del src5
        src4, g4 = wl.cfg4_source(poses)
        cfg4_sharded = sharded(src4, g4, "cfg4 (BASELINE.json configs[3]): raw ~132 k-point scan sharded over the ranks, 2 M-point target replicated, one all-reduce per pass")
        del d_t5

        def sharded_icp(src_np, guess, mode, label):
            """BASELINE.json configs[2] over the exchange step: the source sharded over the ranks against the replicated local map,
            one all-reduce of the iteration's 32-double block (B, H, pairs) per iteration, GetFitnessScore's 2-double sum at the end."""
            lo, hi = lsa_dist.shard_bounds(int(src_np.shape[0]), rank, world)
            d_sh = torch.from_numpy(np.ascontiguousarray(src_np[lo:hi])).to(dev)
            ic = registration.ICPRegistrationHip(ctx, 2.0, 30, mode=mode)
            ic.SetInputTarget(d_map_f.contiguous())
            if exchange["native"] is not None:
                exchange["native"].attach_icp(ic)
            else:
                ic.set_allreduce(exchange["cb"])
            ic.ScanMatch(d_sh, guess)
            sync_all()
            t0 = time.perf_counter()
            _, T_ic = ic.ScanMatch(d_sh, guess)
            sync_all()
            t_al = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(t_al, op=dist.ReduceOp.MAX)
            st_ic = dict(ic.last_stats)
            fit = float(ic.GetFitnessScore())
            Tt = torch.from_numpy(np.ascontiguousarray(T_ic, dtype=np.float32).copy())
            same = [torch.zeros_like(Tt) for _ in range(world)]
            dist.all_gather(same, Tt)
            identical = all(torch.equal(x, same[0]) for x in same)     # every rank took the same solve from the summed block
            ic.close()
            if rank != 0:
                return None
            al = float(t_al.item())
            return {"workload": label, "ranks": world, "source_points": int(src_np.shape[0]), "source_points_per_rank": int(hi - lo),
                    "align_ms": round(al * 1e3, 3), "iterations": int(st_ic["iterations"]), "us_per_iteration": round(al * 1e6 / max(1, int(st_ic["iterations"])), 1),
                    "correspondences": int(st_ic["correspondences"]), "fitness": fit, "poses_identical_across_ranks": bool(identical)}

        cfg3_sharded = {"plane": sharded_icp(src4, g4, capi.ICP_MODE_PLANE, "cfg3 (BASELINE.json configs[2]) at raw-scan size: point-to-plane ICP, ~132 k-point scan sharded over the ranks, local map replicated, one all-reduce per iteration"),
                        "icpmanual": sharded_icp(src4, g4, 0, "ICPMANUAL (the reference's plugin), same sharding")}
        if rank != 0:
            cfg3_sharded = None
        del src4

    # ---- capacity of one GPU beyond a single sensor stream: S independent scan streams (own contexts / HIP streams / filter /
    # registration handles / voxel table each, one host thread per stream in lsa_pipeline_scan_match) share the GPU.  A
    # KITTI-sized pass occupies ~120 of 256 CUs for 8 us and the stream is bound by launch / completion latency, so
    # independent streams (several sensors, loop-closure candidates, offline re-mapping) interleave.  A SINGLE sensor stream
    # cannot use this — scan k+1's prediction needs scan k's pose (front_end.cpp:238-240) — so it is an extra key, never `value`.
    with guarded("concurrent streams"):
        progress("concurrent streams")
        concurrent = None
        if rank == 0 and world == 1 and not dist_mode and args.streams > 1:
            import threading
            S = args.streams

            class _Pipe:
                def __init__(self, sid):
                    self.sid = sid
                    self.ctx, self.ctx_f = capi.Context(dev_index, None), capi.Context(dev_index, None)
                    self.ff = cloud_filter.VoxelFilterHip(self.ctx_f, 0.5, 0.5, 0.5)
                    self.ndt = registration.NDTRegistrationHip(self.ctx, **NDT_PARAMS)
                    self.ndt.SetInputTarget(d_map_f.contiguous())
                    self.ctx.synchronize()
                    self.buf = [torch.empty_like(d_filt2[0]) for _ in range(2)]

                def prep(self, count):
                    idx = [(i + self.sid) % ns for i in range(count)]
                    self.idx, self.count = idx, count
                    self.ptrs = (C.c_void_p * count)(*[d_scans[k].data_ptr() for k in idx])
                    self.nn = (C.c_size_t * count)(*[d_scans[k].shape[0] for k in idx])
                    self.pred = np.ascontiguousarray(guess_cm[idx])
                    self.out = np.zeros((count, 16), dtype=np.float32)
                    self.st = (capi.NdtStats * count)()
                    self.nf = (C.c_size_t * count)()

                def run(self):
                    self.rc = L.lsa_pipeline_scan_match(self.ff.h, self.ndt.h, self.count, self.ptrs, self.nn, capi.fptr(self.pred), self.buf[0].data_ptr(),
                                                        self.buf[1].data_ptr(), 0, 1, capi.fptr(self.out), self.st, self.nf)

                def close(self):
                    self.ndt.close(); self.ff.close(); self.ctx_f.close(); self.ctx.close()

            pipes = [_Pipe(sid) for sid in range(S)]

            def go(count):
                for p_ in pipes:
                    p_.prep(count)
                bar = threading.Barrier(S + 1)

                def work(p_):
                    bar.wait(); p_.run(); bar.wait()
                th = [threading.Thread(target=work, args=(p_,)) for p_ in pipes]
                for t_ in th:
                    t_.start()
                torch.cuda.synchronize()
                bar.wait(); t0 = time.perf_counter(); bar.wait()
                dt_ = time.perf_counter() - t0
                for t_ in th:
                    t_.join()
                for p_ in pipes:
                    if p_.rc:
                        p_.ctx_f.check(p_.rc)
                return dt_
            go(int(min(args.ramp_seconds, 0.5) * 8000)); go(max(1, args.warmup))
            dt_s = go(args.steps)
            # every stream must reproduce, bit for bit, what the single stream produced for the same scans
            same = all(np.array_equal(capi.from_colmajor(p_.out[i]), results[p_.idx[i]][0]) for p_ in pipes for i in range(max(0, p_.count - ns), p_.count)
                       if p_.idx[i] in results)
            concurrent = {"streams": S, "steps_per_stream": args.steps, "value": round(S * args.steps / dt_s, 2), "unit": "scans/s",
                          "ms_per_step_per_stream": round(dt_s / args.steps * 1e3, 4), "poses_identical_to_single_stream": bool(same),
                          "note": "S independent scan streams on one GPU, one host thread + one helper thread each; not applicable to a single sensor stream"}
            for p_ in pipes:
                p_.close()

    # ---- the caller of the hot path, end to end: lsa_front_end_* = FrontEnd::Update with the sliding local
    # map kept in HBM (front_end.cpp:88-424) over the 30 scans of the drive, scan after scan with NO overlap between scans (the
    # prediction of scan k+1 needs the pose of scan k): frame filter + ScanMatch every scan, and on every key-frame (2 m) the
    # local-map rebuild, its 0.3 m filter and the voxel table.  Extra key, rank 0.
    with guarded("front-end loop"):
        progress("front-end loop")
        front_end_loop = None
        if rank == 0 and world == 1 and not dist_mode and not args.no_extra:
            from lidar_slam_amd import front_end as lsa_fe
            d_all = [torch.from_numpy(np.ascontiguousarray(sc)).to(dev) for sc in scans]
            torch.cuda.synchronize()
            front_end_loop = front_end_drive(lsa_fe, ctx, d_all, poses)
            front_end_loop["note"] = "device-resident FrontEnd::Update loop incl. key-frame work (local map of <= 20 key-frames, 0.3 m map filter, voxel table)"
            # the same drive from HOST clouds (PointXYZI rows of 32 bytes, what front_end.cpp's caller holds): lsa_front_end_update, one
            # call per scan — the PCIe-inclusive figure of the whole FrontEnd::Update drop-in (never `value`)
            h_all = [synth.to_xyzi32(sc) for sc in scans]
            t_h = []
            for _rep in range(4):
                fe_h = lsa_fe.FrontEndHip(ctx)
                fe_h.SetInitPose(poses[0].astype(np.float32))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for hc in h_all:
                    _, T_h = fe_h.Update(hc)
                torch.cuda.synchronize()
                t_h.append((time.perf_counter() - t0) * 1e3 / len(h_all))
                ffc = fe_h.frame_filter_counters()
                fe_h.close()
            t_hb = []
            for _rep in range(4):   # ... and as ONE run of host clouds (lsa_front_end_update_batch: scans k + 1, k + 2 packed and on the link while scan k is registered)
                fe_h = lsa_fe.FrontEndHip(ctx)
                fe_h.SetInitPose(poses[0].astype(np.float32))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                T_hb, _kf = fe_h.UpdateBatch(h_all, stats="key_frame")
                torch.cuda.synchronize()
                t_hb.append((time.perf_counter() - t0) * 1e3 / len(h_all))
                fe_h.close()
            if not np.array_equal(T_hb[-1], T_h):
                raise RuntimeError("front end: the host-cloud run and the host-cloud scan-by-scan drive disagree")
            front_end_loop["host_clouds_one_run"] = {"value": round(1e3 / float(np.median(t_hb[1:])), 1), "unit": "scans/s", "ms_per_scan": round(float(np.median(t_hb[1:])), 4),
                                                     "ms_per_scan_spread": spread(t_hb[1:]),
                                                     "what": "the same host clouds as one lsa_front_end_update_batch call; final pose identical to the scan-by-scan host drive"}
            front_end_loop["host_clouds"] = {"value": round(1e3 / float(np.median(t_h[1:])), 1), "unit": "scans/s", "ms_per_scan": round(float(np.median(t_h[1:])), 4),
                                             "ms_per_scan_spread": spread(t_h[1:]), "scans_pulled_by_the_device": int(ffc["pulled"]), "of": len(h_all),
                                             "what": "lsa_front_end_update on 32-byte PointXYZI host rows, one call per scan (median of 3 drives behind one untimed): pack + PCIe inside"}
            del d_all, h_all

    # ---- BASELINE.json configs[2]: point-to-plane ICP (the second RegistrationInterface plugin, lsa_icp_* in its point-to-plane
    # mode; 30 iterations, max correspondence distance 2 m, normals from the target's 1 m voxels) on the same scans against the
    # same local map.  Extra key, rank 0.
    with guarded("point-to-plane ICP"):
        progress("point-to-plane ICP")
        icp_plane = None
        if rank == 0 and world == 1 and not dist_mode and not args.no_extra:
            icp = registration.ICPRegistrationHip(ctx, 2.0, 30, mode=capi.ICP_MODE_PLANE)
            icp.SetInputTarget(d_map_f.contiguous())
            t_icp, e_icp, e_icp_orc = [], [], []
            icp_orc = None
            if not args.no_cpu_baseline:   # the oracle's point-to-plane restatement on the same scans (the checker, not timed)
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle_py as icp_orc
                if _under_profiler() and not os.path.exists(os.path.join(ROOT, "oracle", "liblsa_oracle.so")):
                    icp_orc = None
                else:
                    icp_orc.build(ref=False)
                    map8 = synth.to_xyzi32(d_map_f.cpu().numpy())
            icp_m = registration.ICPRegistrationHip(ctx, 2.0, 30)   # ICPMANUAL (the reference's own plugin) on the same scans: timing only here,
            icp_m.SetInputTarget(d_map_f.contiguous())             # its parity is tests/test_gpu_bench_scans.py::test_cfg3_bench_scans_match_oracle
            t_icp_m, searches, queries = [], 0, 0
            for k in range(len(d_scans)):
                f_k = frame_filter.filter_device(d_scans[k]).clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, T_icp = icp.ScanMatch(f_k, guesses[k])
                t_icp.append(time.perf_counter() - t0)
                sc = icp.search_counts()
                searches += sc[0]; queries += sc[1]
                e_icp.append(pose_err(T_icp, truths[k]))
                t0 = time.perf_counter()
                icp_m.ScanMatch(f_k, guesses[k])
                t_icp_m.append(time.perf_counter() - t0)
                if icp_orc is not None:
                    T_o, _ = icp_orc.icp_plane(map8, synth.to_xyzi32(f_k.cpu().numpy()), guesses[k], 2.0, 30, 1.0)
                    e_icp_orc.append(pose_err(T_icp, T_o))
            e_icp = np.array(e_icp)
            icp_plane = {"workload": "cfg3: point-to-plane ICP, 30 iterations, same scans / local map as the headline", "scans": len(t_icp),
                         "ms_per_scan_match": round(float(np.median(t_icp)) * 1e3, 4), "value": round(1.0 / float(np.median(t_icp)), 1), "unit": "scans/s",
                         "pose_rmse_vs_truth": {"t_m": float(np.sqrt(np.mean(e_icp[:, 0] ** 2))), "r_rad": float(np.sqrt(np.mean(e_icp[:, 1] ** 2)))}}
            if e_icp_orc:
                eo = np.array(e_icp_orc)
                icp_plane["pose_rmse_vs_oracle"] = {"t_m": float(np.sqrt(np.mean(eo[:, 0] ** 2))), "r_rad": float(np.sqrt(np.mean(eo[:, 1] ** 2))), "scans": len(eo)}
            ap, af = icp.armed_counts()
            icp_plane["neighbour_searches"] = {"run": searches, "of": queries, "armed_iterations": ap, "armed_fallbacks": af,
                                               "note": "a query whose old neighbour is provably still the nearest does not search (csrc/icp.hip); same bits as with every search run"}
            icp_plane["icpmanual_ms_per_scan_match"] = round(float(np.median(t_icp_m)) * 1e3, 4)
            icp.close(); icp_m.close()

    # ---- CPU baseline: the oracle (single thread) on a bounded sample of the same scans; also the pose check
    with guarded("CPU baseline + pose checks"):
        progress("CPU baseline + pose checks")
        cpu_baseline = None
        pose_rmse = None
        pose_rmse_pcl = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py
            if _under_profiler() and not os.path.exists(os.path.join(ROOT, "oracle", "liblsa_oracle_fast.so")):
                raise SystemExit("bench.py under a profiler: build the oracle first (python -c 'import __graft_entry__ as g; g.build()') "
                                 "or pass --no-cpu-baseline; it is not built from a process whose GPU is already initialised")
            oracle_py.build(ref=False)
            o = oracle_py.Oracle(fast=True)
            ondt = oracle_py.OracleNDT(fast=True, **NDT_PARAMS)
            ondt.set_input_target(synth.to_xyzi32(d_map_f.cpu().numpy()))
            n_done, t_cpu, errs = 0, 0.0, []
            i = 0
            while t_cpu < args.cpu_seconds and n_done < 5000:
                k = i % len(TEST_SCANS)
                raw = synth.to_xyzi32(scans[TEST_SCANS[k]])
                t0 = time.perf_counter()
                filt, _ = o.voxel_filter(raw, 0.5)
                Tref, st = ondt.scan_match(synth.to_xyzi32(filt), guesses[k])
                t_cpu += time.perf_counter() - t0
                n_done += 1
                if k in results and i < len(TEST_SCANS):
                    errs.append(pose_err(results[k][0], Tref))
                i += 1
            cpu_baseline = {"value": round(n_done / t_cpu, 3), "unit": "scans/s", "cores": 1, "kind": "port",
                            "sample": "%d scans (VoxelFilter 0.5 + NDT align) of the same workload, oracle -O3, 1 thread, %.1f s" % (n_done, t_cpu)}
            if errs:
                e = np.array(errs)
                pose_rmse = {"t_m": float(np.sqrt(np.mean(e[:, 0] ** 2))), "r_rad": float(np.sqrt(np.mean(e[:, 1] ** 2))), "scans": len(errs)}
            # the metric says "RMSE vs PCL": PCL itself is absent, so the closest statement available is the
            # PCL-SEMANTICS mode of the HIP path (the "NDT" plugin's arithmetic) against the oracle's restatement of it, on the
            # same scans (both unpinned against real PCL)
            ndt_p = registration.NDTRegistrationHip(ctx, pcl_mode=True, **NDT_PARAMS)
            ndt_p.SetInputTarget(d_map_f.contiguous())
            ondt_p = oracle_py.OracleNDT(fast=True, pcl_mode=True, **NDT_PARAMS)
            ondt_p.set_input_target(synth.to_xyzi32(d_map_f.cpu().numpy()))
            errs_p, capped = [], []
            for k in range(len(TEST_SCANS)):
                filt, _ = o.voxel_filter(synth.to_xyzi32(scans[TEST_SCANS[k]]), 0.5)
                f8 = synth.to_xyzi32(filt)
                Tref, st_ref = ondt_p.scan_match(f8, guesses[k])
                _, Tg = ndt_p.ScanMatch(f8, guesses[k])
                hit_cap = st_ref["iterations"] > NDT_PARAMS["max_iter"] or ndt_p.last_stats["iterations"] > NDT_PARAMS["max_iter"]
                (capped if hit_cap else errs_p).append(pose_err(Tg, Tref))
            ndt_p.close()
            e = np.array(errs_p) if errs_p else np.zeros((0, 2))
            c = np.array(capped) if capped else np.zeros((0, 2))
            pose_rmse_pcl = {"t_m": float(np.sqrt(np.mean(e[:, 0] ** 2))) if len(e) else None,
                             "r_rad": float(np.sqrt(np.mean(e[:, 1] ** 2))) if len(e) else None, "scans": len(errs_p),
                             "scans_at_iteration_cap": len(capped), "max_t_m_at_iteration_cap": float(c[:, 0].max()) if len(c) else None,
                             "note": "HIP PCL-semantics mode vs the oracle's PCL-semantics restatement, each on its OWN voxel table (real PCL is not "
                                     "available). Cause of the capped scans (pinned by tests/test_gpu_bench_scans.py): PCL's zero-seeded covariances "
                                     "make this map's voxels 3 cm thin, the 8 cm prediction error starts outside the Newton basin, the Hessian is "
                                     "indefinite along the path and every step is the clamped 0.1 (the oracle alone behaves the same); such an "
                                     "iteration is non-contractive, so the 1e-12-level pose gap the two covariance formulas leave after the first "
                                     "step grows, via one-ulp flips of f32-transformed points, to centimetres. With IDENTICAL tables HIP and oracle "
                                     "agree to 1e-14 on every pass of every scan, capped ones included"}

    if rank == 0:
        st_list = [results[k][1] for k in sorted(results)]
        gt_err = [pose_err(results[k][0], truths[k]) for k in sorted(results)] if world == 1 else []
        line = {
            "metric": "KITTI HDL-64E scans/sec at converged NDT; final-pose RMSE vs PCL",
            "value": round(args.steps * (1 if shard_policy else world) / elapsed, 2),
            "unit": "scans/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
            "ramp_steps": int(args.ramp_seconds * 8000),
            "value_survey_def": (round(1e3 / (split["filter_ms"] + split["align_ms"]), 2) if "align_ms" in split else None),
            "higher_is_better": True,
            "scaling": "strong" if shard_policy else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "cfg2: synthetic HDL-64E scan (~%dk pts) -> VoxelFilter 0.5 m -> NDT scan-to-local-map (11 key-frames filtered 0.3 m), res 1.0 m"
                            % (int(np.mean([s.shape[0] for s in d_scans])) // 1000),
                "raw_points": int(np.mean([s.shape[0] for s in d_scans])),
                "source_points": int(np.mean([results[k][2] for k in results])),
                "target_points": int(ginfo["n_target"]),
                "target_voxels_valid": int(ginfo["valid"]),
                "iterations": float(np.mean([s["iterations"] for s in st_list])),
                "passes": float(np.mean([s["passes"] for s in st_list])),
                "converged": int(min(s["converged"] for s in st_list)),
                "set_target_ms": round(set_target_ms, 3), "set_target_ms_spread": spread(st_ms),
                "armed_passes": armed_after_headline,
                "parallelism": "1 GPU" if world == 1 else (
                    "every scan's source sharded x%d, one 256-B RCCL all-reduce per pass (strong scaling)" % world if shard_policy else
                    "%d independent scan streams, one per GPU, no data-path collective (weak scaling): %d source points per scan "
                    "are below the sharding break-even of dist.shard_pays; the sharded path is timed in multi_gpu_alt" % (world, m_probe)),
                **split,
            },
            "roofline": roofline,
            "roofline_cfg5": roofline_cfg5,
            "wide_map": wide_probe,
            "dense_workload": dense_key,
            "kitti_shaped": kitti_key,
            "cfg5_sharded": cfg5_sharded,
            "cfg4_sharded": cfg4_sharded,
            "cfg3_sharded": cfg3_sharded,
            "exchange": exchange_info,
            "multi_gpu_alt": alt,
            "concurrent_streams": concurrent,
            "front_end_loop": front_end_loop,
            "icp_point_to_plane": icp_plane,
            "cpu_baseline": cpu_baseline,
            "pose_rmse_vs_oracle": pose_rmse,
            "pose_rmse_pcl_mode_vs_oracle": pose_rmse_pcl,
        }
        if gt_err:
            ge = np.array(gt_err)
            line["pose_rmse_vs_truth"] = {"t_m": float(np.sqrt(np.mean(ge[:, 0] ** 2))), "r_rad": float(np.sqrt(np.mean(ge[:, 1] ** 2)))}
        print(json.dumps(line), file=real_stdout, flush=True)

    ndt.close(); frame_filter.close(); map_filter.close(); ctx_f.close()
    if exchange["native"] is not None:
        exchange["native"].close()
    if dist_mode:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
