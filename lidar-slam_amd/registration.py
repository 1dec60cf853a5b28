"""Host-side mirror of the reference's registration plugin layer over the C-ABI.

Reference: lidar_localization/include/lidar_localization/models/registration/registration_interface.hpp:14-24
(RegistrationInterface: SetInputTarget / ScanMatch / GetFitnessScore) and
src/models/registration/ndt_registration_manual.cpp:12-78 (NDTRegistrationManual).
Same names and argument meaning; ScanMatch returns (True, result_pose) — the reference returns true and writes
result_pose; result_cloud_ptr is never written by NDTRegistrationManual and is omitted.
Poses are 4x4 float32 numpy arrays (row-major view of Eigen::Matrix4f).
"""
import ctypes as C

import numpy as np

from . import capi


class RegistrationInterface:
    def SetInputTarget(self, input_target):  # noqa: N802
        raise NotImplementedError

    def ScanMatch(self, input_source, predict_pose):  # noqa: N802
        raise NotImplementedError

    def GetFitnessScore(self):  # noqa: N802
        raise NotImplementedError


class NDTRegistrationHip(RegistrationInterface):
    """NDTRegistrationManual(res, step_size, trans_eps, max_iter, outlier_ratio) or (node) with the YAML keys
    of config/mapping/front_end.yaml:24-29."""

    def __init__(self, ctx, res=1.0, step_size=0.1, trans_eps=0.01, max_iter=30, outlier_ratio=0.55, node=None, pcl_mode=False):
        if node is not None:
            res, step_size, trans_eps = float(node["res"]), float(node["step_size"]), float(node["trans_eps"])
            max_iter = int(node["max_iter"])
            outlier_ratio = float(node["outlier_ratio"]) if "outlier_ratio" in node else 0.55  # the "NDT" node has none
        self.ctx = ctx
        self.params = (res, step_size, trans_eps, max_iter, outlier_ratio)
        h = C.c_void_p()
        ctx.check(capi.lib().lsa_ndt_create(C.byref(h), ctx.h, res, step_size, trans_eps, max_iter, outlier_ratio))
        self.h = h
        if pcl_mode:  # stand in for the "NDT" plugin (ndt_registration.cpp, PCL semantics) instead of "NDTMANUAL"
            ctx.check(capi.lib().lsa_ndt_set_mode(h, capi.NDT_MODE_PCL))
        self._ls_struct, self._ls_dict = None, None
        self._cb = None
        self._keep = None

    # ---- RegistrationInterface -------------------------------------------------------------
    def SetInputTarget(self, input_target):  # noqa: N802
        if _is_tensor(input_target):
            check_points_tensor(input_target, "SetInputTarget")
            self._keep = input_target
            self.ctx.check(capi.lib().lsa_ndt_set_input_target_device(self.h, input_target.data_ptr(), input_target.shape[0]))
        else:
            pts = np.ascontiguousarray(input_target, dtype=np.float32)
            self.ctx.check(capi.lib().lsa_ndt_set_input_target(self.h, pts.ctypes.data, pts.shape[0], pts.strides[0], capi.w_offset(pts.shape[1])))
        return True

    def ScanMatch(self, input_source, predict_pose, trace=False):  # noqa: N802
        """trace=True additionally records one row per derivative pass ({p(6), score, step length, kind}) in
        last_stats["trace"] (lsa_ndt_set_trace; diagnostics for the parity tests)."""
        guess = capi.colmajor(predict_pose)
        out = getattr(self, "_out", None)
        if out is None:   # one result buffer for the object's life (a fresh array and its ctypes view per call cost ~5 us of a 45 us ScanMatch)
            out = self._out = np.zeros(16, dtype=np.float32)
        st = capi.NdtStats()
        tr, ntr = None, C.c_int(0)
        if trace:
            tr = np.zeros((512, 9))
            self.ctx.check(capi.lib().lsa_ndt_set_trace(self.h, capi.dptr(tr), 512, C.byref(ntr)))
        try:
            rc = self._scan_match(input_source, guess, out, st)
        finally:
            if trace:
                capi.lib().lsa_ndt_set_trace(self.h, None, 0, None)
        self.ctx.check(rc)
        self._ls_struct, self._ls_dict = st, None   # (the dict is built when somebody asks for it)
        if trace:
            self.last_stats["trace"] = tr[: min(ntr.value, 512)].copy()
        return True, capi.from_colmajor(out)

    @property
    def last_stats(self):
        """lsa_ndt_stats of the last ScanMatch as a dict (None before the first)"""
        if self._ls_dict is None and self._ls_struct is not None:
            self._ls_dict = self._ls_struct.as_dict()
        return self._ls_dict

    @last_stats.setter
    def last_stats(self, value):
        self._ls_struct, self._ls_dict = None, value

    def _scan_match(self, input_source, guess, out, st):
        if _is_tensor(input_source):
            check_points_tensor(input_source, "ScanMatch")
            self._src_keep = input_source
            rc = capi.lib().lsa_ndt_scan_match_device(self.h, input_source.data_ptr(), input_source.shape[0], capi.fptr(guess), capi.fptr(out), C.byref(st))
        else:
            pts = np.ascontiguousarray(input_source, dtype=np.float32)
            rc = capi.lib().lsa_ndt_scan_match(self.h, pts.ctypes.data, pts.shape[0], pts.strides[0], capi.fptr(guess), capi.fptr(out), C.byref(st))
        return rc

    def GetFitnessScore(self):  # noqa: N802
        v = C.c_float(0)
        self.ctx.check(capi.lib().lsa_ndt_get_fitness_score(self.h, C.byref(v)))
        return v.value

    # ---- introspection (parity tests / bench) ---------------------------------------------------
    def grid_info(self):
        info = np.zeros(9, dtype=np.int64)
        self.ctx.check(capi.lib().lsa_ndt_grid_info(self.h, info.ctypes.data_as(C.POINTER(C.c_int64))))
        return dict(dim=info[0:3].copy(), real_min=info[3:6].copy(), occupied=int(info[6]), valid=int(info[7]), n_target=int(info[8]))

    def grid_dump(self):
        n = C.c_size_t(0)
        self.ctx.check(capi.lib().lsa_ndt_grid_dump(self.h, None, None, 0, C.byref(n)))
        ints = np.zeros((n.value, 5), dtype=np.int32)
        rec = np.zeros((n.value, 10), dtype=np.float64)
        self.ctx.check(capi.lib().lsa_ndt_grid_dump(self.h, ints.ctypes.data_as(C.POINTER(C.c_int32)), capi.dptr(rec), n.value, C.byref(n)))
        return ints, rec

    def set_source(self, src):
        if _is_tensor(src):
            check_points_tensor(src, "set_source")
            self._src_keep = src
            self.ctx.check(capi.lib().lsa_ndt_set_source_device(self.h, src.data_ptr(), src.shape[0]))
        else:
            pts = np.ascontiguousarray(src, dtype=np.float32)
            self.ctx.check(capi.lib().lsa_ndt_set_source(self.h, pts.ctypes.data, pts.shape[0], pts.strides[0]))

    def derivatives(self, T, p, kind=0):
        """One pass at (T f32 4x4, p f64[6]) -> (score, g[6], H[6,6], pairs, npoints)."""
        out = np.zeros(32)
        Tc = capi.colmajor(T)
        p = np.ascontiguousarray(p, dtype=np.float64)
        self.ctx.check(capi.lib().lsa_ndt_derivatives(self.h, capi.fptr(Tc), capi.dptr(p), kind, capi.dptr(out)))
        return unpack_out32(out)

    def derivatives_timed(self, T, p, kind=0, repeat=20):
        out = np.zeros(32)
        ms = C.c_double(0)
        Tc = capi.colmajor(T)
        p = np.ascontiguousarray(p, dtype=np.float64)
        self.ctx.check(capi.lib().lsa_ndt_derivatives_timed(self.h, capi.fptr(Tc), capi.dptr(p), kind, repeat, capi.dptr(out), C.byref(ms)))
        return unpack_out32(out), ms.value

    def set_armed(self, on):
        """Armed passes (default on): the next pass's kernel is enqueued ahead and waits on the device for its parameters."""
        self.ctx.check(capi.lib().lsa_ndt_set_armed(self.h, 1 if on else 0))

    def set_armed_carry(self, on):
        """Opt in to leaving the last armed kernel parked for the next ScanMatch on this handle (default: cancelled at every
        ScanMatch end; lsa_pipeline_scan_match carries between the scans of its batch by itself)."""
        self.ctx.check(capi.lib().lsa_ndt_set_armed_carry(self.h, 1 if on else 0))

    def armed_counters(self):
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self.ctx.check(capi.lib().lsa_ndt_armed_counters(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(passes=a.value, cancels=b.value, fallbacks=c.value)

    def armed_stats(self, reset=False):
        """Every non-plain outcome of the armed path / the row hand-over, counted (lsa_ndt_armed_stats)."""
        st = capi.NdtArmedStats()
        self.ctx.check(capi.lib().lsa_ndt_armed_stats(self.h, C.byref(st), 1 if reset else 0))
        return st.as_dict()

    def set_armed_patience(self, patience_us):
        """diagnostics / tests: how long an armed kernel waits for its mail before it leaves by itself (default 300 us)"""
        self.ctx.check(capi.lib().lsa_ndt_set_armed_patience(self.h, float(patience_us)))

    def armed_timing(self, reset=False):
        """Per-pass means of the armed passes' timeline (lsa_ndt_armed_timing), microseconds."""
        out = (C.c_double * 6)()
        self.ctx.check(capi.lib().lsa_ndt_armed_timing(self.h, out, 1 if reset else 0))
        n = max(out[0], 1.0)
        return dict(passes=int(out[0]), post_us=out[1] / n, post_to_rows_us=out[2] / n, rows_to_post_us=out[3] / n,
                    dev_wait_us=out[4] / n, dev_body_us=out[5] / n)

    def set_source_sort(self, mode):
        """0 never, 1 when a sample says a >= 262144-point source is spatially incoherent (default), 2 always."""
        self.ctx.check(capi.lib().lsa_ndt_set_source_sort(self.h, int(mode)))

    def source_sort_count(self):
        n, last = C.c_uint64(0), C.c_int(0)
        self.ctx.check(capi.lib().lsa_ndt_source_sort_count(self.h, C.byref(n), C.byref(last)))
        return n.value, bool(last.value)

    def set_persistent(self, on):
        """ScanMatch as ONE persistent launch (opt-in: measured no faster) instead of one launch per derivative
        pass, the default (lsa_ndt_set_persistent)."""
        self.ctx.check(capi.lib().lsa_ndt_set_persistent(self.h, 1 if on else 0))

    def align_counters(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.ctx.check(capi.lib().lsa_ndt_align_counters(self.h, C.byref(a), C.byref(b)))
        return dict(launches=a.value, fallbacks=b.value)

    def set_allreduce(self, fn):
        """fn(device_ptr:int, n_doubles:int, stream:int) -> int (0 ok). Pass None to clear."""
        if fn is None:
            self._cb = capi.ALLREDUCE_FN(0)
        else:
            def _tramp(user, buf, n, stream):
                try:
                    return int(fn(buf, n, stream) or 0)
                except Exception:  # never let an exception cross the C ABI
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = capi.ALLREDUCE_FN(_tramp)
        self.ctx.check(capi.lib().lsa_ndt_set_allreduce(self.h, self._cb, None))

    def close(self):
        # a handle must be destroyed before its context; if the context is already gone, leak rather than touch it
        if self.h and self.ctx.h:
            capi.lib().lsa_ndt_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ICPRegistrationHip(RegistrationInterface):
    """ICPRegistrationManual(max_correspond_dis, max_iter) or (node) with the YAML keys of
    config/mapping/front_end.yaml:37-39 (reference: src/models/registration/icp_registration_manual.cpp:12-67)."""

    def __init__(self, ctx, max_correspond_dis=2.0, max_iter=30, node=None, mode=0, euclid_eps=0.01, trans_eps=0.01, normal_leaf=1.0):
        """mode 0 = ICPRegistrationManual; mode capi.ICP_MODE_PCL = ICPRegistration(euclid_eps, max_correspond_dis,
        trans_eps, max_iter) (icp_registration.cpp:20-33; YAML node 'ICP', config/mapping/front_end.yaml:31-35)."""
        if node is not None:
            max_correspond_dis, max_iter = float(node["max_correspond_dis"]), int(node["max_iter"])
            if "euclid_eps" in node:
                euclid_eps, trans_eps = float(node["euclid_eps"]), float(node["trans_eps"])
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(capi.lib().lsa_icp_create(C.byref(h), ctx.h, max_correspond_dis, max_iter, mode))
        self.h = h
        ctx.check(capi.lib().lsa_icp_set_criteria(h, euclid_eps, trans_eps))
        if mode == capi.ICP_MODE_PLANE:  # build extension (BASELINE.json configs[2]); not a reference plugin
            ctx.check(capi.lib().lsa_icp_set_normal_leaf(h, normal_leaf))
        self._keep = None
        self._src_keep = None

    @property
    def last_stats(self):
        st = capi.IcpStats()
        self.ctx.check(capi.lib().lsa_icp_get_stats(self.h, C.byref(st)))
        return dict(iterations=st.iterations, converged=st.converged, correspondences=st.correspondences, mse=st.mse)

    def set_armed_patience(self, microseconds):
        self.ctx.check(capi.lib().lsa_icp_set_armed_patience(self.h, float(microseconds)))

    def armed_counts(self):
        """(iterations run as armed passes, iterations that had to be run again the ordinary way) since the handle was created."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.ctx.check(capi.lib().lsa_icp_armed_counts(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def search_counts(self):
        """Last ScanMatch (ICPMANUAL / point-to-plane): (searches, queries, per-iteration list) — lsa_icp_search_counts."""
        a, b, n = C.c_uint64(0), C.c_uint64(0), C.c_size_t(0)
        per = (C.c_uint32 * 256)()
        self.ctx.check(capi.lib().lsa_icp_search_counts(self.h, C.byref(a), C.byref(b), per, 256, C.byref(n)))
        return int(a.value), int(b.value), [int(x) for x in per[: min(n.value, 256)]]

    def SetInputTarget(self, input_target):  # noqa: N802
        if _is_tensor(input_target):
            check_points_tensor(input_target, "SetInputTarget")
            self._keep = input_target
            self.ctx.check(capi.lib().lsa_icp_set_input_target_device(self.h, input_target.data_ptr(), input_target.shape[0]))
        else:
            pts = np.ascontiguousarray(input_target, dtype=np.float32)
            self.ctx.check(capi.lib().lsa_icp_set_input_target(self.h, pts.ctypes.data, pts.shape[0], pts.strides[0]))
        return True

    def ScanMatch(self, input_source, predict_pose):  # noqa: N802
        guess = capi.colmajor(predict_pose)
        out = np.zeros(16, dtype=np.float32)
        if _is_tensor(input_source):
            check_points_tensor(input_source, "ScanMatch")
            self._src_keep = input_source
            rc = capi.lib().lsa_icp_scan_match_device(self.h, input_source.data_ptr(), input_source.shape[0], capi.fptr(guess), capi.fptr(out))
        else:
            pts = np.ascontiguousarray(input_source, dtype=np.float32)
            rc = capi.lib().lsa_icp_scan_match(self.h, pts.ctypes.data, pts.shape[0], pts.strides[0], capi.fptr(guess), capi.fptr(out))
        self.ctx.check(rc)
        return True, capi.from_colmajor(out)

    def GetFitnessScore(self):  # noqa: N802
        v = C.c_float(-1)
        self.ctx.check(capi.lib().lsa_icp_get_fitness_score(self.h, C.byref(v)))
        return v.value

    def set_allreduce(self, fn):
        """Multi-GPU: fn(device_ptr:int, n_doubles:int, stream:int) -> int (0 ok) sums the iteration's 32-double block over the ranks
        in place (one all-reduce per iteration; each rank's ScanMatch gets its shard of the source).  None clears it."""
        if fn is None:
            self._cb = capi.ALLREDUCE_FN(0)
        else:
            def _tramp(user, buf, n, stream):
                try:
                    return int(fn(buf, n, stream) or 0)
                except Exception:  # never let an exception cross the C ABI
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = capi.ALLREDUCE_FN(_tramp)
        self.ctx.check(capi.lib().lsa_icp_set_allreduce(self.h, self._cb, None))

    def close(self):
        if self.h and self.ctx.h:
            capi.lib().lsa_icp_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pipeline_scan_match(frame_filter, ndt, d_scans, predict_poses, d_bufs, rank=0, world=1):
    """lsa_pipeline_scan_match: the batch form of Filter (front_end.cpp:106-107) -> ScanMatch (:225) over device-resident
    scans, the filter of scan k+1 (on the filter's own context) under the registration of scan k.  d_scans: CUDA (n, 4) f32
    tensors; predict_poses: 4x4 arrays; d_bufs: two CUDA tensors of max(n) rows.  Returns (poses, stats dicts, filtered sizes)."""
    count = len(d_scans)
    for t in list(d_scans) + list(d_bufs):
        check_points_tensor(t, "pipeline_scan_match")
    ptrs = (C.c_void_p * count)(*[t.data_ptr() for t in d_scans])
    nn = (C.c_size_t * count)(*[t.shape[0] for t in d_scans])
    pred = np.ascontiguousarray(np.stack([capi.colmajor(g) for g in predict_poses]).astype(np.float32)) if count else np.zeros((0, 16), np.float32)
    out = np.zeros((count, 16), dtype=np.float32)
    st = (capi.NdtStats * max(count, 1))()
    nf = (C.c_size_t * max(count, 1))()
    rc = capi.lib().lsa_pipeline_scan_match(frame_filter.h, ndt.h, count, ptrs, nn, capi.fptr(pred), d_bufs[0].data_ptr(), d_bufs[1].data_ptr(),
                                            rank, world, capi.fptr(out), st, nf)
    if rc:
        frame_filter.ctx.check(rc)
        ndt.ctx.check(rc)
        raise capi.LsaError(rc, "lsa_pipeline_scan_match failed")
    return [capi.from_colmajor(out[i]) for i in range(count)], [st[i].as_dict() for i in range(count)], [int(nf[i]) for i in range(count)]


def pipeline_scan_match_host(filters, ndt, clouds, predict_poses):
    """lsa_pipeline_scan_match_host: the same batch fed with HOST clouds ((n, k) float32 arrays of one row layout), ingest len(filters)
    scans deep on that many filter handles (contexts of their own).  Returns (poses, stats dicts, filtered sizes)."""
    L = capi.lib()
    L.lsa_pipeline_scan_match_host.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t,
                                               C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(capi.NdtStats), C.POINTER(C.c_size_t)]
    fh = (C.c_void_p * len(filters))(*[f.h for f in filters])
    count = len(clouds)
    arrs = [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]
    cols = arrs[0].shape[1] if count else 8
    if any(a.ndim != 2 or a.shape[1] != cols for a in arrs):
        raise ValueError("pipeline_scan_match_host: the clouds of one batch must share their row layout")
    ptrs = (C.c_void_p * max(count, 1))(*[a.ctypes.data for a in arrs])
    nn = (C.c_size_t * max(count, 1))(*[a.shape[0] for a in arrs])
    pred = np.ascontiguousarray(np.stack([capi.colmajor(g) for g in predict_poses]).astype(np.float32)) if count else np.zeros((0, 16), np.float32)
    out = np.zeros((count, 16), dtype=np.float32)
    st = (capi.NdtStats * max(count, 1))()
    nf = (C.c_size_t * max(count, 1))()
    rc = L.lsa_pipeline_scan_match_host(fh, len(filters), ndt.h, count, ptrs, nn, cols * 4, capi.w_offset(cols), capi.fptr(pred), capi.fptr(out), st, nf)
    if rc:
        for f in filters:
            f.ctx.check(rc)
        ndt.ctx.check(rc)
        raise capi.LsaError(rc, "lsa_pipeline_scan_match_host failed")
    return [capi.from_colmajor(out[i]) for i in range(count)], [st[i].as_dict() for i in range(count)], [int(nf[i]) for i in range(count)]


def unpack_out32(out):
    score = out[0]
    g = out[1:7].copy()
    H = np.zeros((6, 6))
    k = 7
    for i in range(6):
        for j in range(i, 6):
            H[i, j] = H[j, i] = out[k]
            k += 1
    return score, g, H, out[28], out[29]


def pack_out32(score, g, H, pairs, npoints):
    out = np.zeros(32)
    out[0] = score
    out[1:7] = g
    k = 7
    for i in range(6):
        for j in range(i, 6):
            out[k] = 0.5 * (H[i, j] + H[j, i])
            k += 1
    out[28] = pairs
    out[29] = npoints
    return out


def icp_gn_solve(eval_fn, max_iter, predict_pose):
    """lsa_icp_gn_solve with a Python evaluator eval_fn(T column-major 16 f32) -> out32 (the iteration's sums, already summed over the
    ranks): the host step of ICPMANUAL without a device."""
    def _tramp(user, T, out):
        try:
            o = np.asarray(eval_fn(np.ctypeslib.as_array(T, shape=(16,)).copy()), dtype=np.float64)
            for k in range(32):
                out[k] = o[k]
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1
    cb = capi.ICP_EVAL_FN(_tramp)
    guess = capi.colmajor(predict_pose)
    out = np.zeros(16, dtype=np.float32)
    rc = capi.lib().lsa_icp_gn_solve(cb, None, int(max_iter), capi.fptr(guess), capi.fptr(out))
    if rc:
        raise capi.LsaError(rc, "lsa_icp_gn_solve failed")
    return capi.from_colmajor(out)


def gn_solve(eval_fn, res, step_size, trans_eps, max_iter, outlier_ratio, predict_pose, mode=0):
    """lsa_gn_solve with a Python evaluator eval_fn(T 4x4 f32, p[6], kind) -> out32 (already summed over ranks)."""
    def _tramp(user, Tp, pp, kind, outp):
        try:
            T = capi.from_colmajor(np.ctypeslib.as_array(Tp, shape=(16,)))
            p = np.ctypeslib.as_array(pp, shape=(6,)).copy()
            out = np.asarray(eval_fn(T, p, kind), dtype=np.float64)
            np.ctypeslib.as_array(outp, shape=(32,))[:] = out
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1
    cb = capi.PASS_EVAL_FN(_tramp)
    out = np.zeros(16, dtype=np.float32)
    st = capi.NdtStats()
    guess = capi.colmajor(predict_pose)
    rc = capi.lib().lsa_gn_solve_mode(cb, None, res, step_size, trans_eps, max_iter, outlier_ratio, mode, capi.fptr(guess), capi.fptr(out), C.byref(st))
    if rc != 0:
        raise capi.LsaError(rc, "lsa_gn_solve failed")
    return capi.from_colmajor(out), st.as_dict()


def _is_tensor(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


def check_points_tensor(t, what):
    """Device clouds are packed float4 {x, y, z, intensity}: anything else would be read with the wrong stride."""
    import torch
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == 4 and t.is_contiguous()):
        raise ValueError("%s: expected a contiguous CUDA float32 tensor of shape (n, 4), got %s %s on %s (contiguous=%s)"
                         % (what, t.dtype, tuple(t.shape), t.device, t.is_contiguous()))
