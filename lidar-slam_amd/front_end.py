"""Host-side mirror of lidar_localization::FrontEnd over the C-ABI (device-resident sliding local map).

Reference: lidar_localization/src/mapping/front_end/front_end.cpp:88-341 (Update), :348-424 (UpdateWithNewFrame),
:343-346 (SetInitPose); configuration keys of config/mapping/front_end.yaml.
"""
import ctypes as C

import numpy as np

from . import capi


def _L():
    return capi.lib()


def _check_points_tensor(t, what):
    from .registration import check_points_tensor
    check_points_tensor(t, what)


class FrontEndConfig(C.Structure):
    _fields_ = [
        ("key_frame_distance", C.c_float),
        ("local_frame_num", C.c_int32),
        ("use_frame_filter", C.c_int32),
        ("use_map_filter", C.c_int32),
        ("frame_leaf", C.c_float * 3),
        ("map_leaf", C.c_float * 3),
        ("ndt_res", C.c_float),
        ("ndt_step_size", C.c_float),
        ("ndt_trans_eps", C.c_float),
        ("ndt_outlier_ratio", C.c_float),
        ("ndt_max_iter", C.c_int32),
        ("reset_intensity", C.c_int32),
    ]


class FrontEndStats(C.Structure):
    _fields_ = [
        ("key_frame", C.c_int32),
        ("local_map_frames", C.c_int32),
        ("filtered_points", C.c_int64),
        ("local_map_points", C.c_int64),
        ("target_points", C.c_int64),
        ("ndt", capi.NdtStats),
    ]


def default_config():
    """config/mapping/front_end.yaml as committed in the reference."""
    c = FrontEndConfig()
    c.key_frame_distance = 2.0
    c.local_frame_num = 20
    c.use_frame_filter = 1
    c.use_map_filter = 1
    c.frame_leaf[:] = [0.5, 0.5, 0.5]
    c.map_leaf[:] = [0.3, 0.3, 0.3]
    c.ndt_res, c.ndt_step_size, c.ndt_trans_eps, c.ndt_outlier_ratio, c.ndt_max_iter = 1.0, 0.1, 0.01, 0.55, 30
    c.reset_intensity = 1
    return c


class FrontEndHip:
    def __init__(self, ctx, config=None):
        L = capi.lib()
        L.lsa_front_end_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(FrontEndConfig)]
        L.lsa_front_end_destroy.argtypes = [C.c_void_p]
        L.lsa_front_end_destroy.restype = None
        L.lsa_front_end_set_init_pose.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.lsa_front_end_update.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_float), C.POINTER(FrontEndStats)]
        L.lsa_front_end_update_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(FrontEndStats)]
        L.lsa_front_end_update_batch_device.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_float),
                                                        C.POINTER(FrontEndStats)]
        L.lsa_front_end_registration.argtypes = [C.c_void_p]
        L.lsa_front_end_registration.restype = C.c_void_p
        L.lsa_front_end_set_incremental.argtypes = [C.c_void_p, C.c_int]
        L.lsa_front_end_map_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.lsa_front_end_map_verify.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.lsa_front_end_reserve.argtypes = [C.c_void_p, C.c_size_t]
        self.ctx = ctx
        self.cfg = config or default_config()
        h = C.c_void_p()
        ctx.check(L.lsa_front_end_create(C.byref(h), ctx.h, C.byref(self.cfg)))
        self.h = h
        self._last_st, self._last_dict = None, None
        # the pose comes back through one buffer kept for the object's life (a fresh array + its ctypes view cost ~5 us per Update)
        self._out = np.zeros(16, dtype=np.float32)
        self._out_c = capi.fptr(self._out)

    def reserve(self, max_scan_points):
        """lsa_front_end_reserve: allocate now what Update would allocate under the first scans (create already does it for 160 k points)."""
        self.ctx.check(capi.lib().lsa_front_end_reserve(self.h, int(max_scan_points)))

    def SetInitPose(self, init_pose):  # noqa: N802
        p = capi.colmajor(init_pose)
        self.ctx.check(capi.lib().lsa_front_end_set_init_pose(self.h, capi.fptr(p)))
        return True

    def Update(self, cloud):  # noqa: N802
        """cloud: (n, 8) float32 host array in PointXYZI layout, or a CUDA float32 tensor (n, 4). Returns (True, pose 4x4).

        A device cloud is the caller's again when this returns (lsa_front_end_update_device): a key-frame's raw cloud is kept in a
        copy the frame filter's first kernel makes, so the local map's update — which runs after this returns — does not read
        the tensor.  Only without a frame filter is the read stream-ordered instead: then refill the tensor by work on the
        context's stream (torch.cuda.ExternalStream(ctx.stream)) or synchronise that stream first."""
        out, out_c = self._out, self._out_c
        st = FrontEndStats()
        if hasattr(cloud, "data_ptr"):
            _check_points_tensor(cloud, "FrontEnd.Update")
            self._keep = cloud
            rc = _L().lsa_front_end_update_device(self.h, cloud.data_ptr(), cloud.shape[0], out_c, C.byref(st))
        else:
            pts = np.ascontiguousarray(cloud, dtype=np.float32)
            w_off = capi.w_offset(pts.shape[1])
            rc = _L().lsa_front_end_update(self.h, pts.ctypes.data, pts.shape[0], pts.strides[0], w_off, out_c, C.byref(st))
        if rc:
            self.ctx.check(rc)
        self._last_st, self._last_dict = st, None   # (the dict is built when somebody asks for it: a few us per scan otherwise)
        return True, out.reshape(4, 4).T.copy()

    @staticmethod
    def _stats_dict(s):
        return dict(key_frame=s.key_frame, local_map_frames=s.local_map_frames, filtered_points=s.filtered_points,
                    local_map_points=s.local_map_points, target_points=s.target_points, ndt=s.ndt.as_dict())

    @property
    def last_stats(self):
        """lsa_front_end_stats of the last scan as a dict (None before the first Update)"""
        if self._last_dict is None and self._last_st is not None:
            self._last_dict = self._stats_dict(self._last_st)
        return self._last_dict

    @property
    def last_key_frame(self):
        """1 if the last scan became a key-frame"""
        return int(self._last_st.key_frame) if self._last_st is not None else 0

    def UpdateBatch(self, clouds, stats=True):  # noqa: N802
        """A run of scans (CUDA float32 tensors (n, 4)) through Update in ONE library call with the next scan's frame filter
        overlapped (lsa_front_end_update_batch_device).  Returns (poses [count, 4, 4], per-scan stats): a list of stats dicts, or —
        stats="key_frame" — just the list of key-frame flags (a replay that only wants poses does not pay for `count` dicts)."""
        from .registration import check_points_tensor
        count = len(clouds)
        if count and not hasattr(clouds[0], "data_ptr"):
            return self._update_batch_host(clouds, stats)
        for c in clouds:
            check_points_tensor(c, "FrontEnd.UpdateBatch")
        self._keep = list(clouds)
        ptrs = (C.c_void_p * max(count, 1))(*[c.data_ptr() for c in clouds])
        ns = (C.c_size_t * max(count, 1))(*[c.shape[0] for c in clouds])
        out = np.zeros((max(count, 1), 16), dtype=np.float32)
        st = (FrontEndStats * max(count, 1))()
        self.ctx.check(capi.lib().lsa_front_end_update_batch_device(self.h, count, ptrs, ns, capi.fptr(out), st))
        if count:
            self._last_st, self._last_dict = st[count - 1], None
        poses = np.ascontiguousarray(out[:count].reshape(count, 4, 4).transpose(0, 2, 1))   # column-major 4x4s -> row-major
        if stats == "key_frame":
            return poses, [int(s.key_frame) for s in st[:count]]
        return poses, [self._stats_dict(s) for s in st[:count]]

    def set_incremental(self, on):
        """Incremental local map (default) or the full rebuild of filter output + voxel table on every key-frame."""
        self.ctx.check(capi.lib().lsa_front_end_set_incremental(self.h, 1 if on else 0))

    def map_counters(self):
        out = (C.c_uint64 * 6)()
        self.ctx.check(capi.lib().lsa_front_end_map_counters(self.h, out))
        return dict(full_builds=out[0], updates=out[1], rebuilds=out[2], refits=out[3], adopted=out[4], switched_off=out[5])

    def _update_batch_host(self, clouds, stats):
        """lsa_front_end_update_batch: the run from HOST clouds ((n, k) float32 arrays of ONE row layout: xyz, xyzw or PointXYZI)."""
        L = capi.lib()
        L.lsa_front_end_update_batch.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t,
                                                 C.POINTER(C.c_float), C.POINTER(FrontEndStats)]
        count = len(clouds)
        arrs = [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]
        cols = arrs[0].shape[1]
        if any(a.ndim != 2 or a.shape[1] != cols for a in arrs):
            raise ValueError("FrontEnd.UpdateBatch: host clouds of one run must share their row layout")
        self._keep = arrs
        ptrs = (C.c_void_p * count)(*[a.ctypes.data for a in arrs])
        ns = (C.c_size_t * count)(*[a.shape[0] for a in arrs])
        out = np.zeros((count, 16), dtype=np.float32)
        st = (FrontEndStats * count)()
        self.ctx.check(L.lsa_front_end_update_batch(self.h, count, ptrs, ns, arrs[0].strides[0] if arrs[0].shape[0] else cols * 4, capi.w_offset(cols), capi.fptr(out), st))
        self._last_st, self._last_dict = st[count - 1], None
        poses = np.ascontiguousarray(out.reshape(count, 4, 4).transpose(0, 2, 1))
        if stats == "key_frame":
            return poses, [int(st[i].key_frame) for i in range(count)]
        return poses, ([self._stats_dict(st[i]) for i in range(count)] if stats else None)

    def frame_filter_counters(self):
        """lsa_voxel_filter_counters of the front end's frame filter: builds, speculative misses, bitmap growths, host scans pulled by the device."""
        L = capi.lib()
        L.lsa_front_end_frame_filter.argtypes = [C.c_void_p]
        L.lsa_front_end_frame_filter.restype = C.c_void_p
        L.lsa_voxel_filter_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        out = (C.c_uint64 * 4)()
        self.ctx.check(L.lsa_voxel_filter_counters(L.lsa_front_end_frame_filter(self.h), out))
        return dict(builds=out[0], misses=out[1], growths=out[2], pulled=out[3])

    def armed_stats(self, reset=False):
        """lsa_ndt_armed_stats of the front end's own registration handle (every outcome of an armed pass other than the plain one)."""
        L = capi.lib()
        st = capi.NdtArmedStats()
        self.ctx.check(L.lsa_ndt_armed_stats(L.lsa_front_end_registration(self.h), C.byref(st), 1 if reset else 0))
        return st.as_dict()

    def map_verify(self):
        """Full build on scratch handles against the incremental map, bit by bit (lsa_front_end_map_verify)."""
        out = (C.c_uint64 * 6)()
        self.ctx.check(capi.lib().lsa_front_end_map_verify(self.h, out))
        return dict(cells=out[0], mismatches=out[1], live_cells=out[2], valid_voxels=out[3], full_filtered_points=out[4], live_filter_cells=out[5])

    def close(self):
        if self.h and self.ctx.h:
            capi.lib().lsa_front_end_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
