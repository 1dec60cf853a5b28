"""Host-side mirror of the reference's cloud_filter plugin layer over the C-ABI.

Reference: lidar_localization/include/lidar_localization/models/cloud_filter/cloud_filter_interface.hpp:13-18
(CloudFilterInterface::Filter), .../voxel_filter.hpp + src/models/cloud_filter/voxel_filter.cpp:12-41 (VoxelFilter),
src/models/cloud_filter/no_filter.cpp:13-16 (NoFilter).  Same names, argument meaning and (non-)error behaviour:
Filter returns True; a cloud is an (n, 8) float32 array in pcl::PointXYZI layout (x y z 1 | intensity 0 0 0).
"""
import ctypes as C

import numpy as np

from . import capi


class CloudFilterInterface:
    def Filter(self, input_cloud, filtered_cloud=None):  # noqa: N802 (reference naming)
        raise NotImplementedError


class NoFilter(CloudFilterInterface):
    """no_filter.cpp:13-16: deep copy."""

    def Filter(self, input_cloud, filtered_cloud=None):  # noqa: N802
        return True, np.array(input_cloud, copy=True)


class VoxelFilterHip(CloudFilterInterface):
    """VoxelFilter(leaf_size_x, leaf_size_y, leaf_size_z) or VoxelFilter(node) with node['leaf_size'] = [x, y, z]."""

    def __init__(self, ctx, leaf_x=None, leaf_y=None, leaf_z=None, node=None):
        if node is not None:
            leaf_x, leaf_y, leaf_z = (float(v) for v in node["leaf_size"])
        self.ctx = ctx
        self.leaf = (float(leaf_x), float(leaf_y), float(leaf_z))
        h = C.c_void_p()
        ctx.check(capi.lib().lsa_voxel_filter_create(C.byref(h), ctx.h, *self.leaf))
        self.h = h

    def Filter(self, input_cloud, filtered_cloud=None):  # noqa: N802
        """Host path. Returns (True, filtered (m, 8) float32). `filtered_cloud` may be the input (viewer.cpp:207) or any (>= n, 8) array to write into."""
        pts = np.ascontiguousarray(input_cloud, dtype=np.float32)
        n = pts.shape[0]
        stride = pts.strides[0]
        w_off = capi.w_offset(pts.shape[1])
        # `filtered_cloud`: the caller's own (>= n, 8) float32 array, as the reference's callers hand in their cloud object
        # (front_end.cpp:106-107) — written in place, a view of its first m rows is returned.  Without one: a buffer kept across
        # calls (a fresh 4 MB array per scan costs an mmap / munmap and its page faults), of which the filtered points are copied out.
        own = (isinstance(filtered_cloud, np.ndarray) and filtered_cloud.dtype == np.float32 and filtered_cloud.ndim == 2 and filtered_cloud.shape[1] == 8
               and filtered_cloud.shape[0] >= max(n, 1) and filtered_cloud.flags.c_contiguous and filtered_cloud.flags.writeable)
        if own:
            buf = filtered_cloud
        else:
            buf = getattr(self, "_out_buf", None)
            if buf is None or buf.shape[0] < max(n, 1):
                buf = self._out_buf = np.zeros((max(n, 1) + max(n, 1) // 4, 8), dtype=np.float32)
        n_out = C.c_size_t(0)
        rc = capi.lib().lsa_voxel_filter_apply(self.h, pts.ctypes.data, n, stride, w_off, buf.ctypes.data, n, capi.POINT_STRIDE, capi.W_OFF, C.byref(n_out))
        if rc != capi.LSA_ERR_RANGE:  # (range: PCL warns and passes the input through; the plugin still returns true)
            self.ctx.check(rc)
        return True, (buf[: n_out.value] if own else buf[: n_out.value].copy())

    def counters(self):
        """lsa_voxel_filter_counters: builds, speculative misses, bitmap growths, host clouds pulled by the device."""
        out = (C.c_uint64 * 4)()
        L = capi.lib()
        L.lsa_voxel_filter_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        self.ctx.check(L.lsa_voxel_filter_counters(self.h, out))
        return dict(builds=out[0], misses=out[1], growths=out[2], pulled=out[3])

    def pull_timeouts(self):
        """lsa_voxel_filter_pull_timeouts: pulled builds whose first kernel gave up waiting for the packed rows of a host cloud."""
        out = C.c_uint64(0)
        L = capi.lib()
        L.lsa_voxel_filter_pull_timeouts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        self.ctx.check(L.lsa_voxel_filter_pull_timeouts(self.h, C.byref(out)))
        return int(out.value)

    def filter_device(self, d_in, d_out=None):
        """Device path: d_in a CUDA float32 tensor (n, 4) {x,y,z,intensity}; returns a view of d_out[:m]."""
        import torch
        assert d_in.is_cuda and d_in.dtype == torch.float32 and d_in.is_contiguous() and d_in.shape[1] == 4
        n = d_in.shape[0]
        if d_out is None:
            d_out = torch.empty((max(n, 1), 4), dtype=torch.float32, device=d_in.device)
        n_out = C.c_size_t(0)
        self.ctx.check(capi.lib().lsa_voxel_filter_apply_device(self.h, d_in.data_ptr(), n, d_out.data_ptr(), C.byref(n_out)))
        return d_out[: n_out.value]

    def close(self):
        # a handle must be destroyed before its context; if the context is already gone, leak rather than touch it
        if self.h and self.ctx.h:
            capi.lib().lsa_voxel_filter_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BoxFilterHip(CloudFilterInterface):
    """BoxFilter(node) with node['box_filter_size'] = [min_x, max_x, min_y, max_y, min_z, max_z]
    (reference: src/models/cloud_filter/box_filter.cpp:14-73 -> pcl::CropBox)."""

    def __init__(self, ctx, size=None, node=None):
        if node is not None:
            size = [float(v) for v in node["box_filter_size"]]
        self.ctx = ctx
        sz = np.ascontiguousarray(size, dtype=np.float32)
        assert sz.shape == (6,)
        h = C.c_void_p()
        ctx.check(capi.lib().lsa_box_filter_create(C.byref(h), ctx.h, capi.fptr(sz)))
        self.h = h

    def SetSize(self, size):  # noqa: N802
        sz = np.ascontiguousarray(size, dtype=np.float32)
        self.ctx.check(capi.lib().lsa_box_filter_set_size(self.h, capi.fptr(sz)))

    def SetOrigin(self, origin):  # noqa: N802
        o = np.ascontiguousarray(origin, dtype=np.float32)
        self.ctx.check(capi.lib().lsa_box_filter_set_origin(self.h, capi.fptr(o)))

    def GetEdge(self):  # noqa: N802
        e = np.zeros(6, dtype=np.float32)
        self.ctx.check(capi.lib().lsa_box_filter_get_edge(self.h, capi.fptr(e)))
        return e

    def Filter(self, input_cloud, filtered_cloud=None):  # noqa: N802
        pts = np.ascontiguousarray(input_cloud, dtype=np.float32)
        n = pts.shape[0]
        w_off = capi.w_offset(pts.shape[1])
        out = np.zeros((max(n, 1), 8), dtype=np.float32)
        n_out = C.c_size_t(0)
        self.ctx.check(capi.lib().lsa_box_filter_apply(self.h, pts.ctypes.data, n, pts.strides[0], w_off, out.ctypes.data, n, capi.POINT_STRIDE,
                                                       capi.W_OFF, C.byref(n_out)))
        return True, out[: n_out.value]

    def filter_device(self, d_in, d_out=None):
        import torch
        assert d_in.is_cuda and d_in.dtype == torch.float32 and d_in.is_contiguous() and d_in.shape[1] == 4
        n = d_in.shape[0]
        if d_out is None:
            d_out = torch.empty((max(n, 1), 4), dtype=torch.float32, device=d_in.device)
        n_out = C.c_size_t(0)
        self.ctx.check(capi.lib().lsa_box_filter_apply_device(self.h, d_in.data_ptr(), n, d_out.data_ptr(), C.byref(n_out)))
        return d_out[: n_out.value]

    def close(self):
        if self.h and self.ctx.h:
            capi.lib().lsa_box_filter_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
