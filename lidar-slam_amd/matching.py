"""Host-side mirror of the part of the reference's Matching class that sits next to the registration plugin:
position-only initialisation (src/matching/matching.cpp — generateGauss2DMapCells :344-394, getInitialYawAngle
:267-307; selected by `init_type: OnlyPosition`, config/matching/matching.yaml:4)."""
import ctypes as C

import numpy as np

from . import capi


class YawInitHip:
    """grid_resolution = matching.yaml `grid_resolution` (default 0.8)."""

    def __init__(self, ctx, grid_resolution=0.8, node=None):
        if node is not None and "grid_resolution" in node:
            grid_resolution = float(node["grid_resolution"])
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(capi.lib().lsa_yaw_init_create(C.byref(h), ctx.h, float(grid_resolution)))
        self.h = h
        self._keep = None

    def SetLocalMap(self, local_map, origin):  # noqa: N802  (ResetLocalMap + SetInitPose -> generateGauss2DMapCells)
        og = np.ascontiguousarray(origin, dtype=np.float32)
        if hasattr(local_map, "data_ptr"):
            from .registration import check_points_tensor
            check_points_tensor(local_map, "SetLocalMap")
            self._keep = local_map
            self.ctx.check(capi.lib().lsa_yaw_init_set_local_map_device(self.h, local_map.data_ptr(), local_map.shape[0], capi.fptr(og)))
        else:
            pts = np.ascontiguousarray(local_map, dtype=np.float32)
            self.ctx.check(capi.lib().lsa_yaw_init_set_local_map(self.h, pts.ctypes.data, pts.shape[0], pts.strides[0], capi.fptr(og)))
        return True

    def GetInitialYawAngle(self, scan):  # noqa: N802
        """Returns (max_prob_yaw_angle, probs[270])."""
        yaw = C.c_double(0)
        probs = np.zeros(270)
        if hasattr(scan, "data_ptr"):
            from .registration import check_points_tensor
            check_points_tensor(scan, "GetInitialYawAngle")
            rc = capi.lib().lsa_yaw_init_get_initial_yaw_device(self.h, scan.data_ptr(), scan.shape[0], C.byref(yaw), capi.dptr(probs))
        else:
            pts = np.ascontiguousarray(scan, dtype=np.float32)
            rc = capi.lib().lsa_yaw_init_get_initial_yaw(self.h, pts.ctypes.data, pts.shape[0], pts.strides[0], C.byref(yaw), capi.dptr(probs))
        self.ctx.check(rc)
        return yaw.value, probs

    def grid_info(self):
        wh = np.zeros(2, dtype=np.int32)
        mn, mx = np.zeros(3, dtype=np.float32), np.zeros(3, dtype=np.float32)
        self.ctx.check(capi.lib().lsa_yaw_init_grid_info(self.h, wh.ctypes.data_as(C.POINTER(C.c_int32)), capi.fptr(mn), capi.fptr(mx)))
        return int(wh[0]), int(wh[1]), mn, mx

    def cells(self):
        w, h, _, _ = self.grid_info()
        mu, sg, cnt = np.zeros(w * h, dtype=np.float32), np.zeros(w * h, dtype=np.float32), np.zeros(w * h, dtype=np.int32)
        self.ctx.check(capi.lib().lsa_yaw_init_cells(self.h, capi.fptr(mu), capi.fptr(sg), cnt.ctypes.data_as(C.POINTER(C.c_int32)), w * h))
        return mu.reshape(w, h), sg.reshape(w, h), cnt.reshape(w, h)

    def close(self):
        if self.h and self.ctx.h:
            capi.lib().lsa_yaw_init_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
