"""ctypes binding of the C-ABI declared in include/lidar_slam_amd.h.

The shared library is built in-tree (lidar-slam_amd/liblidar_slam_amd.so) by
__graft_entry__.build() / `make -C lidar-slam_amd/csrc`.  There is no fallback: if the
library is missing, or no gfx950 device is usable, the calls raise LsaError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblidar_slam_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lidar_slam_amd.h")

LSA_OK = 0
LSA_ERR_INVALID, LSA_ERR_NO_DEVICE, LSA_ERR_HIP, LSA_ERR_NO_TARGET, LSA_ERR_CAPACITY, LSA_ERR_COLLECTIVE, LSA_ERR_RANGE = -1, -2, -3, -4, -5, -6, -7
POINT_STRIDE = 32
W_OFF = 16
NO_WEIGHT = C.c_size_t(-1).value  # SIZE_MAX: the cloud carries no intensity, every static weight is 1


class LsaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lidar_slam_amd error %d: %s" % (code, msg))
        self.code = code


NDT_MODE_MANUAL, NDT_MODE_PCL = 0, 1
ICP_MODE_MANUAL, ICP_MODE_PLANE, ICP_MODE_PCL = 0, 1, 2


class IcpStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("correspondences", C.c_double), ("mse", C.c_double)]


class NdtArmedStats(C.Structure):
    """lsa_ndt_armed_stats_t"""
    _fields_ = [("passes", C.c_uint64), ("cancels", C.c_uint64), ("fallbacks", C.c_uint64), ("late", C.c_uint64), ("rows_late", C.c_uint64),
                ("gave_up_seen", C.c_uint64), ("expired", C.c_uint64), ("rows_wait_max_us", C.c_double), ("parked", C.c_int32), ("pad", C.c_int32),
                ("deferred", C.c_uint64), ("host_gap_max_us", C.c_double), ("row_wait_max_us", C.c_double)]

    def as_dict(self):
        return dict(passes=self.passes, cancels=self.cancels, fallbacks=self.fallbacks, late=self.late, rows_late=self.rows_late,
                    gave_up_seen=self.gave_up_seen, expired=self.expired, rows_wait_max_us=round(self.rows_wait_max_us, 2), parked=self.parked,
                    deferred=self.deferred, host_gap_max_us=round(self.host_gap_max_us, 2), row_wait_max_us=round(self.row_wait_max_us, 2))


class NdtStats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("ls_iterations", C.c_int32),
        ("passes", C.c_int32),
        ("converged", C.c_int32),
        ("score", C.c_double),
        ("trans_probability", C.c_double),
        ("pairs_per_pass", C.c_double),
        ("final_p", C.c_double * 6),
        ("gpu_ms", C.c_double),
    ]

    def as_dict(self):
        return dict(iterations=self.iterations, ls_iterations=self.ls_iterations, passes=self.passes, converged=self.converged,
                    score=self.score, trans_probability=self.trans_probability, pairs_per_pass=self.pairs_per_pass,
                    final_p=np.array(self.final_p[:]), gpu_ms=self.gpu_ms)


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
PASS_EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double))
ICP_EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double))

_lib = None


def lib():
    """Load the product library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LsaError(LSA_ERR_NO_DEVICE, "%s not built; run __graft_entry__.build()" % LIB_PATH)
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so (SONAME libamdhip64.so.7).
    # Loading torch FIRST makes the dynamic loader satisfy this library's NEEDED libamdhip64.so.7 with that
    # already-loaded copy; the other order maps two runtimes and the second one sees no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, f, i = C.c_void_p, C.c_size_t, C.c_float, C.c_int
    PF, PD, PSZ = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_size_t)
    PVP = C.POINTER(C.c_void_p)
    L.lsa_version.restype = i
    L.lsa_context_create.argtypes = [PVP, i, vp]
    L.lsa_context_destroy.argtypes = [vp]
    L.lsa_context_destroy.restype = None
    L.lsa_last_error.argtypes = [vp]
    L.lsa_last_error.restype = C.c_char_p
    L.lsa_last_global_error.restype = C.c_char_p
    L.lsa_context_stream.argtypes = [vp]
    L.lsa_context_stream.restype = vp
    L.lsa_context_synchronize.argtypes = [vp]
    L.lsa_voxel_filter_create.argtypes = [PVP, vp, f, f, f]
    L.lsa_voxel_filter_destroy.argtypes = [vp]
    L.lsa_voxel_filter_destroy.restype = None
    L.lsa_voxel_filter_apply.argtypes = [vp, vp, sz, sz, sz, vp, sz, sz, sz, PSZ]
    L.lsa_voxel_filter_apply_device.argtypes = [vp, vp, sz, vp, PSZ]
    L.lsa_voxel_filter_apply_device_async.argtypes = [vp, vp, sz, vp]
    L.lsa_voxel_filter_wait.argtypes = [vp, PSZ]
    L.lsa_ndt_create.argtypes = [PVP, vp, f, f, f, i, f]
    L.lsa_ndt_destroy.argtypes = [vp]
    L.lsa_ndt_destroy.restype = None
    L.lsa_ndt_set_input_target.argtypes = [vp, vp, sz, sz, sz]
    L.lsa_ndt_set_input_target_device.argtypes = [vp, vp, sz]
    L.lsa_ndt_scan_match.argtypes = [vp, vp, sz, sz, PF, PF, C.POINTER(NdtStats)]
    L.lsa_ndt_scan_match_device.argtypes = [vp, vp, sz, PF, PF, C.POINTER(NdtStats)]
    L.lsa_ndt_get_fitness_score.argtypes = [vp, PF]
    L.lsa_ndt_grid_info.argtypes = [vp, C.POINTER(C.c_int64)]
    L.lsa_ndt_grid_dump.argtypes = [vp, C.POINTER(C.c_int32), PD, sz, PSZ]
    L.lsa_ndt_set_source.argtypes = [vp, vp, sz, sz]
    L.lsa_ndt_set_source_device.argtypes = [vp, vp, sz]
    L.lsa_ndt_derivatives.argtypes = [vp, PF, PD, i, PD]
    L.lsa_ndt_derivatives_timed.argtypes = [vp, PF, PD, i, i, PD, PD]
    L.lsa_ndt_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    L.lsa_icp_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    L.lsa_icp_set_allreduce_rccl.argtypes = [vp, vp]
    L.lsa_icp_gn_solve.argtypes = [ICP_EVAL_FN, vp, i, PF, PF]
    L.lsa_gn_solve.argtypes = [PASS_EVAL_FN, vp, f, f, f, i, f, PF, PF, C.POINTER(NdtStats)]
    L.lsa_pose_to_matrix.argtypes = [PD, PF]
    L.lsa_pose_to_matrix.restype = None
    L.lsa_matrix_to_pose.argtypes = [PF, PD]
    L.lsa_matrix_to_pose.restype = None
    L.lsa_eigen_f32_selftest.argtypes = [C.c_int, PF, PF]
    L.lsa_ndt_gauss_constants.argtypes = [f, f, PD, PD]
    L.lsa_ndt_gauss_constants.restype = None
    L.lsa_icp_create.argtypes = [PVP, vp, f, i, i]
    L.lsa_icp_destroy.argtypes = [vp]
    L.lsa_icp_destroy.restype = None
    L.lsa_icp_set_input_target.argtypes = [vp, vp, sz, sz]
    L.lsa_icp_set_input_target_device.argtypes = [vp, vp, sz]
    L.lsa_icp_scan_match.argtypes = [vp, vp, sz, sz, PF, PF]
    L.lsa_icp_scan_match_device.argtypes = [vp, vp, sz, PF, PF]
    L.lsa_icp_get_fitness_score.argtypes = [vp, PF]
    L.lsa_icp_set_criteria.argtypes = [vp, f, f]
    L.lsa_rccl_get_unique_id.argtypes = [vp]
    L.lsa_rccl_comm_create.argtypes = [PVP, vp, vp, i, i]
    L.lsa_rccl_comm_destroy.argtypes = [vp]
    L.lsa_rccl_comm_destroy.restype = None
    L.lsa_ndt_set_allreduce_rccl.argtypes = [vp, vp]
    L.lsa_rccl_allreduce_f64.argtypes = [vp, vp, sz]
    L.lsa_rccl_comm_count.argtypes = [vp, C.POINTER(C.c_int)]
    L.lsa_icp_set_normal_leaf.argtypes = [vp, f]
    L.lsa_yaw_init_create.argtypes = [PVP, vp, C.c_double]
    L.lsa_yaw_init_destroy.argtypes = [vp]
    L.lsa_yaw_init_destroy.restype = None
    L.lsa_yaw_init_set_local_map.argtypes = [vp, vp, sz, sz, PF]
    L.lsa_yaw_init_set_local_map_device.argtypes = [vp, vp, sz, PF]
    L.lsa_yaw_init_get_initial_yaw.argtypes = [vp, vp, sz, sz, PD, PD]
    L.lsa_yaw_init_get_initial_yaw_device.argtypes = [vp, vp, sz, PD, PD]
    L.lsa_yaw_init_grid_info.argtypes = [vp, C.POINTER(C.c_int32), PF, PF]
    L.lsa_yaw_init_cells.argtypes = [vp, PF, PF, C.POINTER(C.c_int32), sz]
    L.lsa_box_filter_create.argtypes = [PVP, vp, PF]
    L.lsa_box_filter_destroy.argtypes = [vp]
    L.lsa_box_filter_destroy.restype = None
    L.lsa_box_filter_set_size.argtypes = [vp, PF]
    L.lsa_box_filter_set_origin.argtypes = [vp, PF]
    L.lsa_box_filter_get_edge.argtypes = [vp, PF]
    L.lsa_box_filter_apply.argtypes = [vp, vp, sz, sz, sz, vp, sz, sz, sz, C.POINTER(sz)]
    L.lsa_box_filter_apply_device.argtypes = [vp, vp, sz, vp, C.POINTER(sz)]
    L.lsa_icp_get_stats.argtypes = [vp, C.POINTER(IcpStats)]
    L.lsa_icp_set_armed_patience.argtypes = [vp, C.c_double]
    L.lsa_icp_armed_counts.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.lsa_icp_search_counts.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), sz, PSZ]
    L.lsa_ndt_set_mode.argtypes = [vp, i]
    L.lsa_ndt_set_dense_limit.argtypes = [vp, C.c_longlong]
    L.lsa_gn_solve_mode.argtypes = [PASS_EVAL_FN, vp, f, f, f, i, f, i, PF, PF, C.POINTER(NdtStats)]
    L.lsa_gn_solve_trace.argtypes = [PASS_EVAL_FN, vp, f, f, f, i, f, i, PF, PF, C.POINTER(NdtStats), PD, i, C.POINTER(C.c_int)]
    L.lsa_ndt_set_trace.argtypes = [vp, PD, i, C.POINTER(C.c_int)]
    L.lsa_ndt_set_armed.argtypes = [vp, i]
    L.lsa_ndt_set_armed_carry.argtypes = [vp, i]
    L.lsa_ndt_armed_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.lsa_pipeline_scan_match.argtypes = [vp, vp, sz, PVP, PSZ, PF, vp, vp, i, i, PF, C.POINTER(NdtStats), PSZ]
    L.lsa_ndt_armed_stats.argtypes = [vp, C.POINTER(NdtArmedStats), i]
    L.lsa_ndt_set_armed_patience.argtypes = [vp, C.c_double]
    L.lsa_ndt_set_source_sort.argtypes = [vp, i]
    L.lsa_ndt_source_sort_count.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(i)]
    L.lsa_ndt_set_persistent.argtypes = [vp, i]
    L.lsa_ndt_align_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.lsa_ndt_armed_timing.argtypes = [vp, C.POINTER(C.c_double), i]
    L.lsa_ndt_align_stamps.argtypes = [vp, i, C.POINTER(C.c_uint64), sz, PSZ]
    L.lsa_gn_math_selftest.argtypes = [vp, sz, C.c_uint64, C.POINTER(C.c_uint64)]
    _lib = L
    return L


def declared_symbols():
    """Every function name declared in include/lidar_slam_amd.h (for the export test)."""
    import re
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(lsa_[a-z0-9_]+)\s*\(", txt)
    skip = {"lsa_allreduce_fn", "lsa_pass_eval_fn"}
    return sorted(set(n for n in names if n not in skip))


def w_offset(cols):
    """Byte offset of the intensity / static weight in a row of `cols` floats: (n,3) xyz rows have none (weight 1),
    (n,4) rows are packed x,y,z,w, wider rows follow pcl::PointXYZI (intensity at 16)."""
    if cols < 3:
        raise ValueError("point rows need at least 3 floats, got %d" % cols)
    return NO_WEIGHT if cols == 3 else (12 if cols == 4 else W_OFF)


def colmajor(T):
    """4x4 (row-major numpy) -> 16 floats column-major, as Eigen::Matrix4f::data()."""
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).reshape(16)


def from_colmajor(buf):
    if type(buf) is np.ndarray and buf.dtype == np.float32:
        return buf.reshape(4, 4).T.copy()
    return np.asarray(buf, dtype=np.float32).reshape(4, 4).T.copy()


_FARR = {}


def fptr(a):
    """A float32 numpy array as a ctypes argument for `float *` parameters.  (`a.ctypes.data_as(...)` costs 3-4 us per call — a tenth of
    a 40 us ScanMatch when a mirror method makes two of them; a ctypes array over the same buffer, which decays to the pointer, 0.7 us.)"""
    try:
        n = a.size
        t = _FARR.get(n)
        if t is None:
            t = _FARR[n] = C.c_float * n
        return t.from_buffer(a)
    except (TypeError, ValueError, BufferError):   # read-only or non-contiguous: the slow, general way
        return a.ctypes.data_as(C.POINTER(C.c_float))


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Context:
    def __init__(self, device=0, stream=None):
        L = lib()
        h = C.c_void_p()
        rc = L.lsa_context_create(C.byref(h), device, stream)
        if rc != LSA_OK:
            raise LsaError(rc, L.lsa_last_global_error().decode())
        self.h = h
        self.device = device

    def check(self, rc):
        if rc != LSA_OK:
            raise LsaError(rc, lib().lsa_last_error(self.h).decode())

    def synchronize(self):
        self.check(lib().lsa_context_synchronize(self.h))

    @property
    def stream(self):
        return lib().lsa_context_stream(self.h)

    def close(self):
        if self.h:
            lib().lsa_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
