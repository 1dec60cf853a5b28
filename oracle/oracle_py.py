"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (lidar-slam_amd/) never does.  See oracle/lsa_oracle.h for
the parity status ("partially pinned") and the reference citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

POINT_STRIDE = 32  # pcl::PointXYZI: x,y,z,pad | intensity,pad*3
W_OFF = 16


def build(ref: bool = True) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True)
    if ref and os.path.isdir("/root/reference/lidar_localization/third_party/eigen3"):
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


class _Stats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("ls_iterations", C.c_int),
        ("passes", C.c_int),
        ("converged", C.c_int),
        ("score", C.c_double),
        ("trans_probability", C.c_double),
        ("pairs_per_pass", C.c_double),
        ("final_p", C.c_double * 6),
    ]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def to_xyzi32(xyz, w=None):
    """(n,3)/(n,4) float array -> (n,8) float32 PointXYZI-layout array (32 B/pt)."""
    xyz = np.asarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    out = np.zeros((n, 8), dtype=np.float32)
    out[:, :3] = xyz[:, :3]
    out[:, 3] = 1.0
    if w is not None:
        out[:, 4] = np.asarray(w, dtype=np.float32)
    elif xyz.shape[1] >= 4:
        out[:, 4] = xyz[:, 3]
    else:
        out[:, 4] = 1.0
    return out


class Oracle:
    def __init__(self, fast: bool = False):
        name = "liblsa_oracle_fast.so" if fast else "liblsa_oracle.so"
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        self.L = L
        vp, sz, f, d, i = C.c_void_p, C.c_size_t, C.c_float, C.c_double, C.c_int
        PD, PF, PI = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.orc_eig3.argtypes = [PD, PD, PD]
        L.orc_inv3.argtypes = [PD, PD]
        L.orc_svd6_solve.argtypes = [PD, PD, PD]
        L.orc_svd6_solve.restype = i
        L.orc_pose_to_matrix.argtypes = [PD, PF]
        L.orc_matrix_to_pose.argtypes = [PF, PD]
        L.orc_transform_point.argtypes = [PF, PF, PF]
        L.orc_rotation_f32.argtypes = [PF, PF]
        L.orc_det6f.argtypes = [PF]
        L.orc_det6f.restype = f
        L.orc_inv6f.argtypes = [PF, PF]
        L.orc_mul66f_vec.argtypes = [PF, PF, PF]
        L.orc_mul3f.argtypes = [PF, PF, PF]
        L.orc_icp_acc.argtypes = [PF, PF, PF, PF, PF]
        L.orc_dot3.argtypes = [PD, PD]
        L.orc_dot3.restype = d
        L.orc_mv3.argtypes = [PD, PD, PD]
        L.orc_cov_expr.argtypes = [PD, PD, d, PD]
        L.orc_cov_floor_rebuild.argtypes = [PD, PD, PD]
        L.orc_grid_build.argtypes = [vp, sz, sz, sz, f]
        L.orc_grid_build.restype = vp
        L.orc_grid_free.argtypes = [vp]
        L.orc_grid_info.argtypes = [vp, PI]
        L.orc_grid_counts.argtypes = [vp, PI, PI]
        L.orc_grid_dump.argtypes = [vp, PI, PD, i]
        L.orc_grid_dump.restype = i
        L.orc_grid_set_cell.argtypes = [vp, i, i, i, PD, PD, d]
        L.orc_grid_set_cell.restype = i
        L.orc_grid_radius_search.argtypes = [vp, PF, f, PI, i]
        L.orc_grid_radius_search.restype = i
        L.orc_ndt_create.argtypes = [f, f, f, i, f]
        L.orc_ndt_create.restype = vp
        L.orc_ndt_free.argtypes = [vp]
        L.orc_ndt_set_target.argtypes = [vp, vp, sz, sz, sz]
        L.orc_ndt_set_target.restype = i
        L.orc_ndt_grid.argtypes = [vp]
        L.orc_ndt_grid.restype = vp
        L.orc_ndt_gauss.argtypes = [vp, PD, PD]
        L.orc_ndt_derivatives.argtypes = [vp, vp, sz, sz, PF, PD, i, PD, PD, C.POINTER(C.c_long)]
        L.orc_ndt_derivatives.restype = d
        L.orc_ndt_hessian.argtypes = [vp, vp, sz, sz, PF, PD, PD]
        L.orc_ndt_align.argtypes = [vp, vp, sz, sz, PF, PF, C.POINTER(_Stats)]
        L.orc_ndt_align.restype = i
        L.orc_ndt_set_trace.argtypes = [vp, PD, i, PI]
        L.orc_ndt_fitness.argtypes = [vp, vp, sz, sz, PF]
        L.orc_ndt_fitness.restype = d
        L.orc_voxel_filter.argtypes = [vp, sz, sz, sz, f, f, f, PF, sz, C.POINTER(sz)]
        L.orc_voxel_filter.restype = i
        L.orc_icp_manual.argtypes = [vp, sz, sz, vp, sz, sz, f, i, PF, PF]
        L.orc_icp_manual.restype = i
        L.orc_icp_sums.argtypes = [vp, sz, sz, vp, sz, sz, f, PF, PD]
        L.orc_icp_sums.restype = i
        L.orc_ndt_set_mode.argtypes = [vp, i]
        L.orc_icp_plane.argtypes = [vp, sz, sz, vp, sz, sz, f, i, f, PF, PF, PD]
        L.orc_icp_plane.restype = i
        L.orc_yaw_build.argtypes = [vp, sz, sz, PF, d]
        L.orc_yaw_build.restype = vp
        L.orc_yaw_free.argtypes = [vp]
        L.orc_yaw_info.argtypes = [vp, PI, PF, PF]
        L.orc_yaw_cells.argtypes = [vp, PF, PF, PI]
        L.orc_yaw_angle.argtypes = [vp, vp, sz, sz, PD]
        L.orc_yaw_angle.restype = d
        L.orc_box_filter.argtypes = [vp, sz, sz, sz, PF, PF, PF, sz, C.POINTER(sz)]
        L.orc_box_filter.restype = i
        L.orc_eig3_jacobi.argtypes = [PD, PD, PD]
        L.orc_fitness_pcl.argtypes = [vp, sz, sz, vp, sz, sz, PF, d]
        L.orc_fitness_pcl.restype = d
        L.orc_icp_pcl.argtypes = [vp, sz, sz, vp, sz, sz, f, f, f, i, PF, PF, PD]
        L.orc_icp_pcl.restype = i

    # ---- building blocks -------------------------------------------------
    def eig3(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(9)
        ev = np.zeros(3)
        ec = np.zeros(9)
        self.L.orc_eig3(_dp(a), _dp(ev), _dp(ec))
        return ev, ec.reshape(3, 3)

    def inv3(self, m):
        m = np.ascontiguousarray(m, dtype=np.float64).reshape(9)
        o = np.zeros(9)
        self.L.orc_inv3(_dp(m), _dp(o))
        return o.reshape(3, 3)

    def svd6_solve(self, H, b):
        H = np.ascontiguousarray(H, dtype=np.float64).reshape(36)
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(6)
        x = np.zeros(6)
        rank = self.L.orc_svd6_solve(_dp(H), _dp(b), _dp(x))
        return x, rank

    def pose_to_matrix(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64).reshape(6)
        T = np.zeros(16, dtype=np.float32)
        self.L.orc_pose_to_matrix(_dp(p), _fp(T))
        return T.reshape(4, 4).T.copy()  # row-major 4x4 view of the column-major buffer

    def matrix_to_pose(self, T):
        Tc = np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).reshape(16)
        p = np.zeros(6)
        self.L.orc_matrix_to_pose(_fp(Tc), _dp(p))
        return p

    # f32 Eigen blocks of the path (eigen_f32_oracle.c), row-major
    def rotation_f32(self, L):
        L = np.ascontiguousarray(L, dtype=np.float32).reshape(9)
        R = np.zeros(9, dtype=np.float32)
        self.L.orc_rotation_f32(_fp(L), _fp(R))
        return R.reshape(3, 3)

    def det6f(self, H):
        H = np.ascontiguousarray(H, dtype=np.float32).reshape(36)
        return np.float32(self.L.orc_det6f(_fp(H)))

    def inv6f(self, H, B):
        H = np.ascontiguousarray(H, dtype=np.float32).reshape(36)
        B = np.ascontiguousarray(B, dtype=np.float32).reshape(6)
        Hi = np.zeros(36, dtype=np.float32)
        d = np.zeros(6, dtype=np.float32)
        self.L.orc_inv6f(_fp(H), _fp(Hi))
        self.L.orc_mul66f_vec(_fp(Hi), _fp(B), _fp(d))
        return Hi.reshape(6, 6), d

    def mul3f(self, A, B):
        A = np.ascontiguousarray(A, dtype=np.float32).reshape(9)
        B = np.ascontiguousarray(B, dtype=np.float32).reshape(9)
        o = np.zeros(9, dtype=np.float32)
        self.L.orc_mul3f(_fp(A), _fp(B), _fp(o))
        return o.reshape(3, 3)

    def icp_acc(self, R, p, e):
        R = np.ascontiguousarray(R, dtype=np.float32).reshape(9)
        p = np.ascontiguousarray(p, dtype=np.float32).reshape(3)
        e = np.ascontiguousarray(e, dtype=np.float32).reshape(3)
        H = np.zeros(36, dtype=np.float32)
        B = np.zeros(6, dtype=np.float32)
        self.L.orc_icp_acc(_fp(R), _fp(p), _fp(e), _fp(H), _fp(B))
        return H.reshape(6, 6), B

    def dot3(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        return self.L.orc_dot3(_dp(a), _dp(b))

    def mv3(self, m, v):
        m = np.ascontiguousarray(m, dtype=np.float64).reshape(9)
        v = np.ascontiguousarray(v, dtype=np.float64)
        o = np.zeros(3)
        self.L.orc_mv3(_dp(m), _dp(v), _dp(o))
        return o

    def cov_expr(self, tmp_cov, pt_sum, n):
        t = np.ascontiguousarray(tmp_cov, dtype=np.float64).reshape(9)
        s = np.ascontiguousarray(pt_sum, dtype=np.float64)
        o = np.zeros(9)
        self.L.orc_cov_expr(_dp(t), _dp(s), float(n), _dp(o))
        return o.reshape(3, 3)

    def cov_floor_rebuild(self, evecs, evals):
        e = np.ascontiguousarray(evecs, dtype=np.float64).reshape(9)
        v = np.ascontiguousarray(evals, dtype=np.float64)
        o = np.zeros(9)
        self.L.orc_cov_floor_rebuild(_dp(e), _dp(v), _dp(o))
        return o.reshape(3, 3)

    # ---- filter ----------------------------------------------------------
    def voxel_filter(self, pts8, leaf):
        pts8 = np.ascontiguousarray(pts8, dtype=np.float32)
        n = pts8.shape[0]
        lx, ly, lz = (leaf, leaf, leaf) if np.isscalar(leaf) else leaf
        out = np.zeros((max(n, 1), 4), dtype=np.float32)
        n_out = C.c_size_t(0)
        rc = self.L.orc_voxel_filter(pts8.ctypes.data, n, POINT_STRIDE, W_OFF, lx, ly, lz, _fp(out), n, C.byref(n_out))
        return out[: n_out.value].copy(), rc


def box_filter(pts8, size, origin=(0.0, 0.0, 0.0)):
    """BoxFilter::Filter (box_filter.cpp:27-37). Returns (m, 4) x,y,z,intensity in input order."""
    o = Oracle()
    pts8 = np.ascontiguousarray(pts8, dtype=np.float32)
    n = pts8.shape[0]
    out = np.zeros((max(n, 1), 4), dtype=np.float32)
    n_out = C.c_size_t(0)
    sz_ = np.ascontiguousarray(size, dtype=np.float32)
    og = np.ascontiguousarray(origin, dtype=np.float32)
    rc = o.L.orc_box_filter(pts8.ctypes.data, n, POINT_STRIDE, W_OFF, _fp(sz_), _fp(og), _fp(out), n, C.byref(n_out))
    assert rc == 0
    return out[: n_out.value].copy()


def _colmajor(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).reshape(16)


class OracleNDT:
    """Mirror of NDTRegistrationManual (S/models/registration/ndt_registration_manual.cpp)."""

    def __init__(self, res=1.0, step_size=0.1, trans_eps=0.01, max_iter=30, outlier_ratio=0.55, fast=False, pcl_mode=False):
        self.o = Oracle(fast=fast)
        self.h = self.o.L.orc_ndt_create(res, step_size, trans_eps, max_iter, outlier_ratio)
        self.pcl_mode = bool(pcl_mode)
        if pcl_mode:  # the "NDT" plugin (pcl::NormalDistributionsTransform semantics, unpinned)
            self.o.L.orc_ndt_set_mode(self.h, 1)
        self._src = None
        self._T = np.eye(4, dtype=np.float32)
        self._keep = None

    def __del__(self):
        try:
            self.o.L.orc_ndt_free(self.h)
        except Exception:
            pass

    def set_input_target(self, pts8):
        pts8 = np.ascontiguousarray(pts8, dtype=np.float32)
        self._keep = pts8
        return self.o.L.orc_ndt_set_target(self.h, pts8.ctypes.data, pts8.shape[0], POINT_STRIDE, W_OFF)

    def gauss(self):
        d1, d2 = C.c_double(), C.c_double()
        self.o.L.orc_ndt_gauss(self.h, C.byref(d1), C.byref(d2))
        return d1.value, d2.value

    def grid_info(self):
        info = np.zeros(13, dtype=np.int32)
        self.o.L.orc_grid_info(self.o.L.orc_ndt_grid(self.h), _ip(info))
        return info

    def grid_counts(self):
        a, b = C.c_int(), C.c_int()
        self.o.L.orc_grid_counts(self.o.L.orc_ndt_grid(self.h), C.byref(a), C.byref(b))
        return a.value, b.value

    def grid_dump(self):
        occ, _ = self.grid_counts()
        ints = np.zeros((occ, 5), dtype=np.int32)
        dbls = np.zeros((occ, 13), dtype=np.float64)
        k = self.o.L.orc_grid_dump(self.o.L.orc_ndt_grid(self.h), _ip(ints), _dp(dbls), occ)
        assert k == occ
        return ints, dbls

    def override_cells(self, cell_ints, rec10):
        """Overwrite finalised cells with (centroid, symmetric icov, weight) rows, e.g. the GPU's table."""
        g = self.o.L.orc_ndt_grid(self.h)
        for (ix, iy, iz, _, st), r in zip(cell_ints, rec10):
            if st != 1:
                continue
            c = np.ascontiguousarray(r[0:3])
            ic = np.array([r[3], r[4], r[5], r[4], r[6], r[7], r[5], r[7], r[8]])
            assert self.o.L.orc_grid_set_cell(g, int(ix), int(iy), int(iz), _dp(c), _dp(ic), float(r[9])) == 0

    def derivatives(self, src8, T, p, compute_hessian=True):
        src8 = np.ascontiguousarray(src8, dtype=np.float32)
        g = np.zeros(6)
        H = np.zeros(36)
        pairs = C.c_long(0)
        Tc = _colmajor(T)
        p = np.ascontiguousarray(p, dtype=np.float64)
        s = self.o.L.orc_ndt_derivatives(self.h, src8.ctypes.data, src8.shape[0], POINT_STRIDE, _fp(Tc), _dp(p),
                                         1 if compute_hessian else 0, _dp(g), _dp(H), C.byref(pairs))
        return s, g, H.reshape(6, 6), pairs.value

    def hessian(self, src8, T, p):
        src8 = np.ascontiguousarray(src8, dtype=np.float32)
        H = np.zeros(36)
        Tc = _colmajor(T)
        p = np.ascontiguousarray(p, dtype=np.float64)
        self.o.L.orc_ndt_hessian(self.h, src8.ctypes.data, src8.shape[0], POINT_STRIDE, _fp(Tc), _dp(p), _dp(H))
        return H.reshape(6, 6)

    def scan_match(self, src8, predict_pose, trace=False):
        src8 = np.ascontiguousarray(src8, dtype=np.float32)
        self._src = src8
        out = np.zeros(16, dtype=np.float32)
        st = _Stats()
        tr = None
        ntr = C.c_int(0)
        if trace:
            tr = np.zeros((512, 9))
            self.o.L.orc_ndt_set_trace(self.h, _dp(tr), 512, C.byref(ntr))
        rc = self.o.L.orc_ndt_align(self.h, src8.ctypes.data, src8.shape[0], POINT_STRIDE, _fp(_colmajor(predict_pose)), _fp(out), C.byref(st))
        if trace:
            self.o.L.orc_ndt_set_trace(self.h, None, 0, None)
        assert rc == 0
        self._T = out.reshape(4, 4).T.copy()
        stats = dict(iterations=st.iterations, ls_iterations=st.ls_iterations, passes=st.passes, converged=st.converged,
                     score=st.score, trans_probability=st.trans_probability, pairs_per_pass=st.pairs_per_pass,
                     final_p=np.array(st.final_p[:]))
        if trace:
            stats["trace"] = tr[: ntr.value].copy()
        return self._T, stats

    def fitness(self):
        src8 = self._src
        if self.pcl_mode:  # pcl::Registration::getFitnessScore: mean squared distance to the nearest target POINT
            return fitness_pcl(self._keep, src8, self._T)
        return self.o.L.orc_ndt_fitness(self.h, src8.ctypes.data, src8.shape[0], POINT_STRIDE, _fp(_colmajor(self._T)))


def icp_manual(tgt8, src8, guess, max_correspond_dis=2.0, max_iter=30):
    o = Oracle()
    tgt8 = np.ascontiguousarray(tgt8, dtype=np.float32)
    src8 = np.ascontiguousarray(src8, dtype=np.float32)
    out = np.zeros(16, dtype=np.float32)
    o.L.orc_icp_manual(tgt8.ctypes.data, tgt8.shape[0], POINT_STRIDE, src8.ctypes.data, src8.shape[0], POINT_STRIDE,
                       max_correspond_dis, max_iter, _fp(_colmajor(guess)), _fp(out))
    return out.reshape(4, 4).T.copy()


def icp_sums(tgt8, src8, T_colmajor16, max_correspond_dis=2.0):
    """One ICPRegistrationManual iteration's 32 sums over (a shard of) the source at pose T (column-major 16 floats): the payload
    of the exchange step (orc_icp_sums)."""
    o = Oracle()
    tgt8 = np.ascontiguousarray(tgt8, dtype=np.float32)
    src8 = np.ascontiguousarray(src8, dtype=np.float32)
    T = np.ascontiguousarray(T_colmajor16, dtype=np.float32).reshape(16)
    out = np.zeros(32)
    o.L.orc_icp_sums(tgt8.ctypes.data, tgt8.shape[0], POINT_STRIDE, src8.ctypes.data, src8.shape[0], POINT_STRIDE, max_correspond_dis, _fp(T), _dp(out))
    return out


def fitness_pcl(tgt8, src8, T, max_range=float("inf")):
    """pcl::Registration::getFitnessScore (PCL 1.7 registration.hpp), consumer loop_closing.cpp:253."""
    o = Oracle()
    tgt8 = np.ascontiguousarray(tgt8, dtype=np.float32)
    src8 = np.ascontiguousarray(src8, dtype=np.float32)
    return o.L.orc_fitness_pcl(tgt8.ctypes.data, tgt8.shape[0], POINT_STRIDE, src8.ctypes.data, src8.shape[0], POINT_STRIDE,
                               _fp(_colmajor(T)), max_range)


def icp_pcl(tgt8, src8, guess, euclid_eps=0.01, max_correspond_dis=2.0, trans_eps=0.01, max_iter=30):
    """The "ICP" plugin (icp_registration.cpp:20-64 -> pcl::IterativeClosestPoint). Returns (T, stats)."""
    o = Oracle()
    tgt8 = np.ascontiguousarray(tgt8, dtype=np.float32)
    src8 = np.ascontiguousarray(src8, dtype=np.float32)
    out = np.zeros(16, dtype=np.float32)
    st = np.zeros(4)
    rc = o.L.orc_icp_pcl(tgt8.ctypes.data, tgt8.shape[0], POINT_STRIDE, src8.ctypes.data, src8.shape[0], POINT_STRIDE,
                         euclid_eps, max_correspond_dis, trans_eps, max_iter, _fp(_colmajor(guess)), _fp(out), _dp(st))
    assert rc == 0
    return out.reshape(4, 4).T.copy(), dict(iterations=int(st[0]), converged=int(st[1]), correspondences=int(st[2]), mse=st[3])


class OracleYawInit:
    """Matching::generateGauss2DMapCells + getInitialYawAngle (S/matching/matching.cpp:267-307,344-422)."""

    def __init__(self, map8, origin, grid_resolution=0.8):
        self.o = Oracle()
        map8 = np.ascontiguousarray(map8, dtype=np.float32)
        og = np.ascontiguousarray(origin, dtype=np.float32)
        self.h = self.o.L.orc_yaw_build(map8.ctypes.data, map8.shape[0], POINT_STRIDE, _fp(og), float(grid_resolution))

    def __del__(self):
        try:
            self.o.L.orc_yaw_free(self.h)
        except Exception:
            pass

    def info(self):
        wh = np.zeros(2, dtype=np.int32)
        mn, mx = np.zeros(3, dtype=np.float32), np.zeros(3, dtype=np.float32)
        self.o.L.orc_yaw_info(self.h, _ip(wh), _fp(mn), _fp(mx))
        return int(wh[0]), int(wh[1]), mn, mx

    def cells(self):
        w, h, _, _ = self.info()
        mu, sg, cnt = np.zeros(w * h, dtype=np.float32), np.zeros(w * h, dtype=np.float32), np.zeros(w * h, dtype=np.int32)
        if w * h:
            self.o.L.orc_yaw_cells(self.h, _fp(mu), _fp(sg), _ip(cnt))
        return mu.reshape(w, h), sg.reshape(w, h), cnt.reshape(w, h)

    def yaw(self, scan8):
        scan8 = np.ascontiguousarray(scan8, dtype=np.float32)
        probs = np.zeros(270)
        a = self.o.L.orc_yaw_angle(self.h, scan8.ctypes.data, scan8.shape[0], POINT_STRIDE, _dp(probs))
        return a, probs


def icp_plane(tgt8, src8, guess, max_correspond_dis=2.0, max_iter=30, normal_leaf=1.0):
    """Point-to-plane ICP (build extension, BASELINE.json configs[2]). Returns (T, #correspondences of the last iteration)."""
    o = Oracle()
    tgt8 = np.ascontiguousarray(tgt8, dtype=np.float32)
    src8 = np.ascontiguousarray(src8, dtype=np.float32)
    out = np.zeros(16, dtype=np.float32)
    st = np.zeros(2)
    rc = o.L.orc_icp_plane(tgt8.ctypes.data, tgt8.shape[0], POINT_STRIDE, src8.ctypes.data, src8.shape[0], POINT_STRIDE,
                           max_correspond_dis, max_iter, normal_leaf, _fp(_colmajor(guess)), _fp(out), _dp(st))
    assert rc == 0
    return out.reshape(4, 4).T.copy(), int(st[1])
