"""ctypes binding of the CPU oracle (oracle/libppp_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
PARITY UNPINNED (see oracle/ppp_oracle.cpp).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libppp_oracle.so")

PAIR_KD, PAIR_BRUTE = 0, 1
WALK_SECTPATH, WALK_CENTER_INT, WALK_SDIR_INT, WALK_V1_CONTACT, WALK_V1_SLICING = range(5)


class Params(C.Structure):
    _fields_ = [
        ("tool_radius", C.c_double),
        ("path_resolution", C.c_double),
        ("rpy_resolution", C.c_double),
        ("ee_length", C.c_float),
        ("change_range", C.c_int),
        ("pairing", C.c_int),
        ("walk", C.c_int),
        ("trim", C.c_double),
        ("drop_ends", C.c_int),
        ("smooth", C.c_int),
        ("handeye", C.c_float * 6),
        ("viewpoint", C.c_float * 3),
        ("normal_radius", C.c_float),
        ("reference_complexity", C.c_int),
        ("smooth_max_sweeps", C.c_int),
        ("dynamic_adjustment", C.c_int),
        ("depth", C.c_double),
        ("adjust_threshold", C.c_double),
        ("toolthickness", C.c_double),
        ("curvature_k", C.c_int),
        ("threads", C.c_int),
    ]


def build(force=False):
    """Compile the oracle with g++ (oracle/Makefile)."""
    src = os.path.join(_HERE, "ppp_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libppp_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def eigensolver3f(A):
    """Eigen::EigenSolver<Matrix3f> restatement: (rc, eigenvalues, eigenvectors in columns)"""
    L = lib()
    A = np.ascontiguousarray(A, np.float32).reshape(9)
    ev = np.zeros(3, np.float32); V = np.zeros(9, np.float32)
    rc = L.ppo_eigensolver3f(_f(A), _f(ev), _f(V))
    return rc, ev, V.reshape(3, 3)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp = C.c_void_p
        fp = C.POINTER(C.c_float)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.ppo_default_params.argtypes = [C.POINTER(Params)]
        L.ppo_create.restype = vp
        L.ppo_create.argtypes = [fp, C.c_size_t, C.c_size_t, C.POINTER(Params)]
        L.ppo_destroy.argtypes = [vp]
        L.ppo_num_points.restype = C.c_size_t
        L.ppo_num_points.argtypes = [vp]
        L.ppo_get_points.argtypes = [vp, fp]
        L.ppo_remove_outlier.argtypes = [vp, C.c_int, C.c_double, C.POINTER(C.c_double), fp]
        L.ppo_voxel_down.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int)]
        L.ppo_smooth_mls.argtypes = [vp, C.c_double, C.c_int]
        L.ppo_trans2center.argtypes = [vp, fp, fp, fp]
        L.ppo_eigensolver3f.argtypes = [fp, fp, fp]
        L.ppo_minmax.argtypes = [vp, fp, fp]
        L.ppo_slice_positions.argtypes = [vp, fp, C.c_int]
        L.ppo_ranged_x_index.argtypes = [vp, C.c_int, ip, C.c_int]
        L.ppo_insert_point.argtypes = [vp, ip, C.c_int, C.c_float, dp, dp, dp, C.c_int]
        L.ppo_gen_path.argtypes = [vp]
        L.ppo_num_slices.argtypes = [vp]
        L.ppo_get_nodes.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int]
        L.ppo_get_boundary.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int]
        L.ppo_get_slice_indices.argtypes = [vp, C.c_int, ip, C.c_int]
        L.ppo_eval_spline.argtypes = [vp, C.c_int, dp, C.c_int, dp]
        L.ppo_get_path.argtypes = [vp]
        L.ppo_num_waypoints.argtypes = [vp]
        for name in ("ppo_get_waypoints", "ppo_get_waypoints_xyz", "ppo_get_waypoint_normals",
                     "ppo_get_waypoints_presmooth", "ppo_get_waypoints_smoothed"):
            getattr(L, name).argtypes = [vp, fp]
        L.ppo_get_waypoint_nn.argtypes = [vp, ip]
        L.ppo_get_tail_index.argtypes = [vp, ip, C.c_int]
        L.ppo_smooth_sweeps.argtypes = [vp]
        L.ppo_rpy_oob.argtypes = [vp]
        L.ppo_estimate_normals.argtypes = [vp, fp]
        L.ppo_normal_at.argtypes = [vp, C.c_int, fp]
        L.ppo_knn.argtypes = [vp, fp, C.c_int, ip]
        L.ppo_principal_curvature.argtypes = [vp, fp, fp]
        L.ppo_area2cloud.argtypes = [vp, dp, C.c_int, fp]
        L.ppo_nearest.argtypes = [vp, fp, fp]
        L.ppo_radius_search.argtypes = [vp, fp, C.c_float, ip, C.c_int]
        L.ppo_steffen.argtypes = [C.c_int, dp, dp, dp, C.c_int, dp]
        L.ppo_eigen33.argtypes = [fp, fp, fp]
        L.ppo_euler_zyx.argtypes = [fp, fp]
        L.ppo_handeye.argtypes = [fp, fp]
        L.ppo_pose_from_normal.argtypes = [fp, fp]
        L.ppo_position_smooth.argtypes = [fp, C.c_int, C.c_int]
        L.ppo_reduce_rpy.argtypes = [fp, C.c_int, ip, C.c_int, C.c_double]
        L.ppo_trans_flange.argtypes = [fp, C.c_int, C.c_float]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def default_params(**kw):
    p = Params()
    lib().ppo_default_params(C.byref(p))
    for k, v in kw.items():
        if k in ("handeye", "viewpoint"):
            arr = getattr(p, k)
            for j, x in enumerate(v):
                arr[j] = x
        else:
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
    return p


class Oracle:
    """One planner object of the reference (SectPath / path_generater), on the CPU."""

    def __init__(self, xyz, params=None, **kw):
        self.L = lib()
        self.params = params if params is not None else default_params(**kw)
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        assert xyz.ndim == 2 and xyz.shape[1] >= 3
        self.n = xyz.shape[0]
        self.h = self.L.ppo_create(_f(xyz), self.n, xyz.shape[1], C.byref(self.params))

    def close(self):
        if self.h:
            self.L.ppo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- cloud --
    def points(self):
        out = np.empty((self.n, 3), np.float32)
        self.L.ppo_get_points(self.h, _f(out))
        return out

    def minmax(self):
        mn = np.empty(3, np.float32)
        mx = np.empty(3, np.float32)
        self.L.ppo_minmax(self.h, _f(mn), _f(mx))
        return mn, mx

    # -- slicing --
    def slice_positions(self):
        S = self.L.ppo_slice_positions(self.h, None, 0)
        px = np.empty(max(S, 1), np.float32)
        self.L.ppo_slice_positions(self.h, _f(px), S)
        return px[:S]

    def ranged_x_index(self, position):
        n = self.L.ppo_ranged_x_index(self.h, int(position), None, 0)
        out = np.empty(max(n, 1), np.int32)
        self.L.ppo_ranged_x_index(self.h, int(position), _i(out), n)
        return out[:n]

    def insert_point(self, indices, plane_x):
        indices = np.ascontiguousarray(indices, np.int32)
        cap = max(len(indices), 1)
        y = np.empty(cap); x = np.empty(cap); z = np.empty(cap)
        m = self.L.ppo_insert_point(self.h, _i(indices), len(indices), float(plane_x), _d(y), _d(x), _d(z), cap)
        if m < 0:
            return m, None, None, None
        return m, y[:m], x[:m], z[:m]

    def gen_path(self):
        return self.L.ppo_gen_path(self.h)

    def num_slices(self):
        return self.L.ppo_num_slices(self.h)

    def nodes(self, s):
        m = self.L.ppo_get_nodes(self.h, s, None, None, None, 0)
        y = np.empty(m); x = np.empty(m); z = np.empty(m)
        self.L.ppo_get_nodes(self.h, s, _d(y), _d(x), _d(z), m)
        return y, x, z

    def boundary(self, s):
        """knots (y, x, z) of the boundary spline slice s was adjusted against; empty arrays when there is none"""
        m = self.L.ppo_get_boundary(self.h, s, None, None, None, 0)
        y = np.empty(m); x = np.empty(m); z = np.empty(m)
        if m:
            self.L.ppo_get_boundary(self.h, s, _d(y), _d(x), _d(z), m)
        return y, x, z

    def slice_indices(self, s):
        n = self.L.ppo_get_slice_indices(self.h, s, None, 0)
        out = np.empty(max(n, 1), np.int32)
        self.L.ppo_get_slice_indices(self.h, s, _i(out), n)
        return out[:n]

    def eval_spline(self, s, y):
        y = np.ascontiguousarray(y, np.float64)
        out = np.empty((len(y), 3))
        rc = self.L.ppo_eval_spline(self.h, s, _d(y), len(y), _d(out))
        return rc, out

    # -- getPath --
    def get_path(self):
        return self.L.ppo_get_path(self.h)

    def _wp(self, fn, width, dtype=np.float32):
        W = self.L.ppo_num_waypoints(self.h)
        out = np.empty((W, width), dtype)
        if W:
            fn(self.h, out.ctypes.data_as(C.POINTER(C.c_float if dtype == np.float32 else C.c_int)))
        return out

    def waypoints(self):
        return self._wp(self.L.ppo_get_waypoints, 6)

    def waypoints_xyz(self):
        return self._wp(self.L.ppo_get_waypoints_xyz, 3)

    def waypoint_normals(self):
        return self._wp(self.L.ppo_get_waypoint_normals, 4)

    def waypoints_presmooth(self):
        return self._wp(self.L.ppo_get_waypoints_presmooth, 6)

    def waypoints_smoothed(self):
        return self._wp(self.L.ppo_get_waypoints_smoothed, 6)

    def waypoint_nn(self):
        return self._wp(self.L.ppo_get_waypoint_nn, 1, np.int32)[:, 0]

    def tail_index(self):
        n = self.L.ppo_get_tail_index(self.h, None, 0)
        out = np.empty(max(n, 1), np.int32)
        self.L.ppo_get_tail_index(self.h, _i(out), n)
        return out[:n]

    def smooth_sweeps(self):
        return self.L.ppo_smooth_sweeps(self.h)

    def rpy_oob(self):
        return self.L.ppo_rpy_oob(self.h)

    # -- normals / kd-tree --
    def estimate_normals(self):
        out = np.empty((self.n, 4), np.float32)
        self.L.ppo_estimate_normals(self.h, _f(out))
        return out

    def normal_at(self, idx):
        out = np.empty(4, np.float32)
        self.L.ppo_normal_at(self.h, int(idx), _f(out))
        return out

    def num_points(self):
        return int(self.L.ppo_num_points(self.h))

    def remove_outlier(self, mean_k=50, std_mul=1.0):
        """SectPath::remove_outlier; returns (new size, threshold, mean distances of the original points)"""
        n0 = self.num_points()
        thr = C.c_double()
        dist = np.zeros(max(n0, 1), np.float32)
        rc = self.L.ppo_remove_outlier(self.h, int(mean_k), float(std_mul), C.byref(thr), _f(dist))
        if rc >= 0:
            self.n = rc
        return rc, thr.value, dist[:n0]

    def voxel_down(self, lx, ly, lz):
        """path_generater::voxel_down; returns (new size, overflow flag)"""
        ov = C.c_int()
        rc = self.L.ppo_voxel_down(self.h, float(lx), float(ly), float(lz), C.byref(ov))
        if rc >= 0:
            self.n = rc
        return rc, bool(ov.value)

    def smooth_mls(self, radius=15.0, order=3):
        """SectPath::smooth (pcl::MovingLeastSquares); returns the new size"""
        rc = self.L.ppo_smooth_mls(self.h, float(radius), int(order))
        if rc >= 0:
            self.n = rc
        return rc

    def trans2center(self):
        """SectPath::trans2center; returns (rc, TransAlign 4x4, centroid, covariance 3x3)"""
        T = np.zeros(16, np.float32); c = np.zeros(3, np.float32); cov = np.zeros(9, np.float32)
        rc = self.L.ppo_trans2center(self.h, _f(T), _f(c), _f(cov))
        return rc, T.reshape(4, 4), c, cov.reshape(3, 3)

    def knn(self, q, k):
        q = np.ascontiguousarray(q, np.float32)
        out = np.empty(k, np.int32)
        n = self.L.ppo_knn(self.h, _f(q), k, _i(out))
        return out[:n]

    def principal_curvature(self, q):
        q = np.ascontiguousarray(q, np.float32)
        out = np.empty(5, np.float32)
        self.L.ppo_principal_curvature(self.h, _f(q), _f(out))
        return out

    def area2cloud(self, p, key):
        p = np.ascontiguousarray(p, np.float64)
        out = np.empty(3, np.float32)
        self.L.ppo_area2cloud(self.h, _d(p), int(key), _f(out))
        return out

    def nearest(self, q):
        q = np.ascontiguousarray(q, np.float32)
        d2 = C.c_float()
        i = self.L.ppo_nearest(self.h, _f(q), C.byref(d2))
        return i, d2.value

    def radius_search(self, q, r):
        q = np.ascontiguousarray(q, np.float32)
        n = self.L.ppo_radius_search(self.h, _f(q), r, None, 0)
        out = np.empty(max(n, 1), np.int32)
        self.L.ppo_radius_search(self.h, _f(q), r, _i(out), n)
        return out[:n]


# ---- stateless helpers ----
def steffen(xs, ys, xq):
    xs = np.ascontiguousarray(xs, np.float64); ys = np.ascontiguousarray(ys, np.float64)
    xq = np.ascontiguousarray(xq, np.float64)
    out = np.empty(len(xq))
    rc = lib().ppo_steffen(len(xs), _d(xs), _d(ys), _d(xq), len(xq), _d(out))
    return rc, out


def eigen33(cov):
    cov = np.ascontiguousarray(cov, np.float32).reshape(9)
    ev = C.c_float()
    vec = np.empty(3, np.float32)
    lib().ppo_eigen33(_f(cov), C.byref(ev), _f(vec))
    return ev.value, vec


def euler_zyx(m):
    m = np.ascontiguousarray(m, np.float32).reshape(9)
    e = np.empty(3, np.float32)
    lib().ppo_euler_zyx(_f(m), _f(e))
    return e


def handeye(he, wp):
    he = np.ascontiguousarray(he, np.float32)
    wp = np.array(wp, np.float32)
    lib().ppo_handeye(_f(he), _f(wp))
    return wp


def pose_from_normal(n):
    n = np.ascontiguousarray(n, np.float32)
    out = np.empty(3, np.float32)
    lib().ppo_pose_from_normal(_f(n), _f(out))
    return out


def position_smooth(wp6, max_sweeps=200):
    wp6 = np.array(wp6, np.float32)
    s = lib().ppo_position_smooth(_f(wp6), len(wp6), max_sweeps)
    return s, wp6


def reduce_rpy(wp6, tail, rpy_res):
    wp6 = np.array(wp6, np.float32)
    tail = np.ascontiguousarray(tail, np.int32)
    oob = lib().ppo_reduce_rpy(_f(wp6), len(wp6), _i(tail), len(tail), float(rpy_res))
    return oob, wp6


def trans_flange(wp6, ee_len):
    wp6 = np.array(wp6, np.float32)
    lib().ppo_trans_flange(_f(wp6), len(wp6), float(ee_len))
    return wp6
