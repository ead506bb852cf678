"""ctypes binding of the gfx950 engine (libppp_hip.so, C ABI in include/ppp_hip.h).

This module is plumbing only: every number comes from the HIP kernels.  There is no
CPU fallback -- if the shared library is missing or no MI355X is visible, construction
raises.  The oracle under oracle/ is never imported from here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libppp_hip.so")

PAIR_KD, PAIR_BRUTE = 0, 1
WALK_SECTPATH, WALK_CENTER_INT, WALK_SDIR_INT, WALK_V1_CONTACT, WALK_V1_SLICING = range(5)
STAGE_WP_XYZ, STAGE_WP_NN, STAGE_WP_NORMAL, STAGE_WP_PRESMOOTH, STAGE_WP_SMOOTHED = range(5)

OK, ERR_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_SLICE, ERR_CAPACITY, ERR_DOMAIN, ERR_UNSUPPORTED, ERR_IO = 0, -1, -2, -3, -4, -5, -6, -7, -8


class PPPError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ppp error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    pass  # fields assigned after Params


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
        ("normal_radius", C.c_float),
        ("smooth_max_sweeps", C.c_int),
        ("alignment", C.c_int),
        ("dynamic_adjustment", C.c_int),
        ("depth", C.c_double),
        ("adjust_threshold", C.c_double),
        ("toolthickness", C.c_double),
        ("curvature_k", C.c_int),
        ("slice_begin", C.c_int),
        ("slice_end", C.c_int),
        ("range_margin", C.c_float),
    ]


Config._fields_ = [
    ("params", Params),
    ("path_file", C.c_char * 512),
    ("depth", C.c_double),
    ("adjust_threshold", C.c_double),
    ("toolthickness", C.c_double),
    ("smooth_cloud", C.c_int),
    ("remove_outlier", C.c_int),
    ("alignment", C.c_int),
    ("dynamic_adjustment", C.c_int),
]

EXPORTS = [
    "ppp_default_params", "ppp_create", "ppp_destroy", "ppp_last_error", "ppp_version", "ppp_set_params",
    "ppp_set_cloud", "ppp_set_cloud_device", "ppp_num_points", "ppp_gen_path_async", "ppp_get_path_async", "ppp_run_async",
    "ppp_sync", "ppp_failed_slice", "ppp_num_slices", "ppp_num_waypoints", "ppp_get_waypoints",
    "ppp_get_waypoints_device", "ppp_copy_waypoints_to_device", "ppp_get_tail_index", "ppp_minmax", "ppp_get_slice_positions",
    "ppp_get_slice_indices", "ppp_get_nodes", "ppp_get_boundary", "ppp_eval_spline", "ppp_ranged_x_index", "ppp_insert_point",
    "ppp_normals_at", "ppp_estimate_normals", "ppp_area2cloud", "ppp_nearest", "ppp_get_stage", "ppp_smooth_sweeps", "ppp_enable_timing",
    "ppp_get_kernel_times", "ppp_load_pcd", "ppp_save_pcd", "ppp_free", "ppp_default_config", "ppp_read_config",
    "ppp_write_path_file", "ppp_run_batch_async", "ppp_sync_batch", "ppp_get_stream", "ppp_gather_waypoints", "ppp_get_cloud", "ppp_remove_outlier", "ppp_voxel_down", "ppp_smooth_mls", "ppp_trans2center", "ppp_get_waypoint_counts", "ppp_copy_stage_to_device", "ppp_finish_path_async",
    "ppp_save_pcd_rgb", "ppp_range_interval", "ppp_set_cloud_part", "ppp_spline_create", "ppp_spline_restart", "ppp_spline_eval", "ppp_spline_range", "ppp_spline_destroy",
    "ppp_set_fast_path", "ppp_get_fast_path", "ppp_set_plan_reuse", "ppp_set_cloud_pcd", "ppp_pcd_probe", "ppp_set_cloud_device_async",
    "ppp_set_side_by_side", "ppp_queue_create", "ppp_queue_destroy", "ppp_queue_submit", "ppp_queue_wait", "ppp_queue_lanes", "ppp_queue_lane", "ppp_queue_last_error",
]


def build(force=False):
    """Compile the HIP engine for gfx950 (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    newest = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir))
    newest = max(newest, os.path.getmtime(os.path.join(_HERE, "..", "include", "ppp_hip.h")))
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        subprocess.check_call(["make", "-j4", "-C", src_dir], stdout=subprocess.DEVNULL)  # the engine and the window kernels build side by side
    return LIB_PATH


_lib = None


def lib():
    """Loads libppp_hip.so (never builds implicitly: a missing library is an error)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PPPError(ERR_NO_DEVICE, "libppp_hip.so is missing: run __graft_entry__.build() (no CPU fallback exists)")
        L = C.CDLL(LIB_PATH)
        vp, sz = C.c_void_p, C.c_size_t
        fp, dp, ip = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
        szp = C.POINTER(C.c_size_t)
        L.ppp_default_params.argtypes = [C.POINTER(Params)]
        L.ppp_default_params.restype = None
        L.ppp_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.ppp_destroy.argtypes = [vp]
        L.ppp_last_error.argtypes = [vp]
        L.ppp_last_error.restype = C.c_char_p
        L.ppp_version.restype = C.c_char_p
        L.ppp_set_params.argtypes = [vp, C.POINTER(Params)]
        L.ppp_set_cloud.argtypes = [vp, vp, sz, sz, fp]
        L.ppp_set_cloud_device.argtypes = [vp, vp, sz, sz, fp]
        L.ppp_set_cloud_device_async.argtypes = [vp, vp, sz, sz, fp]
        L.ppp_set_side_by_side.argtypes = [vp, C.c_int]
        L.ppp_queue_create.argtypes = [C.c_int, C.c_int, C.POINTER(Params), C.POINTER(vp)]
        L.ppp_queue_destroy.argtypes = [vp]; L.ppp_queue_destroy.restype = None
        L.ppp_queue_submit.argtypes = [vp, vp, sz, sz, fp, C.POINTER(C.c_longlong)]
        L.ppp_queue_wait.argtypes = [vp, C.c_longlong, szp, C.POINTER(vp)]
        L.ppp_queue_lanes.argtypes = [vp]
        L.ppp_queue_lane.argtypes = [vp, C.c_int]; L.ppp_queue_lane.restype = vp
        L.ppp_queue_last_error.argtypes = [vp]; L.ppp_queue_last_error.restype = C.c_char_p
        L.ppp_num_points.argtypes = [vp, szp]
        L.ppp_range_interval.argtypes = [C.POINTER(Params), C.c_float, C.c_float, fp, fp, ip]
        L.ppp_set_cloud_part.argtypes = [vp, vp, sz, sz, fp, ip, fp, fp, sz, C.c_float, C.c_float]
        L.ppp_gen_path_async.argtypes = [vp]
        L.ppp_get_path_async.argtypes = [vp]
        L.ppp_run_async.argtypes = [vp]
        L.ppp_sync.argtypes = [vp]
        L.ppp_failed_slice.argtypes = [vp]
        L.ppp_num_slices.argtypes = [vp, ip]
        L.ppp_num_waypoints.argtypes = [vp, szp]
        L.ppp_get_waypoints.argtypes = [vp, fp, sz, szp]
        L.ppp_get_waypoints_device.argtypes = [vp, C.POINTER(vp), szp]
        L.ppp_copy_waypoints_to_device.argtypes = [vp, vp, sz, szp]
        L.ppp_get_tail_index.argtypes = [vp, ip, sz, szp]
        L.ppp_get_waypoint_counts.argtypes = [vp, ip, sz, szp]
        L.ppp_run_batch_async.argtypes = [C.POINTER(vp), sz, vp, szp, szp]
        L.ppp_sync_batch.argtypes = [C.POINTER(vp), sz, szp]
        L.ppp_get_stream.argtypes = [vp, C.POINTER(vp)]
        L.ppp_gather_waypoints.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, szp, vp]
        L.ppp_get_cloud.argtypes = [vp, fp, sz, szp]
        L.ppp_remove_outlier.argtypes = [vp, C.c_int, C.c_double, szp, C.POINTER(C.c_double)]
        L.ppp_voxel_down.argtypes = [vp, C.c_float, C.c_float, C.c_float, szp, C.POINTER(C.c_int)]
        L.ppp_smooth_mls.argtypes = [vp, C.c_double, C.c_int, szp]
        L.ppp_trans2center.argtypes = [vp, fp, fp, fp]
        L.ppp_copy_stage_to_device.argtypes = [vp, C.c_int, vp, sz, szp]
        L.ppp_finish_path_async.argtypes = [vp, vp, sz, ip, sz]
        L.ppp_minmax.argtypes = [vp, fp, fp]
        L.ppp_get_slice_positions.argtypes = [vp, fp, sz, szp]
        L.ppp_get_slice_indices.argtypes = [vp, C.c_int, ip, sz, szp]
        L.ppp_get_nodes.argtypes = [vp, C.c_int, dp, dp, dp, sz, szp]
        L.ppp_get_boundary.argtypes = [vp, C.c_int, dp, dp, dp, sz, szp, C.POINTER(C.c_int)]
        L.ppp_eval_spline.argtypes = [vp, C.c_int, dp, sz, dp]
        L.ppp_ranged_x_index.argtypes = [vp, C.c_int, ip, sz, szp]
        L.ppp_insert_point.argtypes = [vp, ip, sz, C.c_float, dp, dp, dp, sz, szp]
        L.ppp_normals_at.argtypes = [vp, ip, sz, fp]
        L.ppp_estimate_normals.argtypes = [vp, fp]
        L.ppp_area2cloud.argtypes = [vp, dp, sz, C.c_int, fp]
        L.ppp_nearest.argtypes = [vp, fp, sz, ip]
        L.ppp_get_stage.argtypes = [vp, C.c_int, vp, sz, szp]
        L.ppp_smooth_sweeps.argtypes = [vp, ip]
        L.ppp_enable_timing.argtypes = [vp, C.c_int]
        L.ppp_get_kernel_times.argtypes = [vp, C.c_char_p, fp, ip, sz, szp]
        L.ppp_load_pcd.argtypes = [C.c_char_p, C.POINTER(fp), szp, fp]
        L.ppp_save_pcd.argtypes = [C.c_char_p, fp, sz, sz, fp, C.c_int]
        L.ppp_free.argtypes = [vp]
        L.ppp_free.restype = None
        L.ppp_default_config.argtypes = [C.POINTER(Config)]
        L.ppp_default_config.restype = None
        L.ppp_read_config.argtypes = [C.c_char_p, C.POINTER(Config)]
        L.ppp_write_path_file.argtypes = [C.c_char_p, fp, sz]
        L.ppp_spline_create.argtypes = [C.c_int, sz, dp, dp, dp, C.POINTER(vp)]
        L.ppp_spline_restart.argtypes = [vp, sz, dp, dp, dp]
        L.ppp_spline_eval.argtypes = [vp, dp, sz, dp]
        L.ppp_spline_range.argtypes = [vp, dp, dp, szp]
        L.ppp_spline_destroy.argtypes = [vp]
        L.ppp_set_fast_path.argtypes = [vp, C.c_int]
        L.ppp_set_plan_reuse.argtypes = [vp, C.c_int]
        L.ppp_set_cloud_pcd.argtypes = [vp, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_float)]
        L.ppp_pcd_probe.argtypes = [C.c_char_p, C.c_void_p]
        L.ppp_get_fast_path.argtypes = [vp, ip]
        _lib = L
    return _lib


def default_params(**kw):
    p = Params()
    lib().ppp_default_params(C.byref(p))
    for k, v in kw.items():
        if k == "handeye":
            for j, x in enumerate(v):
                p.handeye[j] = x
        else:
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
    return p


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ---- host-side file formats (no GPU needed) ----
def load_pcd(path):
    """pcl::io::loadPCDFile: returns (xyz float32 [n,3], viewpoint float32 [7])."""
    L = lib()
    p = C.POINTER(C.c_float)()
    n = C.c_size_t()
    vp = np.zeros(7, np.float32)
    rc = L.ppp_load_pcd(path.encode(), C.byref(p), C.byref(n), _f(vp))
    if rc:
        raise PPPError(rc, "cannot read PCD %s" % path)
    xyz = np.ctypeslib.as_array(p, shape=(max(n.value, 1) * 3,))[: n.value * 3].reshape(-1, 3).copy()
    L.ppp_free(p)
    return xyz, vp


class PlannerQueue:
    """ppp_queue_*: workpieces in, lists out, `lanes` engine handles (default 2) behind it taking turns -- the passes of neighbouring
    workpieces overlap on the device.  submit(dptr, n) takes a cloud that is already in device memory (untouched until wait() of its
    ticket has returned) and returns a ticket; wait(ticket) returns (W, device pointer of the W x 6 list) -- valid until `lanes` more
    workpieces have been submitted -- or raises what that workpiece ended with."""

    def __init__(self, device=0, lanes=0, **params):
        self.L = lib()
        p = Params()
        self.L.ppp_default_params(C.byref(p))
        for k, v in params.items():
            if k == "handeye":
                for j, x in enumerate(v):
                    p.handeye[j] = x
            else:
                setattr(p, k, v)
        q = C.c_void_p()
        rc = self.L.ppp_queue_create(device, lanes, C.byref(p), C.byref(q))
        if rc:
            raise PPPError(rc, "ppp_queue_create failed")
        self.q = q
        self.lanes = self.L.ppp_queue_lanes(q)

    def _chk(self, rc):
        if rc:
            raise PPPError(rc, self.L.ppp_queue_last_error(self.q).decode())

    def submit(self, dptr, n, stride_bytes=12, viewpoint=None):
        t = C.c_longlong()
        vp = None if viewpoint is None else _f(np.ascontiguousarray(viewpoint, np.float32))
        self._chk(self.L.ppp_queue_submit(self.q, C.c_void_p(dptr), n, stride_bytes, vp, C.byref(t)))
        return t.value

    def wait(self, ticket):
        W = C.c_size_t()
        d = C.c_void_p()
        self._chk(self.L.ppp_queue_wait(self.q, ticket, C.byref(W), C.byref(d)))
        return W.value, d.value

    def close(self):
        if self.q:
            self.L.ppp_queue_destroy(self.q)
            self.q = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PcdLayout(C.Structure):
    _fields_ = [("data_kind", C.c_int), ("points", C.c_size_t), ("record_bytes", C.c_size_t), ("x_offset", C.c_int), ("y_offset", C.c_int),
                ("z_offset", C.c_int), ("xyz_float32", C.c_int), ("data_offset", C.c_longlong), ("viewpoint", C.c_float * 7)]


def pcd_probe(path):
    """The header of a PCD file: ppp_pcd_layout (data_kind 0 ascii / 1 binary / 2 binary_compressed, points, record layout)."""
    lay = PcdLayout()
    rc = lib().ppp_pcd_probe(path.encode(), C.byref(lay))
    if rc:
        raise PPPError(rc, "cannot read PCD %s" % path)
    return lay


def save_pcd(path, xyz, viewpoint=None, binary=True):
    xyz = np.ascontiguousarray(xyz, np.float32)
    vp = None if viewpoint is None else _f(np.ascontiguousarray(viewpoint, np.float32))
    mode = 2 if binary == "compressed" else (1 if binary else 0)
    rc = lib().ppp_save_pcd(path.encode(), _f(xyz), xyz.shape[0], xyz.shape[1], vp, mode)
    if rc:
        raise PPPError(rc, "cannot write PCD %s" % path)


def read_config(path):
    c = Config()
    lib().ppp_default_config(C.byref(c))
    rc = lib().ppp_read_config(path.encode(), C.byref(c))
    return rc, c


def write_path_file(path, wp6):
    wp6 = np.ascontiguousarray(wp6, np.float32)
    rc = lib().ppp_write_path_file(path.encode(), _f(wp6), wp6.shape[0])
    if rc:
        raise PPPError(rc, "cannot write %s" % path)


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def run_batch_async(engines, dst_ptr=None, offsets=None, caps=None):
    """GenPath + getPath of several handles of one GPU as ONE hipGraph with a branch per handle (BASELINE config 3).
    With dst_ptr every list is also copied to dst_ptr + 24 * offsets[i] bytes (at most caps[i] rows)."""
    L = lib()
    n = len(engines)
    hs = (C.c_void_p * n)(*[e.h for e in engines])
    if dst_ptr is None:
        rc = L.ppp_run_batch_async(hs, n, None, None, None)
    else:
        off = (C.c_size_t * n)(*[int(x) for x in offsets])
        cap = (C.c_size_t * n)(*[int(x) for x in caps])
        rc = L.ppp_run_batch_async(hs, n, C.c_void_p(dst_ptr), off, cap)
    if rc:
        raise PPPError(rc, L.ppp_last_error(engines[0].h).decode())


def sync_batch(engines):
    L = lib()
    n = len(engines)
    hs = (C.c_void_p * n)(*[e.h for e in engines])
    bad = C.c_size_t(0)
    rc = L.ppp_sync_batch(hs, n, C.byref(bad))
    if rc:
        raise PPPError(rc, "handle %d of the batch: %s" % (bad.value, L.ppp_last_error(engines[bad.value].h).decode()))


class Spline:
    """class Spline of the reference (include/Spline.h:7-51) on caller-supplied knots: two Steffen interpolants y -> x, y -> z
    evaluated on the device in double (ppp_spline_create / _restart / _eval)."""

    def __init__(self, y, x, z, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        y, x, z = (np.ascontiguousarray(a, np.float64) for a in (y, x, z))
        rc = self.L.ppp_spline_create(int(device), len(y), _d(y), _d(x), _d(z), C.byref(self.h))
        if rc:
            self.h = None
            raise PPPError(rc, "ppp_spline_create (GSL_EINVAL: fewer than 3 knots or y not strictly increasing)" if rc == ERR_ARG else "ppp_spline_create")

    def restart(self, y, x, z):
        y, x, z = (np.ascontiguousarray(a, np.float64) for a in (y, x, z))
        rc = self.L.ppp_spline_restart(self.h, len(y), _d(y), _d(x), _d(z))
        if rc:
            raise PPPError(rc, "ppp_spline_restart")

    def point(self, y):
        """(rc, [k, 3]): Spline::point for every y; rc = ERR_DOMAIN when a y lies outside [miny, bigy] (NaN rows)."""
        y = np.ascontiguousarray(np.atleast_1d(y), np.float64)
        out = np.empty((len(y), 3))
        rc = self.L.ppp_spline_eval(self.h, _d(y), len(y), _d(out))
        if rc and rc != ERR_DOMAIN:
            raise PPPError(rc, "ppp_spline_eval")
        return rc, out

    def range(self):
        a, b, n = C.c_double(), C.c_double(), C.c_size_t()
        self.L.ppp_spline_range(self.h, C.byref(a), C.byref(b), C.byref(n))
        return a.value, b.value, n.value

    def close(self):
        if self.h:
            self.L.ppp_spline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One planner handle on one MI355X (one HIP stream, one resident cloud)."""

    def __init__(self, device=0, params=None, fast_path=True, **kw):
        self.L = lib()
        self.h = C.c_void_p()
        rc = self.L.ppp_create(int(device), C.byref(self.h))
        if rc:
            self.h = None
            raise PPPError(rc, "ppp_create failed (no gfx950 device? there is no CPU fallback)")
        self.params = params if params is not None else default_params(**kw)
        self._chk(self.L.ppp_set_params(self.h, C.byref(self.params)))
        if not fast_path:
            self.set_fast_path(False)

    def set_fast_path(self, on=True):
        """ppp_set_fast_path: False keeps this handle on the slab-index launch sequence (the window path is the default where it applies)."""
        self._chk(self.L.ppp_set_fast_path(self.h, 1 if on else 0))

    def set_plan_reuse(self, on=True):
        """ppp_set_plan_reuse: False makes every new cloud take its own window census (the default lets a cloud of the same size and
        parameters inherit the capacities of the handle's earlier plan and skip that launch)."""
        self._chk(self.L.ppp_set_plan_reuse(self.h, 1 if on else 0))

    def set_side_by_side(self, handles):
        """ppp_set_side_by_side: how many handles the caller runs side by side on this device (from two on: narrower slice workgroups where
        the windows are small, room for the neighbouring passes' launches)."""
        self._chk(self.L.ppp_set_side_by_side(self.h, int(handles)))

    def fast_path(self):
        """True when the current plan runs the window path (three launches), False for the slab-index path."""
        a = C.c_int()
        self._chk(self.L.ppp_get_fast_path(self.h, C.byref(a)))
        return bool(a.value)

    def _chk(self, rc):
        if rc:
            raise PPPError(rc, self.L.ppp_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.ppp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, **kw):
        for k, v in kw.items():
            if k == "handeye":
                for j, x in enumerate(v):
                    self.params.handeye[j] = x
            else:
                setattr(self.params, k, v)
        self._chk(self.L.ppp_set_params(self.h, C.byref(self.params)))

    # -- cloud (constructor of the reference classes) --
    def set_cloud(self, xyz, viewpoint=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        assert xyz.ndim == 2 and xyz.shape[1] >= 3
        vp = None if viewpoint is None else _f(np.ascontiguousarray(viewpoint, np.float32))
        self._chk(self.L.ppp_set_cloud(self.h, xyz.ctypes.data, xyz.shape[0], xyz.shape[1] * 4, vp))
        self.n = xyz.shape[0]

    def set_cloud_pcd(self, path):
        """The constructors' loadPCDFile + scale loop in one call: the file goes straight to HBM.  Returns (n, viewpoint[7])."""
        n = C.c_size_t()
        vp = np.zeros(7, np.float32)
        self._chk(self.L.ppp_set_cloud_pcd(self.h, path.encode(), C.byref(n), _f(vp)))
        self.n = n.value
        return n.value, vp

    def range_interval(self, min_x, max_x):
        """(lo, hi, S): the planner-unit x interval this handle's slice range indexes for a cloud with these x bounds."""
        lo, hi, S = C.c_float(), C.c_float(), C.c_int()
        self._chk(self.L.ppp_range_interval(C.byref(self.params), float(min_x), float(max_x), C.byref(lo), C.byref(hi), C.byref(S)))
        return lo.value, hi.value, S.value

    def set_cloud_part(self, xyz, cloud_index, mn, mx, n_valid_total, part_lo, part_hi, viewpoint=None):
        """ppp_set_cloud_part: only this handle's part of the cloud + the whole cloud's bounds / count (planner units)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        vp = None if viewpoint is None else _f(np.ascontiguousarray(viewpoint, np.float32))
        idx = None if cloud_index is None else _i(np.ascontiguousarray(cloud_index, np.int32))
        mn = np.ascontiguousarray(mn, np.float32); mx = np.ascontiguousarray(mx, np.float32)
        self._keep = (xyz, cloud_index)
        self._chk(self.L.ppp_set_cloud_part(self.h, xyz.ctypes.data, xyz.shape[0], xyz.shape[1] * 4, vp, idx, _f(mn), _f(mx),
                                            int(n_valid_total), float(part_lo), float(part_hi)))
        self.n = xyz.shape[0]

    def set_cloud_device_async(self, dptr, n, stride_bytes, viewpoint=None):
        """ppp_set_cloud_device_async: no wait for the conversion pass where the handle's plan allows; the buffer must stay untouched
        until a call that waits (sync, any getter) has returned."""
        vp = None if viewpoint is None else _f(np.ascontiguousarray(viewpoint, np.float32))
        self._chk(self.L.ppp_set_cloud_device_async(self.h, C.c_void_p(dptr), n, stride_bytes, vp))
        self.n = n

    def set_cloud_device(self, dptr, n, stride_bytes, viewpoint=None):
        vp = None if viewpoint is None else _f(np.ascontiguousarray(viewpoint, np.float32))
        self._chk(self.L.ppp_set_cloud_device(self.h, C.c_void_p(dptr), n, stride_bytes, vp))
        self.n = n

    # -- hot path --
    def gen_path_async(self):
        self._chk(self.L.ppp_gen_path_async(self.h))

    def get_path_async(self):
        self._chk(self.L.ppp_get_path_async(self.h))

    def run_async(self):
        """GenPath() + getPath() as one enqueue (captured hipGraph)."""
        self._chk(self.L.ppp_run_async(self.h))

    def sync(self):
        self._chk(self.L.ppp_sync(self.h))

    def gen_path(self):
        """GenPath(): returns S."""
        self.gen_path_async()
        self.sync()
        return self.num_slices()

    def get_path(self):
        """getPath(): returns W."""
        self.get_path_async()
        self.sync()
        return self.num_waypoints()

    def failed_slice(self):
        return self.L.ppp_failed_slice(self.h)

    # -- results --
    def num_slices(self):
        s = C.c_int()
        self._chk(self.L.ppp_num_slices(self.h, C.byref(s)))
        return s.value

    def num_waypoints(self):
        w = C.c_size_t()
        self._chk(self.L.ppp_num_waypoints(self.h, C.byref(w)))
        return w.value

    def waypoints(self):
        W = self.num_waypoints()
        out = np.empty((W, 6), np.float32)
        w = C.c_size_t()
        self._chk(self.L.ppp_get_waypoints(self.h, _f(out), W, C.byref(w)))
        return out

    def waypoints_device(self):
        p = C.c_void_p()
        w = C.c_size_t()
        self._chk(self.L.ppp_get_waypoints_device(self.h, C.byref(p), C.byref(w)))
        return p.value, w.value

    def copy_waypoints_to_device(self, dptr, cap):
        w = C.c_size_t()
        self._chk(self.L.ppp_copy_waypoints_to_device(self.h, C.c_void_p(dptr), cap, C.byref(w)))
        return w.value

    def cloud(self):
        """The resident cloud (scaled, preprocessed) as float32 [n, 3] in index order."""
        n = C.c_size_t()
        self._chk(self.L.ppp_get_cloud(self.h, None, 0, C.byref(n)))
        out = np.empty((max(n.value, 1), 3), np.float32)
        self._chk(self.L.ppp_get_cloud(self.h, _f(out), n.value, C.byref(n)))
        return out[:n.value]

    def remove_outlier(self, mean_k=50, stddev_mul=1.0):
        """SectPath::remove_outlier (pcl::StatisticalOutlierRemoval) on the resident cloud; returns (new size, threshold)."""
        n = C.c_size_t()
        thr = C.c_double()
        self._chk(self.L.ppp_remove_outlier(self.h, int(mean_k), float(stddev_mul), C.byref(n), C.byref(thr)))
        return n.value, thr.value

    def voxel_down(self, lx, ly, lz):
        """path_generater::voxel_down (pcl::VoxelGrid) on the resident cloud; returns (new size, overflow flag)."""
        n = C.c_size_t()
        ov = C.c_int()
        self._chk(self.L.ppp_voxel_down(self.h, float(lx), float(ly), float(lz), C.byref(n), C.byref(ov)))
        return n.value, bool(ov.value)

    def smooth_mls(self, radius=15.0, order=3):
        """SectPath::smooth (pcl::MovingLeastSquares) on the resident cloud; returns the new size."""
        n = C.c_size_t()
        self._chk(self.L.ppp_smooth_mls(self.h, float(radius), int(order), C.byref(n)))
        return n.value

    def trans2center(self):
        """SectPath::trans2center on the resident cloud; returns (TransAlign 4x4, centroid, accumulated covariance 3x3)."""
        T = np.zeros(16, np.float32); c = np.zeros(3, np.float32); cov = np.zeros(9, np.float32)
        self._chk(self.L.ppp_trans2center(self.h, _f(T), _f(c), _f(cov)))
        return T.reshape(4, 4), c, cov.reshape(3, 3)

    def gather_waypoints(self, comm_ptr, rank, nranks, root, counts, recv_ptr):
        """ppp_gather_waypoints: the finished lists of all ranks to `root` over RCCL (comm_ptr = ncclComm_t)."""
        c = (C.c_size_t * nranks)(*[int(x) for x in counts])
        self._chk(self.L.ppp_gather_waypoints(self.h, C.c_void_p(comm_ptr), rank, nranks, root, c, C.c_void_p(recv_ptr)))

    def stream_ptr(self):
        """hipStream_t of this handle as an integer (torch.cuda.ExternalStream(ptr) orders framework work behind it)."""
        p = C.c_void_p()
        self._chk(self.L.ppp_get_stream(self.h, C.byref(p)))
        return p.value

    def waypoint_counts(self):
        """Waypoints per kept slice in list order (zero outside this handle's slice range)."""
        n = C.c_size_t()
        self._chk(self.L.ppp_get_waypoint_counts(self.h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.int32)
        self._chk(self.L.ppp_get_waypoint_counts(self.h, _i(out), n.value, C.byref(n)))
        return out[:n.value]

    def copy_stage_to_device(self, stage, dptr, cap):
        """D2D copy of the pre-smoothing (or smoothed) W x 6 list into a caller-owned device buffer; returns W."""
        w = C.c_size_t()
        self._chk(self.L.ppp_copy_stage_to_device(self.h, stage, C.c_void_p(dptr), cap, C.byref(w)))
        return w.value

    def finish_path_async(self, dptr, W, counts):
        """postion_smooth + reduceRPY + TransFlangeposition over a gathered pre-smoothing list in device memory
        (SURVEY.md 8e case ii: the blocks of the slice-range handles, concatenated in slice order)."""
        counts = np.ascontiguousarray(counts, np.int32)
        self._chk(self.L.ppp_finish_path_async(self.h, C.c_void_p(dptr), int(W), _i(counts), counts.size))

    def tail_index(self):
        n = C.c_size_t()
        self._chk(self.L.ppp_get_tail_index(self.h, None, 0, C.byref(n)))
        out = np.empty(max(n.value, 1), np.int32)
        self._chk(self.L.ppp_get_tail_index(self.h, _i(out), n.value, C.byref(n)))
        return out[:n.value]

    def minmax(self):
        mn = np.empty(3, np.float32)
        mx = np.empty(3, np.float32)
        self._chk(self.L.ppp_minmax(self.h, _f(mn), _f(mx)))
        return mn, mx

    def slice_positions(self):
        S = C.c_size_t()
        self._chk(self.L.ppp_get_slice_positions(self.h, None, 0, C.byref(S)))
        px = np.empty(max(S.value, 1), np.float32)
        self._chk(self.L.ppp_get_slice_positions(self.h, _f(px), S.value, C.byref(S)))
        return px[:S.value]

    def slice_indices(self, s):
        cap = 4096
        while True:
            out = np.empty(cap, np.int32)
            n = C.c_size_t()
            self._chk(self.L.ppp_get_slice_indices(self.h, int(s), _i(out), cap, C.byref(n)))
            if n.value <= cap:
                return out[:n.value].copy()
            cap = n.value

    def ranged_x_index(self, position):
        cap = 4096
        while True:
            out = np.empty(cap, np.int32)
            n = C.c_size_t()
            self._chk(self.L.ppp_ranged_x_index(self.h, int(position), _i(out), cap, C.byref(n)))
            if n.value <= cap:
                return out[:n.value].copy()
            cap = n.value

    def nodes(self, s):
        m = C.c_size_t()
        self._chk(self.L.ppp_get_nodes(self.h, int(s), None, None, None, 0, C.byref(m)))
        k = max(m.value, 1)
        y = np.empty(k); x = np.empty(k); z = np.empty(k)
        self._chk(self.L.ppp_get_nodes(self.h, int(s), _d(y), _d(x), _d(z), m.value, C.byref(m)))
        return y[:m.value], x[:m.value], z[:m.value]

    def boundary(self, s):
        """(y, x, z, step): knots of the boundary spline slice s was adjusted against (empty: none) and the slice's step in its chain"""
        m = C.c_size_t(); step = C.c_int()
        self._chk(self.L.ppp_get_boundary(self.h, int(s), None, None, None, 0, C.byref(m), C.byref(step)))
        k = max(m.value, 1)
        y = np.empty(k); x = np.empty(k); z = np.empty(k)
        self._chk(self.L.ppp_get_boundary(self.h, int(s), _d(y), _d(x), _d(z), m.value, C.byref(m), C.byref(step)))
        return y[:m.value], x[:m.value], z[:m.value], step.value

    def eval_spline(self, s, y):
        y = np.ascontiguousarray(y, np.float64)
        out = np.empty((len(y), 3))
        rc = self.L.ppp_eval_spline(self.h, int(s), _d(y), len(y), _d(out))
        if rc and rc != ERR_DOMAIN:
            self._chk(rc)
        return rc, out

    def insert_point(self, indices, plane_x):
        indices = np.ascontiguousarray(indices, np.int32)
        cap = max(len(indices), 1)
        y = np.empty(cap); x = np.empty(cap); z = np.empty(cap)
        m = C.c_size_t()
        rc = self.L.ppp_insert_point(self.h, _i(indices), len(indices), float(plane_x), _d(y), _d(x), _d(z), cap, C.byref(m))
        if rc == ERR_SLICE:
            return rc, None, None, None
        self._chk(rc)
        return m.value, y[:m.value], x[:m.value], z[:m.value]

    def normals_at(self, idx):
        idx = np.ascontiguousarray(idx, np.int32)
        out = np.empty((len(idx), 4), np.float32)
        self._chk(self.L.ppp_normals_at(self.h, _i(idx), len(idx), _f(out)))
        return out

    def estimate_normals(self):
        """estimate_normal() over the whole cloud: [n, 4] = nx ny nz curvature."""
        out = np.empty((self.n, 4), np.float32)
        self._chk(self.L.ppp_estimate_normals(self.h, _f(out)))
        return out

    def area2cloud(self, pts, key):
        """Area2Cloud of the dynamic adjustment for points [k, 3] (float64, mm)."""
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        out = np.empty((len(pts), 3), np.float32)
        self._chk(self.L.ppp_area2cloud(self.h, _d(pts), len(pts), int(key), _f(out)))
        return out

    def nearest(self, q):
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
        out = np.empty(len(q), np.int32)
        self._chk(self.L.ppp_nearest(self.h, _f(q), len(q), _i(out)))
        return out

    def stage(self, stage):
        cnt = C.c_size_t()
        self._chk(self.L.ppp_get_stage(self.h, stage, None, 0, C.byref(cnt)))
        W = cnt.value
        shape, dt = {STAGE_WP_XYZ: ((W, 3), np.float32), STAGE_WP_NN: ((W,), np.int32),
                     STAGE_WP_NORMAL: ((W, 4), np.float32), STAGE_WP_PRESMOOTH: ((W, 6), np.float32),
                     STAGE_WP_SMOOTHED: ((W, 6), np.float32)}[stage]
        out = np.empty(shape, dt)
        if W:
            self._chk(self.L.ppp_get_stage(self.h, stage, out.ctypes.data, out.nbytes, C.byref(cnt)))
        return out

    def smooth_sweeps(self):
        s = C.c_int()
        self._chk(self.L.ppp_smooth_sweeps(self.h, C.byref(s)))
        return s.value

    # -- measurement --
    def enable_timing(self, on=True):
        self._chk(self.L.ppp_enable_timing(self.h, 1 if on else 0))

    def kernel_times(self, with_launches=False):
        """{kernel: ms summed over its launches since the last call} (and {kernel: launches})."""
        cap = 64
        names = C.create_string_buffer(48 * cap)
        ms = np.zeros(cap, np.float32)
        cnt = np.zeros(cap, np.int32)
        n = C.c_size_t()
        self._chk(self.L.ppp_get_kernel_times(self.h, names, _f(ms), _i(cnt), cap, C.byref(n)))
        out, launches = {}, {}
        for k in range(n.value):
            nm = names.raw[48 * k:48 * k + 48].split(b"\0", 1)[0].decode()
            out[nm] = float(ms[k])
            launches[nm] = int(cnt[k])
        return (out, launches) if with_launches else out
