"""ctypes access to the CPU oracle (oracle/slam_oracle.c) and, when it has been
built, to the compiled reference Matrix class (oracle/_ref).

TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (slam_amd/) never touches it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# SLAM_ORACLE_SANITIZED=1 (tests/test_oracle_sanitized.py): the same C compiled with -fsanitize=address,undefined
# (make -C oracle asan); the interpreter must then run with libasan preloaded
SANITIZED = os.environ.get("SLAM_ORACLE_SANITIZED") == "1"
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libslam_oracle_asan.so" if SANITIZED else "libslam_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libslam_ref_matrix.so")

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_i32p = C.POINTER(C.c_int32)
_i8p = C.POINTER(C.c_int8)


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("slam_oracle.c", "gseg_oracle.c", "ccicp_oracle.c", "slam_oracle.h")]
    if (force or not os.path.exists(ORACLE_SO)
            or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(f) for f in srcs)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "asan" if SANITIZED else "oracle"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


class OIcpParams(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("min_delta", C.c_double), ("indist", C.c_double),
                ("nn_method", C.c_int), ("mode", C.c_int)]


class OGsegParams(C.Structure):
    _fields_ = [("rmax", C.c_double), ("num_seedpoints", C.c_int), ("p_l", C.c_double), ("p_sf", C.c_double),
                ("p_sn", C.c_double), ("p_tmodel", C.c_double), ("p_tdata", C.c_double), ("p_tg", C.c_double),
                ("robot_height", C.c_double), ("max_seed_range", C.c_double), ("max_seed_height", C.c_double)]


GSEG_DROPPED, GSEG_GROUND, GSEG_OBSTACLE, GSEG_OVERHEAD = 0, 1, 2, 3


class OGridParams(C.Structure):
    _fields_ = [("size_x", C.c_int), ("size_y", C.c_int), ("resolution", C.c_double),
                ("max_range", C.c_double), ("occupancy_increment", C.c_double),
                ("occupancy_decrement", C.c_double), ("min_cluster_points", C.c_int),
                ("rolling", C.c_int), ("pose_x", C.c_double), ("pose_y", C.c_double)]


NN_KDTREE, NN_BRUTE = 0, 1
MODE_P2P, MODE_P2L = 0, 1

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build_oracle()
    L = C.CDLL(ORACLE_SO)
    L.okd_build.restype = C.c_void_p
    L.okd_build.argtypes = [_fp, C.c_int]
    L.okd_free.argtypes = [C.c_void_p]
    L.okd_nn1.argtypes = [C.c_void_p, C.c_float, C.c_float, _fp, _ip]
    L.obf_nn1.argtypes = [_fp, C.c_int, C.c_float, C.c_float, _fp, _ip]
    L.obf_knn.argtypes = [_fp, C.c_int, C.c_float, C.c_float, C.c_int, _ip]
    L.o_p2p_rotation.argtypes = [_dp, _dp]
    L.o_solve3.restype = C.c_int
    L.o_solve3.argtypes = [_dp, _dp]
    L.o_orthonormal_from_omega.argtypes = [C.c_double, _dp]
    L.oicp_create.restype = C.c_void_p
    L.oicp_create.argtypes = [_dp, C.c_int, _dp, C.c_int]
    L.oicp_free.argtypes = [C.c_void_p]
    L.o_normal2.argtypes = [_dp, C.c_int, _dp]
    L.oicp_compute_normals.argtypes = [C.c_void_p, C.c_int]
    L.oicp_normals.restype = _dp
    L.oicp_normals.argtypes = [C.c_void_p]
    L.oicp_fit_step.restype = C.c_double
    L.oicp_fit_step.argtypes = [C.c_void_p, _dp, C.c_int, _dp, C.c_int, _dp, _dp,
                                C.POINTER(OIcpParams), _ip, _ip]
    L.oicp_fit.restype = C.c_int
    L.oicp_fit.argtypes = [C.c_void_p, _dp, C.c_int, _dp, C.c_int, _dp, _dp,
                           C.POINTER(OIcpParams), _dp, _ip, _dp]
    L.oicp_fit_batch.argtypes = [C.c_void_p, _dp, _ip, _ip, C.c_int, _dp, _dp,
                                 C.POINTER(OIcpParams), _ip, _ip, _dp, C.c_int]
    L.oicp_edge_weight.argtypes = [_dp, _dp, C.c_int, _dp]
    L.ogrid_cell.restype = C.c_int
    L.ogrid_cell.argtypes = [C.POINTER(OGridParams), C.c_float, C.c_float, _ip, _ip]
    L.ogrid_add_endpoints.restype = C.c_long
    L.ogrid_add_endpoints.argtypes = [C.POINTER(OGridParams), _fp, C.c_int, _fp, C.c_int,
                                      C.c_int, _i32p, _i32p, _ip]
    L.ogrid_raycast.restype = C.c_long
    L.ogrid_raycast.argtypes = [C.POINTER(OGridParams), _fp, _fp, C.c_int, _i32p, _i32p]
    L.ogrid_raycast_mt.restype = C.c_long
    L.ogrid_raycast_mt.argtypes = [C.POINTER(OGridParams), _fp, _fp, C.c_int, _i32p, _i32p, C.c_int]
    L.o_transform_points.argtypes = [_dp, C.c_int, _dp, _dp, _fp]
    L.ogrid_finalize.argtypes = [C.POINTER(OGridParams), _i32p, _i32p, _dp, _i8p]
    L.ogrid_add_scan_inorder.argtypes = [C.POINTER(OGridParams), _fp, C.c_int, _fp, C.c_int,
                                         C.c_int, _dp, _i8p, _i8p]
    L.ogseg_default_params.argtypes = [C.POINTER(OGsegParams)]
    L.ogseg_segment.restype = C.c_int
    L.ogseg_segment.argtypes = [C.POINTER(OGsegParams), _fp, C.c_int, C.c_int, C.POINTER(C.c_ubyte), _ip,
                                C.POINTER(C.c_ubyte), _dp]
    L.occicp_classify.argtypes = [_fp, C.c_int, C.c_int, C.POINTER(C.c_ubyte)]
    _lib = L
    return L


def _d(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    return a.ctypes.data_as(_fp)


def _i(a):
    return a.ctypes.data_as(_ip)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ----------------------------------------------------------------- kd / NN
class KdTree:
    def __init__(self, xy_f32):
        self.xy = as_f32(xy_f32).reshape(-1, 2)
        self.h = lib().okd_build(_f(self.xy), len(self.xy))

    def nn1(self, qx, qy):
        d, i = C.c_float(), C.c_int()
        lib().okd_nn1(self.h, np.float32(qx), np.float32(qy), C.byref(d), C.byref(i))
        return d.value, i.value

    def __del__(self):
        if getattr(self, "h", None):
            lib().okd_free(self.h)
            self.h = None


def brute_nn1(xy_f32, qx, qy):
    xy = as_f32(xy_f32).reshape(-1, 2)
    d, i = C.c_float(), C.c_int()
    lib().obf_nn1(_f(xy), len(xy), np.float32(qx), np.float32(qy), C.byref(d), C.byref(i))
    return d.value, i.value


# ------------------------------------------------------------------ solves
def p2p_rotation(H):
    H = as_f64(H).reshape(4)
    out = np.zeros(4)
    lib().o_p2p_rotation(_d(H), _d(out))
    return out.reshape(2, 2)


def solve3(A, b):
    A = as_f64(A).reshape(9).copy()
    b = as_f64(b).reshape(3).copy()
    ok = lib().o_solve3(_d(A), _d(b))
    return ok, b


def normal2(nb_xy):
    nb = as_f64(nb_xy).reshape(-1, 2)
    out = np.zeros(2)
    lib().o_normal2(_d(nb), len(nb), _d(out))
    return out


def orthonormal_from_omega(w):
    out = np.zeros(4)
    lib().o_orthonormal_from_omega(float(w), _d(out))
    return out.reshape(2, 2)


# --------------------------------------------------------------------- ICP
def icp_params(max_iter=20, min_delta=1e-6, indist=5.0, nn_method=NN_KDTREE, mode=MODE_P2P):
    return OIcpParams(max_iter, min_delta, indist, nn_method, mode)


class IcpModel:
    """Icp::Icp (icp.cpp:26-70) on the oracle."""

    def __init__(self, m_ga, m_nga, normals_k=0):
        self.m_ga = as_f64(m_ga).reshape(-1, 2)
        self.m_nga = as_f64(m_nga).reshape(-1, 2)
        self.h = lib().oicp_create(_d(self.m_ga), len(self.m_ga), _d(self.m_nga), len(self.m_nga))
        if self.h and normals_k:
            lib().oicp_compute_normals(self.h, normals_k)

    @property
    def valid(self):
        return bool(self.h)

    def normals(self):
        n = len(self.m_ga) + len(self.m_nga)
        p = lib().oicp_normals(self.h)
        return np.ctypeslib.as_array(p, shape=(n, 2)).copy()

    def fit_step(self, t_ga, t_nga, R, t, params):
        t_ga = as_f64(t_ga).reshape(-1, 2)
        t_nga = as_f64(t_nga).reshape(-1, 2)
        R = as_f64(R).reshape(4).copy()
        t = as_f64(t).reshape(2).copy()
        nc = C.c_int()
        corr = np.full(len(t_ga) + len(t_nga), -1, dtype=np.int32)
        d = lib().oicp_fit_step(self.h, _d(t_ga), len(t_ga), _d(t_nga), len(t_nga), _d(R), _d(t),
                                C.byref(params), C.byref(nc), _i(corr))
        return d, R.reshape(2, 2), t, nc.value, corr

    def fit(self, t_ga, t_nga, R, t, params):
        """Icp::fit (icp.cpp:80-114). Returns R, t, trace[steps,8], steps."""
        t_ga = as_f64(t_ga).reshape(-1, 2)
        t_nga = as_f64(t_nga).reshape(-1, 2)
        R = as_f64(R).reshape(4).copy()
        t = as_f64(t).reshape(2).copy()
        trace = np.zeros((max(params.max_iter, 1), 8))
        nc, dl = C.c_int(), C.c_double()
        steps = lib().oicp_fit(self.h, _d(t_ga), len(t_ga), _d(t_nga), len(t_nga), _d(R), _d(t),
                               C.byref(params), _d(trace), C.byref(nc), C.byref(dl))
        return R.reshape(2, 2), t, trace[:steps].copy(), steps

    def fit_batch(self, pts, scan_off, scan_nga, R, t, params, n_threads=0):
        pts = as_f64(pts).reshape(-1, 2)
        scan_off = np.ascontiguousarray(scan_off, dtype=np.int32)
        scan_nga = np.ascontiguousarray(scan_nga, dtype=np.int32)
        n = len(scan_nga)
        R = as_f64(R).reshape(n, 4).copy()
        t = as_f64(t).reshape(n, 2).copy()
        iters = np.zeros(n, dtype=np.int32)
        ncorr = np.zeros(n, dtype=np.int32)
        delta = np.zeros(n)
        lib().oicp_fit_batch(self.h, _d(pts), _i(scan_off), _i(scan_nga), n, _d(R), _d(t),
                             C.byref(params), _i(iters), _i(ncorr), _d(delta), int(n_threads))
        return R, t, iters, ncorr, delta

    def __del__(self):
        if getattr(self, "h", None):
            lib().oicp_free(self.h)
            self.h = None


def edge_weight(pm, pt):
    pm = as_f64(pm).reshape(-1, 2)
    pt = as_f64(pt).reshape(-1, 2)
    out = np.zeros(9)
    lib().oicp_edge_weight(_d(pm), _d(pt), len(pm), _d(out))
    return out.reshape(3, 3)


# -------------------------------------------------------------------- grid
def grid_params(size_x, size_y, resolution, max_range=75.0, inc=1.0, dec=0.3,
                min_cluster_points=10, rolling=0, pose_x=0.0, pose_y=0.0):
    return OGridParams(size_x, size_y, resolution, max_range, inc, dec, min_cluster_points,
                       rolling, pose_x, pose_y)


def grid_cell(g, px, py):
    cx, cy = C.c_int(-1), C.c_int(-1)
    c = lib().ogrid_cell(C.byref(g), np.float32(px), np.float32(py), C.byref(cx), C.byref(cy))
    return c, cx.value, cy.value


def grid_add_endpoints(g, obs, gnd, hits=None, misses=None):
    obs = as_f32(obs)
    gnd = as_f32(gnd)
    stride = obs.shape[1] if obs.ndim == 2 and obs.size else (gnd.shape[1] if gnd.ndim == 2 and gnd.size else 2)
    n_obs = obs.size // stride
    n_gnd = gnd.size // stride
    cells = g.size_x * g.size_y
    if hits is None:
        hits = np.zeros(cells, dtype=np.int32)
    if misses is None:
        misses = np.zeros(cells, dtype=np.int32)
    cell_out = np.zeros(max(n_obs + n_gnd, 1), dtype=np.int32)
    n = lib().ogrid_add_endpoints(C.byref(g), _f(obs), n_obs, _f(gnd), n_gnd, stride,
                                  hits.ctypes.data_as(_i32p), misses.ctypes.data_as(_i32p),
                                  _i(cell_out))
    return hits, misses, cell_out[:n_obs + n_gnd], n


def grid_raycast(g, origin_xy, end_xy, hits=None, misses=None, n_threads=None):
    origin_xy = as_f32(origin_xy).reshape(-1, 2)
    end_xy = as_f32(end_xy).reshape(-1, 2)
    cells = g.size_x * g.size_y
    if hits is None:
        hits = np.zeros(cells, dtype=np.int32)
    if misses is None:
        misses = np.zeros(cells, dtype=np.int32)
    if n_threads is None:
        n = lib().ogrid_raycast(C.byref(g), _f(origin_xy), _f(end_xy), len(end_xy),
                                hits.ctypes.data_as(_i32p), misses.ctypes.data_as(_i32p))
    else:
        n = lib().ogrid_raycast_mt(C.byref(g), _f(origin_xy), _f(end_xy), len(end_xy),
                                   hits.ctypes.data_as(_i32p), misses.ctypes.data_as(_i32p), int(n_threads))
    return hits, misses, n


def transform_points(pts, R, t):
    pts = as_f64(pts).reshape(-1, 2)
    R = as_f64(R).reshape(4)
    t = as_f64(t).reshape(2)
    out = np.zeros((len(pts), 2), dtype=np.float32)
    lib().o_transform_points(_d(pts), len(pts), _d(R), _d(t), _f(out))
    return out


def grid_finalize(g, hits, misses, num_pts=None, occ=None):
    cells = g.size_x * g.size_y
    if num_pts is None:
        num_pts = np.zeros(cells)
    if occ is None:
        occ = np.full(cells, -1, dtype=np.int8)
    hits = np.ascontiguousarray(hits, dtype=np.int32)
    misses = np.ascontiguousarray(misses, dtype=np.int32)
    lib().ogrid_finalize(C.byref(g), hits.ctypes.data_as(_i32p), misses.ctypes.data_as(_i32p),
                         _d(num_pts), occ.ctypes.data_as(_i8p))
    return num_pts, occ


def grid_add_scan_inorder(g, obs, gnd, num_pts, drivable, occ):
    obs = as_f32(obs)
    gnd = as_f32(gnd)
    stride = obs.shape[1] if obs.ndim == 2 and obs.size else (gnd.shape[1] if gnd.ndim == 2 and gnd.size else 2)
    lib().ogrid_add_scan_inorder(C.byref(g), _f(obs), obs.size // stride, _f(gnd),
                                 gnd.size // stride, stride, _d(num_pts),
                                 drivable.ctypes.data_as(_i8p), occ.ctypes.data_as(_i8p))


# ------------------------------------------------------- ground segmentation
def gseg_params(**kw):
    p = OGsegParams()
    lib().ogseg_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def gseg_segment(xyz, params=None):
    """groundSegmentation::segmentGround on the oracle.  xyz: [n, stride>=3] f32.
    Returns labels[n] (GSEG_*), bin_of[n], state[72*200], value[72*200], outer iterations."""
    xyz = as_f32(xyz)
    n, stride = xyz.shape
    p = params or gseg_params()
    labels = np.zeros(max(n, 1), dtype=np.uint8)
    bin_of = np.zeros(max(n, 1), dtype=np.int32)
    state = np.zeros(72 * 200, dtype=np.uint8)
    value = np.zeros(72 * 200)
    it = lib().ogseg_segment(C.byref(p), _f(xyz), n, stride, labels.ctypes.data_as(C.POINTER(C.c_ubyte)),
                             _i(bin_of), state.ctypes.data_as(C.POINTER(C.c_ubyte)), _d(value))
    return labels[:n], bin_of[:n], state, value, it


def classify_ga(xyz):
    """CCICP::classifyPoints (icpTools.cpp:36-103): flags[n] in {0 NGA, 1 GA, 255 dropped}."""
    xyz = as_f32(xyz)
    n, stride = xyz.shape
    flags = np.zeros(max(n, 1), dtype=np.uint8)
    lib().occicp_classify(_f(xyz), n, stride, flags.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return flags[:n]


# --------------------------------------------------- compiled reference Matrix
_ref = None


def ref_available():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(REF_SO)
        R.ref_svd2.argtypes = [_dp, _dp, _dp, _dp]
        R.ref_p2p_rotation.argtypes = [_dp, _dp]
        R.ref_fitstep_solve.restype = C.c_double
        R.ref_fitstep_solve.argtypes = [_dp, _dp, C.c_int, _dp, _dp]
        R.ref_solve3.restype = C.c_int
        R.ref_solve3.argtypes = [_dp, _dp, _dp]
        R.ref_inv3.argtypes = [_dp, _dp]
        R.ref_orthonormal_from_omega.argtypes = [C.c_double, _dp]
        R.ref_normal2.argtypes = [_dp, C.c_int, _dp]
        _ref = R
    return _ref


def ref_p2p_rotation(H):
    H = as_f64(H).reshape(4)
    out = np.zeros(4)
    ref().ref_p2p_rotation(_d(H), _d(out))
    return out.reshape(2, 2)


def ref_svd2(H):
    H = as_f64(H).reshape(4)
    U, W, V = np.zeros(4), np.zeros(2), np.zeros(4)
    ref().ref_svd2(_d(H), _d(U), _d(W), _d(V))
    return U.reshape(2, 2), W, V.reshape(2, 2)


def ref_fitstep_solve(pm, pt, R, t):
    pm = as_f64(pm).reshape(-1, 2)
    pt = as_f64(pt).reshape(-1, 2)
    R = as_f64(R).reshape(4).copy()
    t = as_f64(t).reshape(2).copy()
    d = ref().ref_fitstep_solve(_d(pm), _d(pt), len(pm), _d(R), _d(t))
    return d, R.reshape(2, 2), t


def ref_solve3(A, b):
    A = as_f64(A).reshape(9)
    b = as_f64(b).reshape(3)
    x = np.zeros(3)
    ok = ref().ref_solve3(_d(A), _d(b), _d(x))
    return ok, x


def ref_normal2(nb_xy):
    nb = as_f64(nb_xy).reshape(-1, 2)
    out = np.zeros(2)
    ref().ref_normal2(_d(nb), len(nb), _d(out))
    return out


def ref_orthonormal_from_omega(w):
    out = np.zeros(4)
    ref().ref_orthonormal_from_omega(float(w), _d(out))
    return out.reshape(2, 2)


# ------------------------------------------------------- CCICP facade steps
def ccicp_crop(xyz, cur_x, cur_y, crop=75.0):
    xyz = as_f32(xyz)
    keep = np.zeros(max(len(xyz), 1), dtype=np.uint8)
    lib().occicp_crop.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p]
    lib().occicp_crop(xyz.ctypes.data, len(xyz), xyz.shape[1], cur_x, cur_y, crop, keep.ctypes.data)
    return keep[:len(xyz)].astype(bool)


def voxel_downsample(xyzg, leaf=(0.5, 0.5, 2.0)):
    xyzg = as_f32(xyzg)
    out = np.zeros((max(len(xyzg), 1), 4), dtype=np.float32)
    f = lib().ovoxel_downsample
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
    n = f(xyzg.ctypes.data, len(xyzg), xyzg.shape[1], leaf[0], leaf[1], leaf[2], out.ctypes.data)
    return out[:max(n, 0)], n


def ccicp_split(xyzg, keep=None, cap=20000):
    xyzg = as_f32(xyzg)
    ga = np.zeros((cap, 2))
    nga = np.zeros((cap, 2))
    na, nb = C.c_int(0), C.c_int(0)
    k = np.ascontiguousarray(keep, dtype=np.uint8) if keep is not None else None
    f = lib().occicp_split
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f.restype = None
    f(xyzg.ctypes.data, k.ctypes.data if k is not None else None, len(xyzg), xyzg.shape[1], cap, ga.ctypes.data,
      C.addressof(na), nga.ctypes.data, C.addressof(nb))
    return ga[:na.value], nga[:nb.value]


def ccicp_height(ground, pose7):
    g = as_f32(ground)
    pose = (C.c_double * 7)(*pose7)
    z = C.c_double(0.0)
    idx = (C.c_int * 4)()
    f = lib().occicp_height
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    nc = f(g.ctypes.data if len(g) else None, len(g), g.shape[1] if g.ndim == 2 else 3, pose, C.addressof(z), idx)
    return z.value, nc, list(idx)


def roll(plane, dx, dy, fill=0):
    """Window after Grid::shiftOrigin(dx,dy) + the clears of mls.cpp:433-477: cell (i,j) shows old (i+dx, j+dy)."""
    sy, sx = plane.shape
    out = np.full_like(plane, fill)
    xs = np.arange(sx) + dx
    ys = np.arange(sy) + dy
    vx = (xs >= 0) & (xs < sx)
    vy = (ys >= 0) & (ys < sy)
    out[np.ix_(vy, vx)] = plane[np.ix_(ys[vy], xs[vx])]
    return out
