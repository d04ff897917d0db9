"""TEST INFRASTRUCTURE -- ctypes binding of oracle/liboracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (cuda-go-icp_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so (and, when /root/reference is present, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


class DT(C.Structure):
    _fields_ = [("V", C.c_int), ("scale", C.c_double), ("xmin", C.c_double), ("ymin", C.c_double),
                ("zmin", C.c_double), ("grid", C.POINTER(C.c_float)), ("owns", C.c_int)]


class Result(C.Structure):
    _fields_ = [("R", C.c_float * 9), ("t", C.c_float * 3), ("sse", C.c_float),
                ("rot_pops", C.c_longlong), ("trans_pops", C.c_longlong), ("cubes", C.c_longlong),
                ("inner_calls", C.c_longlong), ("icp_runs", C.c_longlong), ("icp_iters", C.c_longlong)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        fp, ip, dp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(DT)
        L.orc_dt_geometry.argtypes = [fp, C.c_int, C.c_int, C.c_double, dp]
        L.orc_dt_build.argtypes = [fp, C.c_int, C.c_int, C.c_double, dp]
        L.orc_dt_build.restype = C.c_int
        L.orc_dt_wrap.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, fp]
        L.orc_dt_free.argtypes = [dp]
        L.orc_dt_distance.argtypes = [dp, C.c_double, C.c_double, C.c_double]
        L.orc_dt_distance.restype = C.c_float
        L.orc_dt_seed.argtypes = [dp, fp, C.c_int, C.POINTER(C.c_ubyte)]
        L.orc_dt_seed.restype = C.c_int
        L.orc_rot_radii.argtypes = [fp, C.c_int, fp, fp]
        L.orc_rot_coeff.argtypes = [C.c_int]
        L.orc_rot_coeff.restype = C.c_float
        L.orc_rodrigues.argtypes = [C.c_float, C.c_float, C.c_float, fp]
        L.orc_rotate.argtypes = [fp, fp, C.c_int, fp]
        L.orc_cube_bound.argtypes = [dp, fp, C.c_int, fp, C.c_float, C.c_float, C.c_float, C.c_float, fp, fp]
        L.orc_cube_bound_omp.argtypes = L.orc_cube_bound.argtypes
        L.orc_cube_bound_trim.argtypes = [dp, fp, C.c_int, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, fp, fp]
        L.orc_inner_bnb_trim.argtypes = [dp, fp, C.c_int, fp, C.c_int, C.c_float, C.c_float, fp, fp,
                                         C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.orc_inner_bnb_trim.restype = C.c_float
        L.orc_cube_bounds_batch.argtypes = [dp, fp, C.c_int, fp, fp, C.c_int, fp, fp, C.c_int]
        L.orc_dt_sse.argtypes = [dp, fp, C.c_int, fp, fp]
        L.orc_dt_sse.restype = C.c_float
        L.orc_inner_bnb.argtypes = [dp, fp, C.c_int, fp, C.c_float, C.c_float, fp, fp,
                                    C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.orc_inner_bnb.restype = C.c_float
        L.orc_kd_build.argtypes = [fp, C.c_int]
        L.orc_kd_build.restype = C.c_void_p
        L.orc_kd_free.argtypes = [C.c_void_p]
        L.orc_kd_nn.argtypes = [C.c_void_p, fp, ip, fp]
        L.orc_nn_brute.argtypes = [fp, C.c_int, fp, ip, fp]
        L.orc_kabsch_rotation.argtypes = [fp, fp]
        L.orc_icp_run.argtypes = [C.c_void_p, fp, fp, C.c_int, fp, fp, C.c_int, C.c_float, ip]
        L.orc_icp_run.restype = C.c_float
        L.orc_register.argtypes = [dp, fp, C.c_int, fp, C.c_int, C.c_float, C.POINTER(Result)]
        L.orc_register.restype = C.c_int
        L.orc_register_trim.argtypes = [dp, fp, C.c_int, fp, C.c_int, C.c_float, C.c_float, C.POINTER(Result)]
        L.orc_register_trim.restype = C.c_int
        L.orc_dt_sse_trim.argtypes = [dp, fp, C.c_int, fp, fp, C.c_int]
        L.orc_dt_sse_trim.restype = C.c_float
        L.orc_icp_run_trim.argtypes = [C.c_void_p, fp, fp, C.c_int, C.c_int, fp, fp, C.c_int, C.c_float, ip]
        L.orc_icp_run_trim.restype = C.c_float
        _LIB = L
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


class DistanceTransform:
    """DT3D (jly_3ddt.{h,cpp}): geometry + grid + lookup."""

    def __init__(self, model=None, V=300, expand=2.0, geometry_only=False):
        self.dt = DT()
        self._keep = None
        if model is not None:
            m, mp = _f(model)
            if geometry_only:
                lib().orc_dt_geometry(mp, len(m), V, expand, C.byref(self.dt))
            else:
                if lib().orc_dt_build(mp, len(m), V, expand, C.byref(self.dt)) != 0:
                    raise MemoryError("orc_dt_build")

    @classmethod
    def wrap(cls, V, scale, xmin, ymin, zmin, grid):
        self = cls()
        g, gp = _f(grid)
        assert g.size == V ** 3
        self._keep = g
        lib().orc_dt_wrap(C.byref(self.dt), V, scale, xmin, ymin, zmin, gp)
        return self

    @property
    def V(self):
        return self.dt.V

    @property
    def scale(self):
        return self.dt.scale

    @property
    def origin(self):
        return (self.dt.xmin, self.dt.ymin, self.dt.zmin)

    def grid(self):
        V = self.dt.V
        return np.ctypeslib.as_array(self.dt.grid, shape=(V, V, V))

    def seed(self, model):
        m, mp = _f(model)
        V = self.dt.V
        s = np.zeros(V ** 3, dtype=np.uint8)
        n = lib().orc_dt_seed(C.byref(self.dt), mp, len(m), s.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return s.reshape(V, V, V), n

    def distance(self, q):
        q = np.asarray(q, dtype=np.float64).reshape(-1, 3)
        L = lib()
        return np.array([L.orc_dt_distance(C.byref(self.dt), x, y, z) for x, y, z in q], dtype=np.float32)

    def __del__(self):
        try:
            lib().orc_dt_free(C.byref(self.dt))
        except Exception:
            pass


def rot_radii(data):
    d, dp = _f(data)
    n = len(d)
    norm = np.empty(n, np.float32)
    rho = np.empty((20, n), np.float32)
    lib().orc_rot_radii(dp, n, norm.ctypes.data_as(C.POINTER(C.c_float)), rho.ctypes.data_as(C.POINTER(C.c_float)))
    return norm, rho


def rot_coeff(level):
    return np.float32(lib().orc_rot_coeff(level))


def rodrigues(v):
    R = np.empty(9, np.float32)
    lib().orc_rodrigues(np.float32(v[0]), np.float32(v[1]), np.float32(v[2]), R.ctypes.data_as(C.POINTER(C.c_float)))
    return R.reshape(3, 3)


def rotate(R, data):
    R_, Rp = _f(np.asarray(R).reshape(9))
    d, dp = _f(data)
    out = np.empty_like(d)
    lib().orc_rotate(Rp, dp, len(d), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def cube_bound(dt, prot, rho, t, w_child, omp=False):
    p, pp = _f(prot)
    if rho is not None:
        rho, rp = _f(rho)
    else:
        rp = None
    ub, lb = C.c_float(), C.c_float()
    fn = lib().orc_cube_bound_omp if omp else lib().orc_cube_bound
    fn(C.byref(dt.dt), pp, len(p), rp, np.float32(t[0]), np.float32(t[1]), np.float32(t[2]), np.float32(w_child),
       C.byref(ub), C.byref(lb))
    return np.float32(ub.value), np.float32(lb.value)


def cube_bound_trim(dt, prot, rho, t, w_child, inliers):
    p, pp = _f(prot)
    rp = None
    if rho is not None:
        rho, rp = _f(rho)
    ub, lb = C.c_float(), C.c_float()
    lib().orc_cube_bound_trim(C.byref(dt.dt), pp, len(p), rp, float(np.float32(t[0])), float(np.float32(t[1])),
                              float(np.float32(t[2])), float(np.float32(w_child)), int(inliers), C.byref(ub), C.byref(lb))
    return np.float32(ub.value), np.float32(lb.value)


def inner_bnb_trim(dt, prot, rho, inliers, incumbent, sse_thresh, root=(-0.5, -0.5, -0.5, 1.0)):
    p, pp = _f(prot)
    rp = None
    if rho is not None:
        rho, rp = _f(rho)
    root_, rootp = _f(np.asarray(root))
    best = np.zeros(4, np.float32)
    pops, cubes = C.c_longlong(0), C.c_longlong(0)
    v = lib().orc_inner_bnb_trim(C.byref(dt.dt), pp, len(p), rp, int(inliers), float(np.float32(incumbent)),
                                 float(np.float32(sse_thresh)), rootp, best.ctypes.data_as(C.POINTER(C.c_float)),
                                 C.byref(pops), C.byref(cubes))
    return np.float32(v), best, pops.value, cubes.value


def cube_bounds_batch(dt, prot, rho, cubes4, parallel=False):
    p, pp = _f(prot)
    rp = None
    if rho is not None:
        rho, rp = _f(rho)
    c, cp = _f(np.asarray(cubes4).reshape(-1, 4))
    ub, lb = np.empty(len(c), np.float32), np.empty(len(c), np.float32)
    lib().orc_cube_bounds_batch(C.byref(dt.dt), pp, len(p), rp, cp, len(c), ub.ctypes.data_as(C.POINTER(C.c_float)),
                                lb.ctypes.data_as(C.POINTER(C.c_float)), 1 if parallel else 0)
    return ub, lb


def dt_sse(dt, data, R, t):
    d, dp = _f(data)
    R_, Rp = _f(np.asarray(R).reshape(9))
    t_, tp = _f(np.asarray(t).reshape(3))
    return np.float32(lib().orc_dt_sse(C.byref(dt.dt), dp, len(d), Rp, tp))


def inner_bnb(dt, prot, rho, incumbent, sse_thresh, root=(-0.5, -0.5, -0.5, 1.0)):
    p, pp = _f(prot)
    if rho is not None:
        rho, rp = _f(rho)
    else:
        rp = None
    root_, rootp = _f(np.asarray(root))
    best = np.zeros(4, np.float32)
    pops, cubes = C.c_longlong(0), C.c_longlong(0)
    v = lib().orc_inner_bnb(C.byref(dt.dt), pp, len(p), rp, np.float32(incumbent), np.float32(sse_thresh), rootp,
                            best.ctypes.data_as(C.POINTER(C.c_float)), C.byref(pops), C.byref(cubes))
    return np.float32(v), best, pops.value, cubes.value


class KdTree:
    def __init__(self, model):
        self.model, self.mp = _f(model)
        self.h = lib().orc_kd_build(self.mp, len(self.model))

    def nn(self, q):
        q, _ = _f(np.asarray(q).reshape(-1, 3))
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float32)
        L = lib()
        i, d = C.c_int(), C.c_float()
        for k in range(len(q)):
            L.orc_kd_nn(self.h, q[k].ctypes.data_as(C.POINTER(C.c_float)), C.byref(i), C.byref(d))
            idx[k], d2[k] = i.value, d.value
        return idx, d2

    def icp_run_trim(self, data, inliers, R, t, max_iter=10000, err_diff=1e-7):
        d, dp = _f(data)
        R_ = np.array(R, dtype=np.float32).reshape(9).copy()
        t_ = np.array(t, dtype=np.float32).reshape(3).copy()
        it = C.c_int(0)
        err = lib().orc_icp_run_trim(self.h, self.mp, dp, len(d), int(inliers), R_.ctypes.data_as(C.POINTER(C.c_float)),
                                     t_.ctypes.data_as(C.POINTER(C.c_float)), int(max_iter), float(np.float32(err_diff)), C.byref(it))
        return np.float32(err), R_.reshape(3, 3), t_, it.value

    def icp_run(self, data, R, t, max_iter=10000, err_diff=1e-7):
        d, dp = _f(data)
        R_ = np.array(R, dtype=np.float32).reshape(9).copy()
        t_ = np.array(t, dtype=np.float32).reshape(3).copy()
        it = C.c_int(0)
        err = lib().orc_icp_run(self.h, self.mp, dp, len(d), R_.ctypes.data_as(C.POINTER(C.c_float)),
                                t_.ctypes.data_as(C.POINTER(C.c_float)), int(max_iter), np.float32(err_diff), C.byref(it))
        return np.float32(err), R_.reshape(3, 3), t_, it.value

    def __del__(self):
        try:
            lib().orc_kd_free(self.h)
        except Exception:
            pass


def nn_brute(model, q):
    m, mp = _f(model)
    q, _ = _f(np.asarray(q).reshape(-1, 3))
    idx = np.empty(len(q), np.int32)
    d2 = np.empty(len(q), np.float32)
    L = lib()
    i, d = C.c_int(), C.c_float()
    for k in range(len(q)):
        L.orc_nn_brute(mp, len(m), q[k].ctypes.data_as(C.POINTER(C.c_float)), C.byref(i), C.byref(d))
        idx[k], d2[k] = i.value, d.value
    return idx, d2


def kabsch_rotation(H):
    H_, Hp = _f(np.asarray(H).reshape(9))
    R = np.empty(9, np.float32)
    lib().orc_kabsch_rotation(Hp, R.ctypes.data_as(C.POINTER(C.c_float)))
    return R.reshape(3, 3)


def dt_sse_trim(dt, data, R, t, inliers):
    d, dp = _f(data)
    R_, Rp = _f(np.asarray(R).reshape(9))
    t_, tp = _f(np.asarray(t).reshape(3))
    return np.float32(lib().orc_dt_sse_trim(C.byref(dt.dt), dp, len(d), Rp, tp, int(inliers)))


def register(dt, model, data, mse_thresh, trim_fraction=0.0):
    m, mp = _f(model)
    d, dp = _f(data)
    res = Result()
    lib().orc_register_trim(C.byref(dt.dt), mp, len(m), dp, len(d), float(np.float32(mse_thresh)), float(np.float32(trim_fraction)), C.byref(res))
    return {
        "R": np.array(res.R, dtype=np.float32).reshape(3, 3), "t": np.array(res.t, dtype=np.float32),
        "sse": np.float32(res.sse), "rot_pops": res.rot_pops, "trans_pops": res.trans_pops, "cubes": res.cubes,
        "inner_calls": res.inner_calls, "icp_runs": res.icp_runs, "icp_iters": res.icp_iters,
    }
