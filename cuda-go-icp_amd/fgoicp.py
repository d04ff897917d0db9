"""Python mirrors of the reference's operator interface for this path, over the C ABI.

Reference classes mirrored (names, argument meaning, error behaviour):
  Config                      src/common.h:133-180, src/common.cpp:12-77
  load_cloud                  src/common.cpp:205-228
  icp::RotNode / TransNode    src/fgoicp/fgoicp_common.hpp:64-129 (here with the CPU path's
                              corner+width parametrisation, src/goicp/jly_goicp.h:44-72)
  icp::Registration           src/fgoicp/registration.hpp:44-98
  icp::IterativeClosestPoint3D  src/fgoicp/icp3d.hpp:9-41
  icp::FastGoICP              src/fgoicp/fgoicp.hpp:11-69
All compute happens in libgoicp_mi355.so on the GPU; these classes only marshal arguments.
"""
import ctypes as C
import threading

import numpy as np

from . import binding as B


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Config:
    """Config(toml_filepath): same keys, defaults and clamps as the reference; raises on parse error."""

    class _NS:
        pass

    def __init__(self, toml_filepath):
        c = B.CConfig()
        B.check(B.load_library().goicp_config_load(str(toml_filepath).encode(), C.byref(c)))
        self.mode, self.trim = c.mode, bool(c.trim)
        self.subsample, self.mse_threshold, self.resize = c.subsample, c.mse_threshold, c.resize
        self.description = c.description.decode()
        self.io = Config._NS()
        self.io.target, self.io.source = c.target.decode(), c.source.decode()
        self.io.output, self.io.visualization = c.output.decode(), c.visualization.decode()
        self.viz = Config._NS()
        self.viz.theta, self.viz.phi, self.viz.spin_after_finish = c.viz_theta, c.viz_phi, bool(c.viz_spin_after_finish)
        self.rotation = Config._NS()
        self.translation = Config._NS()
        for k, ax in enumerate("xyz"):
            setattr(self.rotation, ax + "min", c.rot_min[k]); setattr(self.rotation, ax + "max", c.rot_max[k])
            setattr(self.translation, ax + "min", c.trans_min[k]); setattr(self.translation, ax + "max", c.trans_max[k])
        self.rotation.search_depth, self.translation.search_depth = c.rot_search_depth, c.trans_search_depth
        self.rotation.present, self.translation.present = bool(c.has_rotation_range), bool(c.has_translation_range)
        self._c = c

    def engine_params(self):
        """Keyword arguments for Registration / FastGoICP carrying what the TOML holds for the engine: the search
        ranges and depths when the [params.rotation] / [params.translation] tables are present."""
        p = B.CParams()
        B.load_library().goicp_params_from_config(C.byref(self._c), C.byref(p))
        out = {}
        if p.use_rot_range:
            out.update(use_rot_range=1, rot_min=list(p.rot_min), rot_max=list(p.rot_max), rot_search_depth=p.rot_search_depth)
        if p.use_trans_range:
            out.update(use_trans_range=1, trans_min=list(p.trans_min), trans_max=list(p.trans_max), trans_search_depth=p.trans_search_depth)
        return out


def load_cloud(filepath, subsample=1.0, resize=1.0, seed=0):
    """load_cloud(path, subsample, resize) -> (n,3) float32.  .ply / .txt; raises GoicpError(GOICP_ERR_IO)."""
    lib = B.load_library()
    p = C.POINTER(C.c_float)()
    n = C.c_size_t(0)
    B.check(lib.goicp_cloud_load(str(filepath).encode(), float(subsample), float(resize), int(seed), C.byref(p), C.byref(n)))
    try:
        out = np.ctypeslib.as_array(p, shape=(n.value * 3,)).copy().reshape(-1, 3) if n.value else np.zeros((0, 3), np.float32)
    finally:
        lib.goicp_cloud_free(p)
    return out


def rodrigues(v):
    v = _f32(v, (3,))
    R = np.empty(9, np.float32)
    B.load_library().goicp_rodrigues(_fptr(v), _fptr(R))
    return R.reshape(3, 3)


class RotNode:
    """Rotation cube in angle-axis space: corner (a,b,c), width w, level l (jly_goicp.h:44-57)."""

    def __init__(self, a, b, c, w, lb=0.0, ub=0.0, l=0):
        self.a, self.b, self.c, self.w, self.lb, self.ub, self.l = map(float, (a, b, c, w, lb, ub, l))
        self.l = int(l)

    @property
    def centre(self):
        h = np.float32(self.w) / np.float32(2)
        return np.array([np.float32(self.a) + h, np.float32(self.b) + h, np.float32(self.c) + h], np.float32)

    @property
    def R(self):
        return rodrigues(self.centre)

    def __lt__(self, o):        # priority order of the reference queues
        return self.lb > o.lb if self.lb != o.lb else self.w < o.w


class TransNode:
    """Translation cube: corner (x,y,z), width w (jly_goicp.h:59-72)."""

    def __init__(self, x, y, z, w, lb=0.0, ub=0.0):
        self.x, self.y, self.z, self.w, self.lb, self.ub = map(float, (x, y, z, w, lb, ub))

    @property
    def centre(self):
        h = np.float32(self.w) / np.float32(2)
        return np.array([np.float32(self.x) + h, np.float32(self.y) + h, np.float32(self.z) + h], np.float32)

    def __lt__(self, o):
        return self.lb > o.lb if self.lb != o.lb else self.w < o.w


class Registration:
    """icp::Registration(pct, nt, pcs, ns): owns the device-resident clouds, DT and k-d tree."""

    def __init__(self, pct, pcs, mse_threshold=1e-3, **params):
        self._lib = B.load_library()
        self.pct, self.pcs = _f32(pct, (-1, 3)), _f32(pcs, (-1, 3))
        p = B.CParams()
        self._lib.goicp_params_default(C.byref(p))
        p.mse_threshold = float(mse_threshold)
        for k, v in params.items():
            if not hasattr(p, k):
                raise TypeError("unknown engine parameter %r" % k)
            if isinstance(getattr(p, k), C.Array):
                v = type(getattr(p, k))(*[float(x) for x in v])
            setattr(p, k, v)
        self.params = p
        h = C.c_void_p()
        B.check(self._lib.goicp_create(C.byref(p), _fptr(self.pct), len(self.pct), _fptr(self.pcs), len(self.pcs), C.byref(h)))
        self.handle = h
        self.ns, self.nt = len(self.pcs), len(self.pct)
        thr, inl = C.c_float(), C.c_int32()
        B.check(self._lib.goicp_thresholds(h, C.byref(thr), C.byref(inl)))
        self.sse_threshold, self.inliers = np.float32(thr.value), inl.value     # as the engine uses them (trimming included)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.goicp_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- DT ----
    def dt_info(self):
        V, s, o = C.c_int32(), C.c_double(), (C.c_double * 3)()
        B.check(self._lib.goicp_dt_info(self.handle, C.byref(V), C.byref(s), o))
        return V.value, s.value, tuple(o)

    def dt_download(self):
        V = self.dt_info()[0]
        g = np.empty(V ** 3, np.float32)
        B.check(self._lib.goicp_dt_download(self.handle, _fptr(g)))
        return g.reshape(V, V, V)

    def rot_coeff(self, level):
        return np.float32(self._lib.goicp_rot_coeff(self.handle, int(level)))

    # ---- the two compute_sse_error overloads ----
    def compute_sse_error(self, *args, **kw):
        """compute_sse_error(R, t) -> sse   |   compute_sse_error(rnode, tnodes, fix_rot) -> (lb[], ub[])"""
        if len(args) >= 2 and isinstance(args[0], RotNode):
            rnode, tnodes = args[0], args[1]
            fix_rot = args[2] if len(args) > 2 else kw.get("fix_rot", True)
            cubes = np.array([[*t.centre, np.float32(t.w)] for t in tnodes], np.float32).reshape(-1, 4)
            ub, lb = self.eval_bounds(rnode.R, cubes, -1 if fix_rot else rnode.l)
            return lb, ub
        R, t = _f32(args[0], (9,)), _f32(args[1], (3,))
        sse = C.c_float()
        B.check(self._lib.goicp_eval_sse(self.handle, _fptr(R), _fptr(t), C.byref(sse)))
        return np.float32(sse.value)

    def eval_bounds(self, R, cubes, level=-1):
        R, cubes = _f32(R, (9,)), _f32(cubes, (-1, 4))
        n = len(cubes)
        ub, lb = np.empty(n, np.float32), np.empty(n, np.float32)
        B.check(self._lib.goicp_eval_bounds(self.handle, _fptr(R), _fptr(cubes), n, int(level), _fptr(ub), _fptr(lb)))
        return ub, lb

    def eval_bounds_batch(self, rots, cubes):
        """rots (K,3,3); cubes structured array of binding.CCube (or (B,6) with rot in the last col)."""
        rots = _f32(rots, (-1, 9))
        arr = (B.CCube * len(cubes))(*[B.CCube(*map(float, c[:5]), int(c[5])) for c in cubes])
        n = len(cubes)
        ub, lb = np.empty(n, np.float32), np.empty(n, np.float32)
        B.check(self._lib.goicp_eval_bounds_batch(self.handle, _fptr(rots), len(rots), arr, n, _fptr(ub), _fptr(lb)))
        return ub, lb

    def inner_bnb(self, R, level=-1, incumbent=1e10):
        """branch_and_bound_R3(rnode, fix_rot): -> (value, best_node[4] or None, counters)"""
        R = _f32(R, (9,))
        val, node, cnt = C.c_float(), np.full(4, np.nan, np.float32), B.CCounters()
        B.check(self._lib.goicp_inner_bnb(self.handle, _fptr(R), int(level), float(incumbent), C.byref(val), _fptr(node), C.byref(cnt)))
        return np.float32(val.value), (None if np.isnan(node[3]) else node), cnt

    def nn_query(self, q):
        q = _f32(q, (-1, 3))
        idx, d2 = np.empty(len(q), np.int32), np.empty(len(q), np.float32)
        B.check(self._lib.goicp_nn_query(self.handle, _fptr(q), len(q), idx.ctypes.data_as(C.POINTER(C.c_int32)), _fptr(d2)))
        return idx, d2

    def transform_source(self, R, t):
        R, t = _f32(R, (9,)), _f32(t, (3,))
        out = np.empty((self.ns, 3), np.float32)
        B.check(self._lib.goicp_transform_source(self.handle, _fptr(R), _fptr(t), _fptr(out)))
        return out

    def icp_step(self):
        B.check(self._lib.goicp_icp_step(self.handle))
        return self.poll()

    def poll(self):
        r = B.CResult()
        B.check(self._lib.goicp_poll(self.handle, C.byref(r)))
        return r


class IterativeClosestPoint3D:
    """IterativeClosestPoint3D(reg, pct, pcs, max_iter, threshold, R, t).run() -> (sse, R, t).
    `threshold` is the reference CPU path's err_diff (mean squared error decrease per point)."""

    def __init__(self, reg, max_iter=10000, convergence_threshold=1e-7, R=None, t=None):
        self.reg, self.max_iter, self.thr = reg, int(max_iter), float(convergence_threshold)
        self.R = _f32(np.eye(3) if R is None else R, (9,)).copy()
        self.t = _f32(np.zeros(3) if t is None else t, (3,)).copy()
        self.iters = 0

    def run(self):
        err, it = C.c_float(), C.c_int32()
        B.check(self.reg._lib.goicp_icp_run(self.reg.handle, _fptr(self.R), _fptr(self.t), self.max_iter, self.thr,
                                            C.byref(err), C.byref(it)))
        self.iters = it.value
        return np.float32(err.value), self.R.reshape(3, 3).copy(), self.t.copy()


class FastGoICP:
    """icp::FastGoICP(pct, pcs, mse_threshold, mtx): run() blocks (use a worker thread), the result
    fields are a consistent snapshot (the reference published them unlocked)."""

    def __init__(self, pct, pcs, mse_threshold, mtx=None, **params):
        self.registration = Registration(pct, pcs, mse_threshold, **params)
        self.mtx = mtx or threading.Lock()
        self.mse_threshold = float(mse_threshold)
        self.sse_threshold = self.registration.sse_threshold      # mse_threshold * inlierNum (jly_goicp.cpp:198-208), from the engine

    def run(self):
        B.check(self.registration._lib.goicp_register(self.registration.handle))

    def cancel(self):
        B.check(self.registration._lib.goicp_cancel(self.registration.handle))

    def _snap(self):
        return self.registration.poll()

    def get_best_error(self):
        return np.float32(self._snap().best_sse)

    optR = property(lambda s: np.array(s._snap().optR, np.float32).reshape(3, 3))
    optT = property(lambda s: np.array(s._snap().optT, np.float32))
    curR = property(lambda s: np.array(s._snap().curR, np.float32).reshape(3, 3))
    curT = property(lambda s: np.array(s._snap().curT, np.float32))
    finished = property(lambda s: bool(s._snap().finished))
    counters = property(lambda s: s._snap().counters)

    def write_output(self, path):
        B.check(self.registration._lib.goicp_result_write_toml(self.registration.handle, str(path).encode()))

    def write_visualization(self, path):
        B.check(self.registration._lib.goicp_result_write_ply(self.registration.handle, str(path).encode()))

    # stepped API used by the sharded driver
    def set_shard(self, rank, world):
        B.check(self.registration._lib.goicp_set_shard(self.registration.handle, rank, world))

    def register_begin(self):
        B.check(self.registration._lib.goicp_register_begin(self.registration.handle))

    def register_step(self, max_rot_pops=8):
        s = B.CStepStatus()
        B.check(self.registration._lib.goicp_register_step(self.registration.handle, int(max_rot_pops), C.byref(s)))
        return {"finished": bool(s.finished), "early_exit": bool(s.early_exit), "best_sse": float(s.best_sse),
                "frontier_lb": float(s.frontier_lb), "rot_pops": int(s.rot_pops)}

    def offer_best(self, sse, R, t):
        R, t = _f32(R, (9,)), _f32(t, (3,))
        B.check(self.registration._lib.goicp_offer_best(self.registration.handle, float(sse), _fptr(R), _fptr(t)))

    def register_end(self):
        B.check(self.registration._lib.goicp_register_end(self.registration.handle))

    def pose(self):
        r = self._snap()
        return float(r.best_sse), np.array(r.optR, np.float32), np.array(r.optT, np.float32)
