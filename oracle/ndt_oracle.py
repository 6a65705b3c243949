"""ctypes binding of the CPU oracle (oracle/ndt_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (ndt_slam_amd/) never imports this module.
PARITY UNPINNED: see oracle/ndt_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Params(C.Structure):
    _fields_ = [
        ("resolution", C.c_float), ("step_size", C.c_double), ("trans_eps", C.c_double),
        ("max_iter", C.c_int), ("outlier_ratio", C.c_double), ("min_pts", C.c_int),
        ("eig_mult", C.c_double), ("cov_unbiased", C.c_int), ("cov_init_identity", C.c_int),
        ("conv_ge", C.c_int), ("radius_inclusive", C.c_int), ("transform_sse", C.c_int),
        ("stale_h_ang", C.c_int), ("snap_thresh", C.c_double), ("mt_max_iter", C.c_int),
        ("mt_mu", C.c_double), ("mt_nu", C.c_double), ("libm_f32", C.c_int), ("grid_margin", C.c_int),
    ]


class Result(C.Structure):
    _fields_ = [
        ("pose", C.c_double * 3), ("T00", C.c_float), ("T10", C.c_float), ("T03", C.c_float),
        ("T13", C.c_float), ("fitness", C.c_double), ("trans_prob", C.c_double),
        ("score", C.c_double), ("H", C.c_double * 9), ("p", C.c_double * 3), ("iters", C.c_int),
        ("evals", C.c_int), ("ref_evals", C.c_int), ("converged", C.c_int), ("status", C.c_int),
        ("flags", C.c_int), ("kbar", C.c_double),
    ]


class MapInfo(C.Structure):
    _fields_ = [("min_bx", C.c_int), ("min_by", C.c_int), ("div_x", C.c_int), ("div_y", C.c_int),
                ("n_cells", C.c_int), ("n_valid", C.c_int), ("n_points", C.c_size_t)]


RESULT_DTYPE = np.dtype([
    ("pose", "f8", 3), ("T00", "f4"), ("T10", "f4"), ("T03", "f4"), ("T13", "f4"),
    ("fitness", "f8"), ("trans_prob", "f8"), ("score", "f8"), ("H", "f8", 9), ("p", "f8", 3),
    ("iters", "i4"), ("evals", "i4"), ("ref_evals", "i4"), ("converged", "i4"), ("status", "i4"),
    ("flags", "i4"), ("kbar", "f8")], align=True)
assert RESULT_DTYPE.itemsize == C.sizeof(Result)


class RunStats(C.Structure):
    _fields_ = [("evals_run", C.c_int), ("pad_", C.c_int), ("pairs_run", C.c_double), ("kbar_run", C.c_double)]


RUN_DTYPE = np.dtype([("evals_run", "i4"), ("pad_", "i4"), ("pairs_run", "f8"), ("kbar_run", "f8")], align=True)
assert RUN_DTYPE.itemsize == C.sizeof(RunStats)
# a result record followed by the run statistics of the match (align_batch(..., run_stats=True))
RESULT_RUN_DTYPE = np.dtype(RESULT_DTYPE.descr + [("evals_run", "i4"), ("pairs_run", "f8"), ("kbar_run", "f8")])


def _with_run(res, st):
    out = np.zeros(len(res), dtype=RESULT_RUN_DTYPE)
    for k in RESULT_DTYPE.names:
        out[k] = res[k]
    for k in ("evals_run", "pairs_run", "kbar_run"):
        out[k] = st[k]
    return out


def build(force=False):
    so = os.path.join(_HERE, "libndt_oracle.so")
    deps = [os.path.join(_HERE, f) for f in ("ndt_oracle.c", "ndt_oracle_octree.c", "ndt_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libndt_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "libndt_oracle.so")
    if not os.path.exists(so):
        so = build()
    L = C.CDLL(so)
    vp, sz, dp, fp = C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_float)
    L.ndt_oracle_default_params.argtypes = [C.POINTER(Params)]
    L.ndt_oracle_params_preset.argtypes = [C.POINTER(Params), C.c_int]
    L.ndt_oracle_map_build.restype = vp
    L.ndt_oracle_map_build.argtypes = [vp, sz, sz, C.POINTER(Params)]
    L.ndt_oracle_map_destroy.argtypes = [vp]
    L.ndt_oracle_map_info_get.argtypes = [vp, C.POINTER(MapInfo)]
    L.ndt_oracle_map_export.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ndt_oracle_eval_at.restype = C.c_double
    L.ndt_oracle_eval_at.argtypes = [vp, vp, sz, sz, dp, dp, dp, dp]
    L.ndt_oracle_align.restype = C.c_int
    L.ndt_oracle_align.argtypes = [vp, vp, sz, sz, dp, C.POINTER(Result), vp, C.c_int]
    L.ndt_oracle_align_batch.restype = C.c_int
    L.ndt_oracle_align_batch.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int]
    L.ndt_oracle_align_seeds.restype = C.c_int
    L.ndt_oracle_align_seeds.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp, C.c_int]
    L.ndt_oracle_align_ex.restype = C.c_int
    L.ndt_oracle_align_ex.argtypes = [vp, vp, sz, sz, dp, C.POINTER(Result), vp, C.c_int, C.POINTER(RunStats)]
    L.ndt_oracle_align_batch_ex.restype = C.c_int
    L.ndt_oracle_align_batch_ex.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp]
    L.ndt_oracle_align_seeds_ex.restype = C.c_int
    L.ndt_oracle_align_seeds_ex.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp, C.c_int, vp]
    L.ndt_oracle_set_memoise.argtypes = [C.c_int]
    L.ndt_oracle_fitness.restype = C.c_double
    L.ndt_oracle_fitness.argtypes = [vp, vp, sz, sz, C.c_float, C.c_float, C.c_float, C.c_float]
    L.ndt_oracle_approx_voxel_filter.restype = sz
    L.ndt_oracle_approx_voxel_filter.argtypes = [vp, sz, sz, C.c_float, vp]
    L.ndt_oracle_yaw_from_T.restype = C.c_double
    L.ndt_oracle_yaw_from_T.argtypes = [C.c_float, C.c_float]
    L.ndt_oracle_mt_trial.restype = C.c_double
    L.ndt_oracle_mt_trial.argtypes = [C.c_double] * 9
    L.ndt_oracle_mt_update.restype = C.c_int
    L.ndt_oracle_mt_update.argtypes = [dp] * 6 + [C.c_double] * 3
    L.ndt_oracle_gauss.argtypes = [C.POINTER(Params), dp, dp]
    L.ndt_oracle_solve3.argtypes = [dp, dp, dp]
    _LIB = L
    return L


PRESETS = {"default": 0, "pcl110": 0, "pcl18": 1, "pcl_new": 2}


def default_params(preset="default", **kw):
    p = Params()
    lib().ndt_oracle_params_preset(C.byref(p), PRESETS[preset])
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _f32c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 2
    return a


class Map:
    """VoxelGridCovariance restatement."""

    def __init__(self, xy, params):
        self.xy = _f32c(xy)
        self.params = params
        self.h = lib().ndt_oracle_map_build(self.xy.ctypes.data, len(self.xy), 8, C.byref(params))
        if not self.h:
            raise RuntimeError("oracle map build failed")

    def __del__(self):
        if getattr(self, "h", None):
            lib().ndt_oracle_map_destroy(self.h)
            self.h = None

    def info(self):
        i = MapInfo()
        lib().ndt_oracle_map_info_get(self.h, C.byref(i))
        return i

    def export(self):
        n = self.info().n_cells
        idx = np.zeros(n, np.int32); cent = np.zeros((n, 2), np.float32)
        mean = np.zeros((n, 2), np.float64); icov = np.zeros((n, 3), np.float64)
        npts = np.zeros(n, np.int32)
        lib().ndt_oracle_map_export(self.h, idx.ctypes.data, cent.ctypes.data, mean.ctypes.data,
                                    icov.ctypes.data, npts.ctypes.data)
        return dict(idx=idx, cent=cent, mean=mean, icov=icov, npts=npts)

    def eval_at(self, scan, p):
        scan = _f32c(scan)
        p = (C.c_double * 3)(*p); g = (C.c_double * 3)(); H = (C.c_double * 9)(); pr = C.c_double()
        s = lib().ndt_oracle_eval_at(self.h, scan.ctypes.data, len(scan), 8, p, g, H, C.byref(pr))
        return s, np.array(g), np.array(H).reshape(3, 3), pr.value

    def align(self, scan, init, trace_cap=0, memoise=False, run_stats=False):
        scan = _f32c(scan)
        r = Result()
        st = RunStats()
        tr = np.zeros((max(trace_cap, 1), 8)) if trace_cap else None
        lib().ndt_oracle_set_memoise(1 if memoise else 0)
        try:
            lib().ndt_oracle_align_ex(self.h, scan.ctypes.data, len(scan), 8, (C.c_double * 3)(*init),
                                      C.byref(r), tr.ctypes.data if trace_cap else None, trace_cap, C.byref(st))
        finally:
            lib().ndt_oracle_set_memoise(0)
        out = np.frombuffer(bytes(r), dtype=RESULT_DTYPE)[0].copy()
        if run_stats:
            out = _with_run(np.array([out]), np.frombuffer(bytes(st), dtype=RUN_DTYPE))[0]
        if trace_cap:
            return out, tr[:min(trace_cap, int(out["flags"]))]
        return out

    def align_batch(self, scans, offsets, inits, nthreads=1, shared_scan=False, memoise=False, run_stats=False):
        """memoise: a line-search trial at the step length of the pass just run re-uses that pass's totals (what the device
        path does; same results).  run_stats: records of RESULT_RUN_DTYPE -- the result plus evals_run / pairs_run / kbar_run,
        the passes and (point, voxel) pairs the DEVICE path runs of the match (kbar_run = the device's ndt_result.kbar)."""
        scans = _f32c(scans)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        inits = np.ascontiguousarray(inits, dtype=np.float64).reshape(-1, 3)
        B = len(inits) if shared_scan else len(offsets) - 1
        res = np.zeros(B, dtype=RESULT_DTYPE)
        st = np.zeros(B, dtype=RUN_DTYPE)
        lib().ndt_oracle_set_memoise(1 if memoise else 0)
        try:
            if shared_scan:                       # every initial guess against scan 0
                one = scans[int(offsets[0]):int(offsets[1])]
                lib().ndt_oracle_align_seeds_ex(self.h, one.ctypes.data, len(one), B, inits.ctypes.data, res.ctypes.data,
                                                nthreads, st.ctypes.data)
            else:
                lib().ndt_oracle_align_batch_ex(self.h, scans.ctypes.data, offsets.ctypes.data, B,
                                                inits.ctypes.data, res.ctypes.data, nthreads, st.ctypes.data)
        finally:
            lib().ndt_oracle_set_memoise(0)
        return _with_run(res, st) if run_stats else res

    def fitness(self, scan, c, s, tx, ty):
        scan = _f32c(scan)
        return lib().ndt_oracle_fitness(self.h, scan.ctypes.data, len(scan), 8, c, s, tx, ty)


def approx_voxel_filter(xy, leaf):
    xy = _f32c(xy)
    out = np.zeros_like(xy)
    n = lib().ndt_oracle_approx_voxel_filter(xy.ctypes.data, len(xy), 8, leaf, out.ctypes.data)
    return out[:n].copy()


def yaw_from_T(T00, T10):
    return lib().ndt_oracle_yaw_from_T(T00, T10)


def gauss(params):
    d1, d2 = C.c_double(), C.c_double()
    lib().ndt_oracle_gauss(C.byref(params), C.byref(d1), C.byref(d2))
    return d1.value, d2.value


def solve3(H, b):
    H = (C.c_double * 9)(*np.asarray(H, float).ravel()); b = (C.c_double * 3)(*b); x = (C.c_double * 3)()
    lib().ndt_oracle_solve3(H, b, x)
    return np.array(x)


def mt_trial(*a):
    return lib().ndt_oracle_mt_trial(*[float(v) for v in a])


def mt_update(a_l, f_l, g_l, a_u, f_u, g_u, a_t, f_t, g_t):
    v = [C.c_double(x) for x in (a_l, f_l, g_l, a_u, f_u, g_u)]
    r = lib().ndt_oracle_mt_update(*[C.byref(x) for x in v], a_t, f_t, g_t)
    return r, [x.value for x in v]


# ---- SURVEY.md 8f row f2 (oracle/ndt_oracle.h) ----
class FuseParams(C.Structure):
    _fields_ = [("coe_ndt_cov", C.c_double), ("coe_vel", C.c_double), ("coe_omega", C.c_double),
                ("del_time", C.c_double), ("score_thre", C.c_double)]


def default_fuse_params(**kw):
    p = FuseParams()
    lib().ndt_oracle_fuse_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def predict(odo_cur, odo_prev, last_pose):
    """-> (motion[3], pred[3]); poses (tx, ty, th_deg)."""
    a = [(C.c_double * 3)(*np.asarray(v, float)) for v in (odo_cur, odo_prev, last_pose)]
    motion, pred = (C.c_double * 3)(), (C.c_double * 3)()
    lib().ndt_oracle_predict(a[0], a[1], a[2], motion, pred)
    return np.array(motion), np.array(pred)


def fuse(result, pred, motion, last_pose, last_cov, prm):
    """result: one record of the result dtype (numpy) -> (successful, fused[3], cov[3,3])."""
    r = Result()
    C.memmove(C.addressof(r), np.ascontiguousarray(result).tobytes(), C.sizeof(Result))
    a = [(C.c_double * 3)(*np.asarray(v, float)) for v in (pred, motion, last_pose)]
    lc = (C.c_double * 9)(*np.asarray(last_cov, float).ravel())
    fused, cov = (C.c_double * 3)(), (C.c_double * 9)()
    ok = lib().ndt_oracle_fuse(C.byref(r), a[0], a[1], a[2], lc, C.byref(prm), fused, cov)
    return int(ok), np.array(fused), np.array(cov).reshape(3, 3)


def remove_neighbors(base, point_list, thre_neighbor):
    """PCFilter::remove_neighborPoint."""
    base = _f32c(base)
    lst = np.ascontiguousarray(point_list, dtype=np.float32).reshape(-1, 2)
    out = np.zeros_like(base)
    L = lib()
    L.ndt_oracle_remove_neighbors.restype = C.c_size_t
    L.ndt_oracle_remove_neighbors.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p]
    n = L.ndt_oracle_remove_neighbors(base.ctypes.data, len(base), lst.ctypes.data, len(lst), thre_neighbor, out.ctypes.data)
    return out[:n].copy()


def difference_indices(base, test, resol):
    """PCFilter::difference_extraction on the literal octree: indices into `test`, in the detector's order."""
    base = np.ascontiguousarray(base, dtype=np.float32).reshape(-1, 2)
    test = np.ascontiguousarray(test, dtype=np.float32).reshape(-1, 2)
    idx = np.zeros(len(test) + 1, dtype=np.int32)
    L = lib()
    L.ndt_oracle_difference_indices.restype = C.c_size_t
    L.ndt_oracle_difference_indices.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p]
    n = L.ndt_oracle_difference_indices(base.ctypes.data, len(base), test.ctypes.data, len(test), resol, idx.ctypes.data)
    if n == C.c_size_t(-1).value:
        raise ValueError("clouds span more than 2^30 voxels")
    return idx[:n].copy()


def difference_extraction(base, test, resol):
    test = np.ascontiguousarray(test, dtype=np.float32).reshape(-1, 2)
    return test[difference_indices(base, test, resol)]


def make_map(scans, first_submap, newest, remove_moving, resol, thre_neighbor):
    """Submap::makeMap over a list of (n_i, 2) float32 scans -> (n, 2) float32 local-map points."""
    scans = [np.ascontiguousarray(s, dtype=np.float32).reshape(-1, 2) for s in scans]
    off = np.zeros(len(scans) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in scans])
    allp = np.concatenate(scans) if scans else np.zeros((0, 2), np.float32)
    out = np.zeros(((2 if len(scans) == 1 else 1) * len(allp) + 1, 2), dtype=np.float32)   # one scan: appended twice
    L = lib()
    L.ndt_oracle_make_map.restype = C.c_size_t
    L.ndt_oracle_make_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                      C.c_void_p]
    n = L.ndt_oracle_make_map(allp.ctypes.data, off.ctypes.data, len(scans), int(first_submap), int(newest),
                              int(remove_moving), resol, thre_neighbor, out.ctypes.data)
    if n == C.c_size_t(-1).value:
        raise ValueError("clouds span more than 2^30 voxels")
    return out[:n].copy()
