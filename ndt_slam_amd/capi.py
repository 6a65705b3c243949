"""ctypes binding of libndt_mi355x.so (include/ndt_mi355x.h).

This is the only way Python reaches the kernels: every call goes through the C ABI a C++
maintainer of the reference would bind.  There is no CPU fallback: a missing
library or a missing GPU raises.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NDT_LIB_PATH") or os.path.join(_HERE, "libndt_mi355x.so")   # override: diagnostic builds
_LIB = None


class NdtError(RuntimeError):
    pass


class Params(C.Structure):
    """ndt_params (include/ndt_mi355x.h)."""
    _fields_ = [
        ("resolution", C.c_float), ("step_size", C.c_double), ("trans_eps", C.c_double),
        ("max_iter", C.c_int), ("outlier_ratio", C.c_double), ("min_pts", C.c_int),
        ("eig_mult", C.c_double), ("cov_unbiased", C.c_int), ("cov_init_identity", C.c_int),
        ("conv_ge", C.c_int), ("radius_inclusive", C.c_int), ("transform_sse", C.c_int),
        ("stale_h_ang", C.c_int), ("snap_thresh", C.c_double), ("mt_max_iter", C.c_int),
        ("mt_mu", C.c_double), ("mt_nu", C.c_double), ("libm_f32", C.c_int), ("grid_margin", C.c_int),
    ]


class FuseParams(C.Structure):
    """ndt_fuse_params (include/ndt_mi355x.h)."""
    _fields_ = [("coe_ndt_cov", C.c_double), ("coe_vel", C.c_double), ("coe_omega", C.c_double),
                ("del_time", C.c_double), ("score_thre", C.c_double)]


class MapInfo(C.Structure):
    _fields_ = [("min_bx", C.c_int), ("min_by", C.c_int), ("div_x", C.c_int), ("div_y", C.c_int),
                ("n_cells", C.c_int), ("n_valid", C.c_int), ("n_points", C.c_size_t)]


# ndt_result as a numpy record (same layout as the C struct)
RESULT_DTYPE = np.dtype([
    ("pose", "f8", 3), ("T00", "f4"), ("T10", "f4"), ("T03", "f4"), ("T13", "f4"),
    ("fitness", "f8"), ("trans_prob", "f8"), ("score", "f8"), ("H", "f8", 9), ("p", "f8", 3),
    ("iters", "i4"), ("evals", "i4"), ("ref_evals", "i4"), ("converged", "i4"), ("status", "i4"),
    ("flags", "i4"), ("kbar", "f8")], align=True)
RESULT_BYTES = RESULT_DTYPE.itemsize
FLAG_WINDOW_SPILL, FLAG_REGION_CLIPPED, FLAG_UNSORTED = 1, 2, 4      # ndt_result.flags
NDT_OK, NDT_E_ARG, NDT_E_HIP, NDT_E_NO_DEVICE, NDT_E_GRID, NDT_E_NOMEM = 0, -1, -2, -3, -4, -5    # ndt_status
OPT_MAX_HELPERS, OPT_WORKGROUPS, OPT_INJECT_FAULT, OPT_DEFER_FITNESS = 1, 2, 3, 4            # ndt_ctx_set_option

EXPORTS = [
    "ndt_default_params", "ndt_params_pcl110", "ndt_params_pcl18", "ndt_params_pcl_new", "ndt_ctx_create", "ndt_ctx_destroy", "ndt_last_error", "ndt_ctx_stream",
    "ndt_ctx_set_stream", "ndt_ctx_set_option", "ndt_ctx_wait_launch",
    "ndt_map_build", "ndt_map_build_dev", "ndt_map_rebuild_begin", "ndt_map_rebuild_end", "ndt_map_destroy", "ndt_map_info_get", "ndt_map_export",
    "ndt_align", "ndt_align_batch", "ndt_align_batch_dev", "ndt_align_batch_prepare_dev", "ndt_prepare_timing", "ndt_align_batch_trace", "ndt_eval_at",
    "ndt_fitness_at", "ndt_last_timing", "ndt_kernel_timing", "ndt_launch_interval", "ndt_align_batch_sharded", "ndt_prefilter", "ndt_prefilter_batch_dev",
    "ndt_fuse_default_params", "ndt_predict_batch_dev", "ndt_fuse_batch_dev",
    "ndt_remove_neighbors", "ndt_remove_neighbors_dev",
    "ndt_difference_extraction", "ndt_difference_extraction_dev", "ndt_make_map", "ndt_make_map_dev",
    "ndt_selftest_libm_f32",
]


def lib():
    """Load the shared library; fail loudly when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise NdtError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    for name in ("ndt_default_params", "ndt_params_pcl110", "ndt_params_pcl18", "ndt_params_pcl_new"):
        getattr(L, name).argtypes = [C.POINTER(Params)]
    L.ndt_ctx_create.argtypes = [i, C.POINTER(vp)]
    L.ndt_ctx_destroy.argtypes = [vp]
    L.ndt_last_error.restype = C.c_char_p
    L.ndt_last_error.argtypes = [vp]
    L.ndt_ctx_stream.restype = vp
    L.ndt_ctx_stream.argtypes = [vp]
    L.ndt_ctx_set_stream.argtypes = [vp, vp]
    L.ndt_ctx_set_option.argtypes = [vp, i, C.c_longlong]
    L.ndt_ctx_wait_launch.argtypes = [vp, i, vp]
    L.ndt_map_build.argtypes = [vp, vp, sz, sz, C.POINTER(Params), C.POINTER(vp)]
    L.ndt_map_build_dev.argtypes = [vp, vp, sz, sz, C.POINTER(Params), C.POINTER(vp)]
    L.ndt_map_rebuild_begin.argtypes = [vp, vp, sz, sz, C.POINTER(Params), vp]
    L.ndt_map_rebuild_end.argtypes = [vp, vp]
    L.ndt_map_destroy.argtypes = [vp]
    L.ndt_map_info_get.argtypes = [vp, C.POINTER(MapInfo)]
    L.ndt_map_export.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ndt_align.argtypes = [vp, vp, vp, sz, sz, vp, vp]
    L.ndt_align_batch.argtypes = [vp, vp, vp, vp, i, i, vp, vp]
    L.ndt_align_batch_dev.argtypes = [vp, vp, vp, vp, i, sz, i, vp, vp, vp]
    L.ndt_align_batch_prepare_dev.argtypes = [vp, vp, vp, vp, i, sz, i, vp, vp]
    L.ndt_prepare_timing.argtypes = [vp, C.POINTER(C.c_float)]
    L.ndt_align_batch_trace.argtypes = [vp, vp, vp, vp, i, i, vp, vp, vp, i, vp]
    L.ndt_eval_at.argtypes = [vp, vp, vp, sz, sz, vp, vp, vp, vp, vp]
    L.ndt_fitness_at.argtypes = [vp, vp, vp, sz, sz, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    L.ndt_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ndt_selftest_libm_f32.argtypes = [vp, vp, C.c_size_t, vp, vp, vp]
    L.ndt_kernel_timing.argtypes = [vp, i, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ndt_launch_interval.argtypes = [vp, i, C.POINTER(C.c_float)]
    L.ndt_align_batch_sharded.argtypes = [vp, vp, i, vp, vp, i, i, vp, vp]
    L.ndt_prefilter.argtypes = [vp, vp, sz, sz, C.c_float, vp, C.POINTER(sz)]
    L.ndt_prefilter_batch_dev.argtypes = [vp, vp, sz, vp, i, sz, C.c_float, vp, vp, vp]
    L.ndt_fuse_default_params.argtypes = [C.POINTER(FuseParams)]
    L.ndt_predict_batch_dev.argtypes = [vp, vp, vp, vp, i, vp, vp, vp, vp]
    L.ndt_fuse_batch_dev.argtypes = [vp, vp, vp, vp, vp, vp, i, C.POINTER(FuseParams), vp, vp, vp, vp]
    L.ndt_remove_neighbors.argtypes = [vp, vp, sz, sz, vp, sz, sz, C.c_double, vp, C.POINTER(sz)]
    L.ndt_remove_neighbors_dev.argtypes = [vp, vp, sz, sz, vp, sz, sz, C.c_double, vp, vp, vp]
    L.ndt_difference_extraction.argtypes = [vp, vp, sz, sz, vp, sz, sz, C.c_double, vp, C.POINTER(sz)]
    L.ndt_difference_extraction_dev.argtypes = [vp, vp, sz, sz, vp, sz, sz, C.c_double, vp, vp, vp]
    L.ndt_make_map.argtypes = [vp, vp, sz, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp,
                               C.POINTER(sz)]
    L.ndt_make_map_dev.argtypes = [vp, vp, sz, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp, vp, vp]
    for name in EXPORTS:
        if name not in ("ndt_last_error", "ndt_ctx_stream"):
            getattr(L, name).restype = i
    _LIB = L
    return L


PRESETS = {"default": "ndt_default_params", "pcl110": "ndt_params_pcl110", "pcl18": "ndt_params_pcl18",
           "pcl_new": "ndt_params_pcl_new"}


def default_params(preset="default", **kw):
    """ndt_params of a PCL-version preset (include/ndt_mi355x.h), fields overridden by keyword."""
    p = Params()
    getattr(lib(), PRESETS[preset])(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def align_batch_sharded(maps, scans, offsets, inits, shared_scan=False, partial=False):
    """ndt_align_batch_sharded: `maps` = one Map per device (each with its own Context), the batch on the host.
    `partial`: return (rc, records) instead of raising when a shard failed -- every record of a failed shard carries
    that shard's error in `status`, the others are complete."""
    scans = _f32c(scans)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    inits = np.ascontiguousarray(inits, dtype=np.float64).reshape(-1, 3)
    B = len(inits)
    res = np.zeros(B, dtype=RESULT_DTYPE)
    n = len(maps)
    cx = (C.c_void_p * n)(*[m.ctx.h for m in maps])
    mp = (C.c_void_p * n)(*[m.h for m in maps])
    rc = lib().ndt_align_batch_sharded(cx, mp, n, scans.ctypes.data, offsets.ctypes.data, B, int(shared_scan),
                                       inits.ctypes.data, res.ctypes.data)
    if partial:
        return rc, res
    if rc:
        raise NdtError("ndt_align_batch_sharded -> %d: %s" % (rc, lib().ndt_last_error(None).decode()))
    return res


def default_fuse_params(**kw):
    p = FuseParams()
    lib().ndt_fuse_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _f32c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 2:
        raise ValueError("expected an [n, 2] float32 array")
    return a


class Context:
    """ndt_ctx: one per process and device."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        rc = lib().ndt_ctx_create(device, C.byref(self.h))
        if rc:
            raise NdtError("ndt_ctx_create(%d) -> %d: %s" % (device, rc, lib().ndt_last_error(None).decode()))
        self.device = device

    def check(self, rc, what):
        if rc:
            raise NdtError("%s -> %d: %s" % (what, rc, lib().ndt_last_error(self.h).decode()))

    @property
    def stream(self):
        return lib().ndt_ctx_stream(self.h)

    def set_stream(self, stream):
        """Order all work of this context on a caller-owned hipStream_t (int handle or None)."""
        self.check(lib().ndt_ctx_set_stream(self.h, stream), "ndt_ctx_set_stream")

    def set_option(self, option, value):
        """ndt_ctx_set_option: OPT_MAX_HELPERS (0 = no work sharing), OPT_WORKGROUPS (0 = one per CU)."""
        self.check(lib().ndt_ctx_set_option(self.h, option, value), "ndt_ctx_set_option")

    def wait_launch(self, back, stream=None):
        """ndt_ctx_wait_launch: `stream` (int handle; None = the context's stream) waits for the match launch `back`
        launches ago (0 = the most recent), fitness kernels included -- no event record on the launch's stream."""
        self.check(lib().ndt_ctx_wait_launch(self.h, back, stream), "ndt_ctx_wait_launch")

    def prefilter(self, xy, leaf):
        """pcl::ApproximateVoxelGrid on one scan ([n, 2] float32) -> filtered [m, 2] float32."""
        xy = _f32c(xy)
        out = np.empty_like(xy)
        m = C.c_size_t()
        self.check(lib().ndt_prefilter(self.h, xy.ctypes.data, len(xy), 8, leaf, out.ctypes.data, C.byref(m)),
                   "ndt_prefilter")
        return out[:m.value].copy()

    def prefilter_batch_dev(self, raw_ptr, stride, raw_offsets_ptr, B, total_raw_points, leaf, out_ptr,
                            out_offsets_ptr, stream=None):
        """Device pointers in and out (see include/ndt_mi355x.h); asynchronous."""
        self.check(lib().ndt_prefilter_batch_dev(self.h, raw_ptr, stride, raw_offsets_ptr, B, total_raw_points, leaf,
                                                 out_ptr, out_offsets_ptr, stream), "ndt_prefilter_batch_dev")

    def remove_neighbors(self, base, point_list, thre_neighbor):
        """PCFilter::remove_neighborPoint: base points with no list point within thre_neighbor, in order."""
        base = _f32c(base)
        lst = np.ascontiguousarray(point_list, dtype=np.float32).reshape(-1, 2)
        out = np.empty_like(base)
        m = C.c_size_t()
        self.check(lib().ndt_remove_neighbors(self.h, base.ctypes.data, 8, len(base), lst.ctypes.data if len(lst) else None,
                                              8, len(lst), thre_neighbor, out.ctypes.data, C.byref(m)),
                   "ndt_remove_neighbors")
        return out[:m.value].copy()

    def difference_extraction(self, base, test, resol):
        """PCFilter::difference_extraction: points of `test` in octree voxels `base` does not occupy (input order)."""
        base = np.ascontiguousarray(base, dtype=np.float32).reshape(-1, 2)
        test = np.ascontiguousarray(test, dtype=np.float32).reshape(-1, 2)
        out = np.empty((len(test) + 1, 2), dtype=np.float32)
        m = C.c_size_t()
        self.check(lib().ndt_difference_extraction(self.h, base.ctypes.data if len(base) else None, 8, len(base),
                                                   test.ctypes.data if len(test) else None, 8, len(test), resol,
                                                   out.ctypes.data, C.byref(m)), "ndt_difference_extraction")
        return out[:m.value].copy()

    def make_map(self, scans, first_submap, newest, remove_moving=True, resol=0.05, thre_neighbor=0.1):
        """Submap::makeMap over a list of (n_i, 2) float32 scans in the map frame -> (n, 2) float32."""
        scans = [np.ascontiguousarray(s, dtype=np.float32).reshape(-1, 2) for s in scans]
        off = np.zeros(len(scans) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(s) for s in scans])
        allp = np.ascontiguousarray(np.concatenate(scans)) if scans else np.zeros((0, 2), np.float32)
        out = np.empty(((2 if len(scans) == 1 else 1) * len(allp) + 1, 2), dtype=np.float32)
        m = C.c_size_t()
        self.check(lib().ndt_make_map(self.h, allp.ctypes.data, 8, off.ctypes.data, len(scans), int(first_submap),
                                      int(newest), int(remove_moving), resol, thre_neighbor, out.ctypes.data,
                                      C.byref(m)), "ndt_make_map")
        return out[:m.value].copy()

    def make_map_dev(self, scans_ptr, stride, offsets, first_submap, newest, remove_moving, resol, thre_neighbor,
                     out_ptr, n_out_ptr, stream=None):
        """Device pointers for the points, the result and its uint64 count; `offsets` a host uint64 array."""
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.check(lib().ndt_make_map_dev(self.h, scans_ptr, stride, off.ctypes.data, len(off) - 1, int(first_submap),
                                          int(newest), int(remove_moving), resol, thre_neighbor, out_ptr, n_out_ptr,
                                          stream), "ndt_make_map_dev")

    def predict_batch_dev(self, odo_cur_ptr, odo_prev_ptr, last_pose_ptr, B, motion_ptr, pred_ptr, init_ptr=None,
                          stream=None):
        """Row f2, before the match: device pointers to B x 3 doubles (degrees); asynchronous."""
        self.check(lib().ndt_predict_batch_dev(self.h, odo_cur_ptr, odo_prev_ptr, last_pose_ptr, B, motion_ptr, pred_ptr,
                                               init_ptr, stream), "ndt_predict_batch_dev")

    def fuse_batch_dev(self, results_ptr, pred_ptr, motion_ptr, last_pose_ptr, last_cov_ptr, B, prm, fused_ptr, cov_ptr,
                       successful_ptr=None, stream=None):
        """Row f2, after the match: device pointers; asynchronous."""
        self.check(lib().ndt_fuse_batch_dev(self.h, results_ptr, pred_ptr, motion_ptr, last_pose_ptr, last_cov_ptr, B,
                                            C.byref(prm), fused_ptr, cov_ptr, successful_ptr, stream),
                   "ndt_fuse_batch_dev")

    def selftest_libm_f32(self, yaws):
        """Device cosf / sinf / initial yaw (ndt_libm_f32.hip.h) for an array of float32 yaws -> (cos, sin, init_yaw)."""
        y = np.ascontiguousarray(yaws, dtype=np.float32).ravel()
        c, s, y0 = np.zeros_like(y), np.zeros_like(y), np.zeros_like(y)
        self.check(lib().ndt_selftest_libm_f32(self.h, y.ctypes.data, len(y), c.ctypes.data, s.ctypes.data, y0.ctypes.data),
                   "ndt_selftest_libm_f32")
        return c, s, y0

    def last_timing(self):
        a, b = C.c_float(), C.c_float()
        lib().ndt_last_timing(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def kernel_timing(self, back=0):
        """(match kernel ms, fitness kernels ms) of one of this context's last 64 launches; blocks until it is done."""
        a, b = C.c_float(), C.c_float()
        self.check(lib().ndt_kernel_timing(self.h, back, C.byref(a), C.byref(b)), "ndt_kernel_timing")
        return a.value, b.value

    def prepare_timing(self):
        """ms of the kernel of the prepared batch(es) this context's launches used (0.0: none)."""
        a = C.c_float()
        self.check(lib().ndt_prepare_timing(self.h, C.byref(a)), "ndt_prepare_timing")
        return a.value

    def launch_interval(self, back=0):
        """ms from the start of the match kernel of launch back + 1 to the start of launch back (both among the last 64)."""
        a = C.c_float()
        self.check(lib().ndt_launch_interval(self.h, back, C.byref(a)), "ndt_launch_interval")
        return a.value

    def close(self):
        if self.h:
            lib().ndt_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be gone: leave the handles to the OS
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class Map:
    """ndt_map: the NDT voxel grid of one target cloud (replaces ndt.setInputTarget)."""

    def __init__(self, ctx, xy=None, params=None, dev_ptr=None, n=None, stride=8):
        self.ctx = ctx
        self.params = params if params is not None else default_params()
        self.h = C.c_void_p()
        self.rebuild(xy=xy, dev_ptr=dev_ptr, n=n, stride=stride)

    def rebuild(self, xy=None, dev_ptr=None, n=None, stride=8):
        if dev_ptr is not None:
            rc = lib().ndt_map_build_dev(self.ctx.h, dev_ptr, n, stride, C.byref(self.params), C.byref(self.h))
        else:
            xy = _f32c(xy)
            rc = lib().ndt_map_build(self.ctx.h, xy.ctypes.data, len(xy), 8, C.byref(self.params), C.byref(self.h))
        self.ctx.check(rc, "ndt_map_build")

    def rebuild_begin(self, dev_ptr, n, stride=8):
        """Queue the rebuild with the voxel grid of the previous build and return (no host wait); see rebuild_end."""
        self.ctx.check(lib().ndt_map_rebuild_begin(self.ctx.h, dev_ptr, n, stride, C.byref(self.params), self.h),
                       "ndt_map_rebuild_begin")

    def rebuild_end(self):
        """True if the cloud's bounding box had moved: the build was queued again and launches queued since
        rebuild_begin used a stale grid."""
        rc = lib().ndt_map_rebuild_end(self.ctx.h, self.h)
        if rc < 0:
            self.ctx.check(rc, "ndt_map_rebuild_end")
        return rc == 1

    def info(self):
        i = MapInfo()
        self.ctx.check(lib().ndt_map_info_get(self.h, C.byref(i)), "ndt_map_info_get")
        return i

    def export(self):
        n = self.info().n_cells
        idx = np.zeros(n, np.int32); cent = np.zeros((n, 2), np.float32)
        mean = np.zeros((n, 2), np.float64); icov = np.zeros((n, 3), np.float64)
        npts = np.zeros(n, np.int32)
        self.ctx.check(lib().ndt_map_export(self.h, idx.ctypes.data, cent.ctypes.data, mean.ctypes.data,
                                            icov.ctypes.data, npts.ctypes.data), "ndt_map_export")
        return dict(idx=idx, cent=cent, mean=mean, icov=icov, npts=npts)

    def align(self, scan, init):
        scan = _f32c(scan)
        init = np.ascontiguousarray(init, dtype=np.float64)
        res = np.zeros(1, dtype=RESULT_DTYPE)
        self.ctx.check(lib().ndt_align(self.ctx.h, self.h, scan.ctypes.data, len(scan), 8, init.ctypes.data,
                                       res.ctypes.data), "ndt_align")
        return res[0]

    def align_batch(self, scans, offsets, inits, shared_scan=False, trace_cap=0):
        scans = _f32c(scans)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        inits = np.ascontiguousarray(inits, dtype=np.float64).reshape(-1, 3)
        B = len(inits)
        res = np.zeros(B, dtype=RESULT_DTYPE)
        if trace_cap:
            trace = np.zeros((B, trace_cap, 8)); rows = np.zeros(B, np.int32)
            self.ctx.check(lib().ndt_align_batch_trace(
                self.ctx.h, self.h, scans.ctypes.data, offsets.ctypes.data, B, int(shared_scan),
                inits.ctypes.data, res.ctypes.data, trace.ctypes.data, trace_cap, rows.ctypes.data),
                "ndt_align_batch_trace")
            return res, [trace[b, :min(rows[b], trace_cap)] for b in range(B)]
        self.ctx.check(lib().ndt_align_batch(self.ctx.h, self.h, scans.ctypes.data, offsets.ctypes.data, B,
                                             int(shared_scan), inits.ctypes.data, res.ctypes.data),
                       "ndt_align_batch")
        return res

    def align_batch_dev(self, scans_ptr, offsets_ptr, B, total_points, inits_ptr, out_ptr, shared_scan=False,
                        stream=None, ctx=None):
        """All pointers are device addresses (e.g. torch.Tensor.data_ptr()); asynchronous.  `ctx`: the context
        whose scratch the launch uses (default: the map's own) -- two contexts keep two batches in flight."""
        cx = ctx if ctx is not None else self.ctx
        cx.check(lib().ndt_align_batch_dev(cx.h, self.h, scans_ptr, offsets_ptr, B, total_points,
                                           int(shared_scan), inits_ptr, out_ptr, stream), "ndt_align_batch_dev")

    def prepare_batch_dev(self, scans_ptr, offsets_ptr, B, total_points, inits_ptr, shared_scan=False, stream=None, ctx=None):
        """ndt_align_batch_prepare_dev: the optimiser's start, the window geometry and the voxel order of every scan of a batch,
        queued on `stream` ahead of the align_batch_dev call with the same arguments (and the same `ctx`); asynchronous."""
        cx = ctx if ctx is not None else self.ctx
        cx.check(lib().ndt_align_batch_prepare_dev(cx.h, self.h, scans_ptr, offsets_ptr, B, total_points,
                                                   int(shared_scan), inits_ptr, stream), "ndt_align_batch_prepare_dev")

    def eval_at(self, scan, p):
        scan = _f32c(scan)
        p = np.ascontiguousarray(p, dtype=np.float64)
        s = C.c_double(); pr = C.c_double(); g = np.zeros(3); H = np.zeros(9)
        self.ctx.check(lib().ndt_eval_at(self.ctx.h, self.h, scan.ctypes.data, len(scan), 8, p.ctypes.data,
                                         C.addressof(s), g.ctypes.data, H.ctypes.data, C.addressof(pr)),
                       "ndt_eval_at")
        return s.value, g, H.reshape(3, 3), pr.value

    def fitness_at(self, scan, c, s, tx, ty):
        scan = _f32c(scan)
        f = C.c_double()
        self.ctx.check(lib().ndt_fitness_at(self.ctx.h, self.h, scan.ctypes.data, len(scan), 8, c, s, tx, ty,
                                            C.addressof(f)), "ndt_fitness_at")
        return f.value

    def close(self):
        if self.h:
            lib().ndt_map_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass
