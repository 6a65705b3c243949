"""Host-side mirror of the reference's PoseEstimator boundary over the C ABI.

Same names, argument meaning and error behaviour as
/root/reference/include/ndt_slam/PoseEstimator.h:36-133 and src/PoseEstimator.cpp:4-69:
poses in and out are in DEGREES (Pose2D.h:14), the covariance is in (m, m, rad), the return
value is the fitness cost in m^2 with the 1e7 sentinel for a non-converged match.
The reference's caller is ScanMatcher::matchScan (src/ScanMatcher.cpp:40,45).

The C++ form of this shim (what a maintainer links instead of PoseEstimator.cpp) is in
INTEGRATION.md and ndt_slam_amd/host/.
"""
import math

import numpy as np

from . import capi

NOT_CONVERGED_COST = 10000000.0   # src/PoseEstimator.cpp:44-46


def DEG2RAD(x):   # include/ndt_slam/MyUtil.h:22
    return x * math.pi / 180


def RAD2DEG(x):   # include/ndt_slam/MyUtil.h:23
    return x * 180 / math.pi


class Pose2D:
    """tx, ty [m], th [deg] (include/ndt_slam/Pose2D.h:11-59)."""

    def __init__(self, tx=0.0, ty=0.0, th=0.0):
        self.setPose(tx, ty, th)

    def setPose(self, x, y, a):
        self.tx, self.ty, self.th = float(x), float(y), float(a)
        r = DEG2RAD(self.th)
        self.Rmat = [[math.cos(r), -math.sin(r)], [math.sin(r), math.cos(r)]]

    def __repr__(self):
        return "Pose2D(%.6f, %.6f, %.6f deg)" % (self.tx, self.ty, self.th)


class Scan2D:
    """sid + odometry pose + scan points (include/ndt_slam/Scan2D.h:15-35); lps is an [n,2] array
    of the LPoint2D x,y doubles (include/ndt_slam/LPoint2D.h:15-22)."""

    def __init__(self, lps, sid=0, pose=None):
        self.sid = sid
        self.pose = pose if pose is not None else Pose2D()
        self.lps = np.asarray(lps, dtype=np.float64).reshape(-1, 2)


def approximate_voxel_grid(xy32, leaf):
    """pcl::ApproximateVoxelGrid::filter on a z = 0 cloud (src/PoseEstimator.cpp:6-10;
    SURVEY.md 8a row a1): 512-slot direct-mapped history, flush on collision, order dependent.
    Host restatement kept for the tests; estimatePose uses the device filter (ndt_prefilter, row f1)."""
    xy32 = np.ascontiguousarray(xy32, dtype=np.float32)
    inv = np.float32(1.0) / np.float32(leaf)
    ix = np.floor(xy32[:, 0] * inv).astype(np.int64)
    iy = np.floor(xy32[:, 1] * inv).astype(np.int64)
    hsh = ((ix * 7171 + iy * 3079) & 511).astype(np.int64)
    h_ix = [0] * 512; h_iy = [0] * 512; h_n = [0] * 512
    h_cx = [np.float32(0)] * 512; h_cy = [np.float32(0)] * 512
    out = []
    for i in range(len(xy32)):
        h = int(hsh[i])
        if h_n[h] and (ix[i] != h_ix[h] or iy[i] != h_iy[h]):
            out.append((h_cx[h] / np.float32(h_n[h]), h_cy[h] / np.float32(h_n[h])))
            h_n[h] = 0; h_cx[h] = np.float32(0); h_cy[h] = np.float32(0)
        h_ix[h] = ix[i]; h_iy[h] = iy[i]; h_n[h] += 1
        h_cx[h] = np.float32(h_cx[h] + xy32[i, 0]); h_cy[h] = np.float32(h_cy[h] + xy32[i, 1])
    for h in range(512):
        if h_n[h]:
            out.append((h_cx[h] / np.float32(h_n[h]), h_cy[h] / np.float32(h_n[h])))
    return np.array(out, dtype=np.float32).reshape(-1, 2)


_LIBM = None


def yaw_from_T_platform(T00, T10):
    """src/PoseEstimator.cpp:31-35 on the float32 matrix entries with THIS platform's asinf / acosf (what std::asin / std::acos
    are for a float argument): the reference's own lines run on the library's T00 / T10, so the reported yaw is the reference's
    on the same machine (ndt_result.pose[2] holds the correctly-rounded model of the same branches)."""
    global _LIBM
    if _LIBM is None:
        import ctypes
        import ctypes.util
        _LIBM = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        for f in (_LIBM.asinf, _LIBM.acosf):
            f.restype = ctypes.c_float
            f.argtypes = [ctypes.c_float]
    c, s = float(np.float32(T00)), float(np.float32(T10))
    if c > 0 and s != 0:
        return float(_LIBM.asinf(s))
    if c < 0 and s > 0:
        return float(_LIBM.acosf(c))
    return float(_LIBM.acosf(c)) * (-1.0)


class PoseEstimator:
    """Drop-in for the reference class of the same name."""

    def __init__(self, ctx=None, coeNDTCov=1.0, TransformationEpsilon=0.01, StepSize=0.1, Resolution=1.0,
                 MaximumIterations=35, LeafSize=0.1, **switches):
        # constructor defaults: include/ndt_slam/PoseEstimator.h:63-64 (launch file overrides them)
        self.ctx = ctx if ctx is not None else capi.Context(0)
        self.coeNDTCov = coeNDTCov
        self.LeafSize = LeafSize
        self.params = capi.default_params(resolution=Resolution, step_size=StepSize,
                                          trans_eps=TransformationEpsilon, max_iter=MaximumIterations,
                                          **dict(dict(grid_margin=8), **switches))   # sliding local map: 8 voxels to spare
        self.totalError = 0.0          # PoseEstimator.h:58 (never written by the reference either)
        self.source_cloud = None
        self.target_cloud = None
        self._map = None
        self.last_result = None

    def setScanPair(self, curScan, refScan):
        """PoseEstimator.h:91-104: LPoint2D doubles -> float32 cloud; the target is taken as is.
        refScan: [m,2] float32 array (the pcl::PointCloud<PointXYZ> of the local map) or a Scan2D
        (PoseEstimator.h:106-128)."""
        self.source_cloud = curScan.lps.astype(np.float32)
        tgt = refScan.lps if isinstance(refScan, Scan2D) else refScan
        self.target_cloud = np.ascontiguousarray(tgt, dtype=np.float32).reshape(-1, 2)

    def estimatePose(self, initPose):
        """src/PoseEstimator.cpp:4-69.  Returns (cost, estPose, cov)."""
        filtered = self.ctx.prefilter(self.source_cloud, self.LeafSize)             # :6-10, on the device (f1)
        # :17-19 -- the target is rebuilt on every call, as the reference does (the local map is
        # refilled in place each scan, src/PointCloudMap.cpp:119-131)
        if self._map is None:
            self._map = capi.Map(self.ctx, self.target_cloud, self.params)
        else:
            self._map.params = self.params
            self._map.rebuild(xy=self.target_cloud)
        est = Pose2D()
        cov = np.full((3, 3), np.nan)
        try:
            r = self._map.align(filtered, [initPose.tx, initPose.ty, DEG2RAD(initPose.th)])   # :22-28
        except capi.NdtError:
            return NOT_CONVERGED_COST, est, cov
        self.last_result = r
        est.setPose(float(r["T03"]), float(r["T13"]), RAD2DEG(yaw_from_T_platform(r["T00"], r["T10"])))   # :29-36
        cost = float(r["fitness"])                                                  # :43
        if not r["converged"]:                                                      # :44-46
            cost = NOT_CONVERGED_COST
        hessian3d = -np.array(r["H"], dtype=np.float64).reshape(3, 3)               # :57-61
        with np.errstate(all="ignore"):
            try:
                cov = np.linalg.inv(hessian3d) * self.coeNDTCov                     # :64
            except np.linalg.LinAlgError:
                cov = np.full((3, 3), np.inf)
        return cost, est, cov
