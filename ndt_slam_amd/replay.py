"""Replay harness: the reference's log-driven pipeline around the match, so that a recorded
(or synthetic) log goes in and the poses file and PCD maps of the reference come out, with every heavy step on
the device through the C ABI.

Host-side mirrors, same names and argument meaning as the reference:
  * SlamLauncher   -- text-log reader (src/SlamLauncher.cpp:37-105, 4 header lines: SlamLauncher.h:91-101),
                      pose dump (src/SlamLauncher.cpp:30-35), loop (:107-141)
  * ScanPointResampler (src/ScanPointResampler.cpp:4-62)
  * ScanMatcher    -- matchScan / growMap (src/ScanMatcher.cpp:4-116)
  * PoseFuser      -- fusePose / calOdometryCovariance (src/PoseFuser.cpp:3-61), Pose2D::calMotion / calPredPose
                      (src/Pose2D.cpp:5-37), MyUtil::add_angle / sub_angle (src/MyUtil.cpp:4-23)
  * Submap / PointCloudMap (include/ndt_slam/PointCloudMap.h:22-151, src/PointCloudMap.cpp:4-134) with PCD ASCII
                      output (PointCloudMap.h:124-136)
  * FrontEnd::process (src/FrontEnd.cpp:4-47)

What runs on the MI355X: the source pre-filter and NDT match (PoseEstimator -> ndt_prefilter, ndt_map_build,
ndt_align), Submap::makeMap (ndt_make_map: octree change detection + moving-object removal) and
Submap::filterPoints (ndt_prefilter).  `ops` is the object that provides them (capi.Context); the bookkeeping
and the 3x3 filter algebra stay on the host as in the reference.  ROS publishing (tf, PoseArray, RViz clouds)
is left out: it does not feed back into the estimate.
"""
import math

import numpy as np

from .pose_estimator import DEG2RAD, RAD2DEG, Pose2D, PoseEstimator, Scan2D

# launch-file parameters (ndt_mapping.launch:8-36); constructor defaults differ and are noted at their classes
LAUNCH_PARAMS = dict(
    draw_skip=5, start_frame=0, end_frame=690, keyframe_skip=5, score_thre=0.5, space=0.05, space_thre=0.25,
    sepThre=10.0, removeMoving=True, resol=0.05, thre_neighbor=0.2, sidelidar=False, delTime=0.5, coeVel=0.1,
    coeOmega=0.5, coeNDTCov=1.0, TransformationEpsilon=0.01, StepSize=0.1, Resolution=0.3, MaximumIterations=35,
    LeafSize=0.05)


# ---------------------------------------------------------------------------------------------------------
# log file (SlamLauncher)
# ---------------------------------------------------------------------------------------------------------
def write_log(path, records, header=("# ndt_slam log", "# stamp x y theta[deg] image", "# n_front x y ...",
                                     "# n_left ... / n_right ...")):
    """records: iterable of dict(stamp, x, y, th, image, front[n,2], left[n,2], right[n,2]) -> the text format
    SlamLauncher::input_file_line reads: four free header lines, then per scan one line `stamp x y th image` and
    three groups `count x y x y ... ` (front, left, right lidar), every token followed by one space."""
    with open(path, "w") as f:
        for h in header:
            f.write(h + "\n")
        for r in records:
            f.write("%d %.9g %.9g %.9g %s\n" % (r["stamp"], r["x"], r["y"], r["th"], r.get("image", "none")))
            for key in ("front", "left", "right"):
                pts = np.asarray(r.get(key, np.zeros((0, 2))), dtype=np.float64).reshape(-1, 2)
                f.write("%d " % len(pts))
                f.write("".join("%.9g %.9g " % (p[0], p[1]) for p in pts))
                f.write("\n")


def read_log(path, sidelidar=True):
    """-> list of Scan2D (sid = stamp, pose = odometry in degrees, lps = front (+ left + right when sidelidar),
    SlamLauncher.cpp:37-93).  Every complete record is returned; the reference itself stops at end_frame and
    either drops or trips over the record that meets the end of the file (:95-98)."""
    with open(path) as f:
        for _ in range(4):                                   # readFormat(): four header lines
            f.readline()
        text = f.read()
    scans, pos = [], 0
    n = len(text)

    def token():
        nonlocal pos
        while pos < n and text[pos].isspace():
            pos += 1
        a = pos
        while pos < n and not text[pos].isspace():
            pos += 1
        return text[a:pos]

    while True:
        t = token()
        if not t:
            break
        stamp = int(t)
        x, y, th = float(token()), float(token()), float(token())
        eol = text.find("\n", pos)                           # image name: the rest of the line
        pos = n if eol < 0 else eol + 1
        groups = []
        ok = True
        for _ in range(3):
            c = token()
            if not c:
                ok = False
                break
            pts = np.empty((int(c), 2), dtype=np.float64)
            for i in range(int(c)):
                a, b = token(), token()
                if not b:
                    ok = False
                    break
                pts[i] = (float(a), float(b))
            if not ok:
                break
            groups.append(pts)
        if not ok:
            break                                            # truncated record at the end of the file
        lps = groups[0] if not sidelidar else np.concatenate(groups)
        scans.append(Scan2D(lps, sid=stamp, pose=Pose2D(x, y, th)))
    return scans


def write_poses(path, poses):
    """SlamLauncher::output_file_poses (:30-35): the count, then every 10th pose as `tx ty th ` (th in degrees),
    numbers as operator<< prints doubles (6 significant digits)."""
    with open(path, "w") as f:
        f.write("%d\n" % len(poses))
        for i in range(0, len(poses), 10):
            f.write("%s %s %s \n" % tuple(_cout(v) for v in (poses[i].tx, poses[i].ty, poses[i].th)))


def _cout(v):
    return "%g" % v          # std::ostream default: precision 6, general format


def save_pcd_ascii(path, xy):
    """pcl::io::savePCDFileASCII of a PointXYZ cloud with z = 0 (PointCloudMap.h:124-136): PCD v0.7 header,
    one `x y z` line per point, 8 significant digits (PCL's default precision)."""
    xy = np.asarray(xy, dtype=np.float32).reshape(-1, 2)
    with open(path, "w") as f:
        f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n"
                "COUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA ascii\n" % (len(xy), len(xy)))
        for p in xy:
            f.write("%.8g %.8g 0\n" % (p[0], p[1]))


def load_pcd_ascii(path):
    with open(path) as f:
        lines = f.read().split("\n")
    i = next(k for k, l in enumerate(lines) if l.startswith("DATA"))
    rows = [l.split() for l in lines[i + 1:] if l.strip()]
    return np.array([[float(r[0]), float(r[1])] for r in rows], dtype=np.float32).reshape(-1, 2)


# ---------------------------------------------------------------------------------------------------------
# ScanPointResampler
# ---------------------------------------------------------------------------------------------------------
def resample_points(lps, space, space_thre):
    """ScanPointResampler::resamplePoints (src/ScanPointResampler.cpp:4-62): walk the scan, drop points until the
    accumulated distance reaches `space`, interpolate a point at exactly `space` (and look at the same input
    point again), keep the point itself across gaps of `space_thre` or more.  lps: [n,2] doubles."""
    lps = np.asarray(lps, dtype=np.float64).reshape(-1, 2)
    if len(lps) == 0:
        return lps
    out = [(lps[0, 0], lps[0, 1])]
    dis = 0.0
    prev = (lps[0, 0], lps[0, 1])
    i = 1
    n = len(lps)
    while i < n:
        cx, cy = lps[i, 0], lps[i, 1]
        dx, dy = cx - prev[0], cy - prev[1]
        L = math.sqrt(dx * dx + dy * dy)
        if dis + L < space:                                   # findInterpolatePoint: too close, dropped
            dis += L
            prev = (cx, cy)
        elif dis + L >= space_thre:                           # a gap: the point is kept as it is
            out.append((cx, cy))
            prev = (cx, cy)
            dis = 0.0
        else:                                                 # interpolate at `space` along prev -> current
            ratio = (space - dis) / L
            npt = (dx * ratio + prev[0], dy * ratio + prev[1])
            out.append(npt)
            prev = npt
            dis = 0.0
            continue                                          # inserted before lp: lp is looked at again (i--)
        i += 1
    return np.array(out, dtype=np.float64).reshape(-1, 2)


# ---------------------------------------------------------------------------------------------------------
# Pose2D helpers and PoseFuser
# ---------------------------------------------------------------------------------------------------------
def add_angle(a1, a2):                                        # src/MyUtil.cpp:4-12
    s = a1 + a2
    if s < -180:
        s += 360
    elif s >= 180:
        s -= 360
    return s


def sub_angle(a1, a2):                                        # src/MyUtil.cpp:15-23
    d = a1 - a2
    if d < -180:
        d += 360
    elif d >= 180:
        d -= 360
    return d


def calMotion(cur, prev):                                     # src/Pose2D.cpp:5-16
    dx, dy = cur.tx - prev.tx, cur.ty - prev.ty
    return Pose2D(prev.Rmat[0][0] * dx + prev.Rmat[1][0] * dy, prev.Rmat[0][1] * dx + prev.Rmat[1][1] * dy,
                  sub_angle(cur.th, prev.th))


def calPredPose(motion, last):                                # src/Pose2D.cpp:28-37
    return Pose2D(last.Rmat[0][0] * motion.tx + last.Rmat[0][1] * motion.ty + last.tx,
                  last.Rmat[1][0] * motion.tx + last.Rmat[1][1] * motion.ty + last.ty,
                  add_angle(last.th, motion.th))


class PoseFuser:
    """src/PoseFuser.cpp; constructor defaults include/ndt_slam/PoseFuser.h:19 (0.1, 0.1, 0.5)."""

    def __init__(self, coeVel=0.1, coeOmega=0.1, delTime=0.5):
        self.coeVel, self.coeOmega, self.delTime = coeVel, coeOmega, delTime

    def calOdometryCovariance(self, odoMotion, lastPose, lastCov):           # :39-61
        dt = self.delTime
        v = math.sqrt(odoMotion.tx * odoMotion.tx + odoMotion.ty * odoMotion.ty) / dt
        omega = DEG2RAD(odoMotion.th / dt)
        M = np.array([[self.coeVel * v * v, 0.0], [0.0, self.coeOmega * omega * omega]])
        c, s = math.cos(DEG2RAD(lastPose.th)), math.sin(DEG2RAD(lastPose.th))
        A = np.array([[dt * c, 0.0], [dt * s, 0.0], [0.0, dt]])
        F = np.array([[1.0, 0.0, -v * dt * s], [0.0, 1.0, v * dt * c], [0.0, 0.0, 1.0]])
        return F @ lastCov @ F.T + A @ M @ A.T

    def fusePose(self, predPose, estPose, odoMotion, lastPose, lastCov, Qmat):  # :3-37
        cov_hat = self.calOdometryCovariance(odoMotion, lastPose, lastCov)
        mu_hat = np.array([predPose.tx, predPose.ty, DEG2RAD(predPose.th)])
        K = cov_hat @ np.linalg.inv(Qmat + cov_hat)
        cov = (np.eye(3) - K) @ cov_hat
        zh = np.array([estPose.tx - predPose.tx, estPose.ty - predPose.ty, DEG2RAD(sub_angle(estPose.th, predPose.th))])
        mu = K @ zh + mu_hat
        return Pose2D(mu[0], mu[1], RAD2DEG(mu[2])), cov


# ---------------------------------------------------------------------------------------------------------
# Submap / PointCloudMap
# ---------------------------------------------------------------------------------------------------------
_EMPTY = np.zeros((0, 2), dtype=np.float32)


class Submap:
    """include/ndt_slam/PointCloudMap.h:22-69; constructor defaults LeafSize 0.2, removeMoving false."""

    def __init__(self, ops, atdS=0.0, cntS=0, removeMoving=False, LeafSize=0.2, resol=0.05, thre_neighbor=0.1):
        self.ops = ops
        self.atdS, self.cntS, self.cntE, self.newest = atdS, cntS, -1, True
        self.removeMoving, self.LeafSize = removeMoving, LeafSize
        self.resol, self.thre_neighbor = resol, thre_neighbor          # PCFilter.h:20-23
        self.scans = []
        self.p_cloud = _EMPTY

    def addPoints(self, cloud):
        self.scans.append(cloud)

    def filterPoints(self):                                             # src/PointCloudMap.cpp:4-13
        if len(self.p_cloud) == 0:
            return _EMPTY
        return self.ops.prefilter(self.p_cloud, self.LeafSize)

    def makeMap(self):                                                  # src/PointCloudMap.cpp:15-39
        self.p_cloud = self.ops.make_map(self.scans, self.cntS == 0, self.newest, self.removeMoving, self.resol,
                                         self.thre_neighbor)


class PointCloudMap:
    """include/ndt_slam/PointCloudMap.h:72-151, src/PointCloudMap.cpp:44-134; default sepThre 30."""

    def __init__(self, ops, sepThre=30.0, **submap_kw):
        self.ops, self.sepThre, self.submap_kw = ops, sepThre, submap_kw
        self.poses = []
        self.lastPose = Pose2D()
        self.lastScan = None
        self.atd = 0.0
        self.globalMap_cloud = _EMPTY
        self.localMap_cloud = _EMPTY
        self.maps = []
        self.submaps = [Submap(ops, **submap_kw)]

    def setLastPose(self, p):
        self.lastPose = p

    def getLastPose(self):
        return self.lastPose

    def setLastScan(self, s):
        self.lastScan = s

    def addPose(self, p):                                               # :44-56
        if self.poses:
            pp = self.poses[-1]
            self.atd += math.sqrt((p.tx - pp.tx) * (p.tx - pp.tx) + (p.ty - pp.ty) * (p.ty - pp.ty))
        else:
            self.atd = 0.0
        self.poses.append(p)

    def addPoints(self, lps):                                           # :59-96
        cloud = np.ascontiguousarray(lps, dtype=np.float32).reshape(-1, 2)   # double -> PointXYZ floats, z = 0
        cur = self.submaps[-1]
        if self.atd - cur.atdS >= self.sepThre:
            size = len(self.poses)
            cur.cntE = size - 2
            cur.p_cloud = cur.filterPoints()
            cur.newest = False
            sub = Submap(self.ops, self.atd, size - 1, **self.submap_kw)
            if len(cur.scans) >= 2:                                     # two scans of overlap for the triple test
                sub.addPoints(cur.scans[-2])
                sub.addPoints(cur.scans[-1])
            sub.addPoints(cloud)
            sub.makeMap()
            self.submaps.append(sub)
        else:
            cur.addPoints(cloud)
            cur.makeMap()

    def makeGlobalMap(self):                                            # :101-117
        self.maps = [s.p_cloud for s in self.submaps[:-1]]
        self.maps.append(self.submaps[-1].filterPoints())
        self.globalMap_cloud = np.concatenate(self.maps) if self.maps else _EMPTY

    def makeLocalMap(self):                                             # :119-134
        parts = []
        if len(self.submaps) >= 2:
            parts.append(self.submaps[-2].p_cloud)
        parts.append(self.submaps[-1].filterPoints())
        self.localMap_cloud = np.concatenate(parts)

    def saveGlobalMap(self, map_name, separated_map_name):              # PointCloudMap.h:124-136
        save_pcd_ascii(map_name, self.globalMap_cloud)
        for i, m in enumerate(self.maps):
            save_pcd_ascii("%s%d.pcd" % (separated_map_name, i), m)


# ---------------------------------------------------------------------------------------------------------
# ScanMatcher / FrontEnd / SlamLauncher
# ---------------------------------------------------------------------------------------------------------
class ScanMatcher:
    """src/ScanMatcher.cpp:4-116; default score threshold 0.0 (ScanMatcher.h:49)."""

    def __init__(self, estim, pcmap, pfu, scthre=0.0, space=0.0, space_thre=0.0):
        self.estim, self.pcmap, self.pfu = estim, pcmap, pfu
        self.scthre, self.space, self.space_thre = scthre, space, space_thre
        self.cnt = 0
        self.prevScan = None
        self.poses, self.Covs = [], []
        self.lastCov = np.zeros((3, 3))      # (the reference reads it uninitialised on the second scan, ScanMatcher.h:42)
        self.costs, self.accepted = [], []

    def matchScan(self, curScan):
        curScan.lps = resample_points(curScan.lps, self.space, self.space_thre)         # :6
        if self.cnt == 0:                                                               # :9-22
            self.growMap(curScan, curScan.pose)
            self.savePose(curScan.pose, np.zeros((3, 3)))
            self.prevScan = curScan
            self.cnt += 1
            return True
        odoMotion = calMotion(curScan.pose, self.prevScan.pose)                         # :27-28
        lastPose = self.pcmap.getLastPose()
        predPose = calPredPose(odoMotion, lastPose)                                     # :30-32
        self.estim.setScanPair(curScan, self.pcmap.localMap_cloud)                      # :40
        cost, estPose, Qmat = self.estim.estimatePose(predPose)                         # :45
        successful = cost <= self.scthre                                                # :49-53
        if successful:                                                                  # :58-65
            fusedPose, cov = self.pfu.fusePose(predPose, estPose, odoMotion, lastPose, self.lastCov, Qmat)
        else:
            cov = self.pfu.calOdometryCovariance(odoMotion, lastPose, self.lastCov)
            fusedPose = predPose
        self.lastCov = cov
        self.growMap(curScan, fusedPose)                                                # :71
        self.prevScan = curScan
        self.savePose(fusedPose, cov)
        self.costs.append(cost)
        self.accepted.append(bool(successful))
        self.cnt += 1
        return successful

    def savePose(self, pose, cov):
        self.poses.append(pose)
        self.Covs.append(cov)

    def growMap(self, scan, pose):                                                      # :92-116
        R = pose.Rmat
        x = R[0][0] * scan.lps[:, 0] + R[0][1] * scan.lps[:, 1] + pose.tx
        y = R[1][0] * scan.lps[:, 0] + R[1][1] * scan.lps[:, 1] + pose.ty
        self.pcmap.addPose(pose)
        self.pcmap.addPoints(np.stack([x, y], axis=1))
        self.pcmap.setLastPose(pose)
        self.pcmap.setLastScan(scan)
        self.pcmap.makeLocalMap()


class FrontEnd:
    """src/FrontEnd.cpp:4-47 (the pose-graph and loop-closure parts are commented out in the reference)."""

    def __init__(self, smat, pcmap, keyframeSkip=5, startFrame=0):
        self.smat, self.pcmap = smat, pcmap
        self.keyframeSkip, self.startFrame = keyframeSkip, startFrame
        self.cnt = 0

    def process(self, scan):
        if scan.sid < self.startFrame:
            return
        self.smat.matchScan(scan)
        if self.cnt % self.keyframeSkip == 0:
            self.pcmap.makeGlobalMap()
        self.cnt += 1

    def get_poses(self):
        return self.smat.poses


class SlamLauncher:
    """Wires the pipeline as SlamLauncher::init does (src/SlamLauncher.cpp:7-28) and runs the replay loop
    (:107-141).  `ops` provides prefilter / make_map (a capi.Context); `estim` defaults to the device
    PoseEstimator on the same context."""

    def __init__(self, ops, estim=None, **params):
        p = dict(LAUNCH_PARAMS)
        p.update(params)
        self.p = p
        self.estim = estim if estim is not None else PoseEstimator(
            ctx=ops, coeNDTCov=p["coeNDTCov"], TransformationEpsilon=p["TransformationEpsilon"], StepSize=p["StepSize"],
            Resolution=p["Resolution"], MaximumIterations=p["MaximumIterations"], LeafSize=p["LeafSize"])
        self.pcmap = PointCloudMap(ops, sepThre=p["sepThre"], removeMoving=p["removeMoving"], LeafSize=p["LeafSize"],
                                   resol=p["resol"], thre_neighbor=p["thre_neighbor"])
        self.pfu = PoseFuser(p["coeVel"], p["coeOmega"], p["delTime"])
        self.smat = ScanMatcher(self.estim, self.pcmap, self.pfu, scthre=p["score_thre"], space=p["space"],
                                space_thre=p["space_thre"])
        self.frontEnd = FrontEnd(self.smat, self.pcmap, keyframeSkip=p["keyframe_skip"], startFrame=p["start_frame"])

    def run(self, scans, poses_name=None, map_name=None, separated_map_name=None):
        """scans: list of Scan2D (read_log).  Processes at most end_frame scans, then writes what the reference
        writes: the poses file and, when names are given, the PCD maps.  Returns the list of fused poses."""
        for cnt, scan in enumerate(scans, start=1):
            if cnt > self.p["end_frame"]:
                break
            self.frontEnd.process(scan)
        poses = self.frontEnd.get_poses()
        if poses_name:
            write_poses(poses_name, poses)
        if map_name:
            self.pcmap.saveGlobalMap(map_name, separated_map_name or (map_name + "_sep"))
        return poses
