"""Diagnostic: wall-clock of the drop-in call sequence of one estimatePose (host pointers in, host results out)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ndt_slam_amd import capi, synth
from ndt_slam_amd.pose_estimator import PoseEstimator, Pose2D, Scan2D
cfg = synth.CONFIGS["C2"]
m = synth.make_map(cfg["n_map"], cfg["half"])
sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
ctx = capi.Context(0)
prm = capi.default_params(resolution=cfg["resolution"])
gm = capi.Map(ctx, m, prm)
scan, truth, init = sf.make(0)
for what, fn in (("ndt_map_build (1M pts from host memory)", lambda: gm.rebuild(xy=m)),
                 ("ndt_align (10k pts from host memory)", lambda: gm.align(scan, init)),
                 ("ndt_prefilter (30k raw pts, host memory)", lambda: ctx.prefilter(np.repeat(scan, 3, axis=0), 0.05))):
    fn(); ts = []
    for _ in range(10):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    print("%-45s median %.3f ms  min %.3f ms" % (what, np.median(ts) * 1e3, np.min(ts) * 1e3))
pe = PoseEstimator(ctx, Resolution=cfg["resolution"], LeafSize=0.05)
raw = Scan2D(np.repeat(scan, 3, axis=0).astype(np.float64))
pe.setScanPair(raw, m)
ip = Pose2D(init[0], init[1], np.degrees(init[2]))
pe.estimatePose(ip); ts = []
for _ in range(10):
    t = time.perf_counter(); cost, est, cov = pe.estimatePose(ip); ts.append(time.perf_counter() - t)
print("%-45s median %.3f ms  min %.3f ms  (cost %.2e)" % ("PoseEstimator.estimatePose (Python mirror)", np.median(ts) * 1e3, np.min(ts) * 1e3, cost))
