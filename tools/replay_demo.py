"""End-to-end replay: synthetic log -> poses file + PCD maps, every heavy step on the device;
prints the time per scan next to the same replay with the oracle on one host core.
Usage: python tools/replay_demo.py [n_frames] [n_beams] [--no-oracle]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ndt_slam_amd import capi, replay, synth       # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_frames = int(args[0]) if args else 60
    n_beams = int(args[1]) if len(args) > 1 else 1081
    out = tempfile.mkdtemp(prefix="replay_")
    recs, truth = synth.replay_records(n_frames=n_frames, n_beams=n_beams, step=0.4)
    log = os.path.join(out, "log.txt")
    replay.write_log(log, recs)
    params = dict(replay.LAUNCH_PARAMS, end_frame=n_frames)
    ctx = capi.Context(0)
    spent = {}
    if "--profile" in sys.argv:                            # host-side time per device operation
        def timed(obj, name):
            f = getattr(obj, name)

            def g(*a, **k):
                t0 = time.perf_counter()
                r = f(*a, **k)
                spent[name] = spent.get(name, 0.0) + time.perf_counter() - t0
                return r
            setattr(obj, name, g)
        for name in ("prefilter", "make_map"):
            timed(ctx, name)
    sl = replay.SlamLauncher(ctx, **params)
    if "--profile" in sys.argv:
        timed(sl.estim, "estimatePose")
        timed(sl.smat, "matchScan")
    scans = replay.read_log(log, sidelidar=False)
    n_raw = int(np.mean([len(s.lps) for s in scans]))
    t = time.perf_counter()
    poses = sl.run(scans, poses_name=os.path.join(out, "poses.txt"), map_name=os.path.join(out, "map.pcd"),
                   separated_map_name=os.path.join(out, "sep"))
    t_dev = time.perf_counter() - t
    est = np.array([[p.tx, p.ty] for p in poses])
    err = np.linalg.norm(est - truth[:len(est), :2], axis=1)
    odo = np.array([[r["x"], r["y"]] for r in recs])
    print("replay of %d scans (%d returns each, ~%d points after resampling): %.1f ms per scan on the device path "
          "(host bookkeeping in Python included); accepted %d; max position error %.3f m (odometry alone: %.3f m); "
          "submaps %d; local map %d points" % (
              len(poses), n_raw, int(np.mean([len(s.lps) for s in scans])), 1e3 * t_dev / len(poses),
              sum(sl.smat.accepted), err.max(), np.linalg.norm(odo - truth[:, :2], axis=1).max(), len(sl.pcmap.submaps),
              len(sl.pcmap.localMap_cloud)))
    if spent:
        print("host time per scan [ms]:", {k: round(1e3 * v / len(poses), 3) for k, v in spent.items()},
              "(estimatePose contains one prefilter; matchScan contains everything)")
    if "--no-oracle" not in sys.argv:
        from oracle import ndt_oracle as O                 # checker / CPU figure only
        from replay_helpers import OracleEstimator, OracleOps
        ref = replay.SlamLauncher(OracleOps(O), estim=OracleEstimator(O, params), **params)
        t = time.perf_counter()
        p_ref = ref.run(replay.read_log(log, sidelidar=False))
        t_ref = time.perf_counter() - t
        d = max(max(abs(a.tx - b.tx), abs(a.ty - b.ty)) for a, b in zip(poses, p_ref))
        print("same replay with the oracle on one host core: %.1f ms per scan; max pose difference %.2e m" % (
            1e3 * t_ref / len(p_ref), d))
    print("files:", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
