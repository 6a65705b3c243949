#!/usr/bin/env python3
"""Generates tests/golden/front_golden.npz: vectors for the rows around the match.

As for c1_golden.npz the reference ships nothing to compare with and its own sources for these rows need PCL /
ROS / Eigen, so the vectors come from this repo's CPU restatement (oracle/ndt_oracle.c, ndt_oracle_octree.c) and
pin it -- and the HIP path -- against silent drift; they are NOT outputs of the reference.

Contents: a submap of registered scans with a moving object -> PCFilter::difference_extraction of one triple
(indices in the octree's leaf order), Submap::makeMap for three flag combinations, Submap::filterPoints of the
result; odometry prediction + EKF fusion vectors; and the poses of a short replay of a synthetic log.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ndt_slam_amd import replay, synth      # noqa: E402
from oracle import ndt_oracle as O          # noqa: E402
from replay_helpers import OracleEstimator, OracleOps   # noqa: E402


def main():
    scans = synth.submap_scans(5, 700, seed=77)
    off = np.zeros(len(scans) + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in scans])
    base = np.concatenate([scans[0], scans[2]])
    diff_idx = O.difference_indices(base, scans[1], 0.05)
    mm = {}
    for name, (first, newest) in (("first_newest", (True, True)), ("later_newest", (False, True)), ("later_closed", (False, False))):
        mm[name] = O.make_map(scans, first, newest, True, 0.05, 0.2)
    filt = O.approx_voxel_filter(mm["first_newest"], 0.05)
    # prediction / fusion
    rng = np.random.default_rng(5)
    prm = O.default_fuse_params(coe_omega=0.5, score_thre=0.5)
    pf = []
    for _ in range(6):
        cur = np.array([*rng.uniform(-5, 5, 2), rng.uniform(-180, 180)])
        prev = cur + np.array([*rng.normal(0, 0.2, 2), rng.normal(0, 3)])
        last = np.array([*rng.uniform(-5, 5, 2), rng.uniform(-179, 179)])
        mo, pr = O.predict(cur, prev, last)
        pf.append(np.concatenate([cur, prev, last, mo, pr]))
    # replay
    recs, truth = synth.replay_records(n_frames=12, n_beams=181, step=0.6)
    scans_r = [replay.Scan2D(r["front"], sid=r["stamp"], pose=replay.Pose2D(r["x"], r["y"], r["th"])) for r in recs]
    params = dict(replay.LAUNCH_PARAMS, end_frame=12, sepThre=4.0)
    sl = replay.SlamLauncher(OracleOps(O), estim=OracleEstimator(O, params), **params)
    poses = sl.run(scans_r)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "front_golden.npz")
    np.savez_compressed(
        out, sub_scans=np.concatenate(scans), sub_offsets=off, diff_idx=diff_idx.astype(np.int32),
        mm_first_newest=mm["first_newest"], mm_later_newest=mm["later_newest"], mm_later_closed=mm["later_closed"],
        filtered=filt, predict=np.stack(pf),
        replay_front=np.concatenate([r["front"] for r in recs]),
        replay_counts=np.array([len(r["front"]) for r in recs]),
        replay_odo=np.array([[r["x"], r["y"], r["th"]] for r in recs]),
        replay_poses=np.array([[p.tx, p.ty, p.th] for p in poses]),
        replay_accepted=np.array(sl.smat.accepted), replay_submaps=np.int32(len(sl.pcmap.submaps)))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
