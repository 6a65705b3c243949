"""the replay harness' host pieces (log format, resampler, pose dump, PCD, bookkeeping) on the
CPU, with the oracle standing in for the device operations."""
import numpy as np
import pytest

from ndt_slam_amd import replay, synth
from ndt_slam_amd.pose_estimator import Pose2D


from replay_helpers import OracleEstimator, OracleOps


def test_log_round_trip(tmp_path):
    recs, _ = synth.replay_records(n_frames=3, n_beams=91)
    recs[1]["left"] = np.array([[1.0, 2.0], [3.0, -4.5]])
    recs[1]["right"] = np.array([[-1.25, 0.5]])
    path = tmp_path / "log.txt"
    replay.write_log(path, recs)
    text = open(path).read().split("\n")
    assert len(text) == 4 + 3 * 4 + 1 and text[4].split(" ")[4] == "img0000.png"
    front = replay.read_log(path, sidelidar=False)
    both = replay.read_log(path, sidelidar=True)
    assert [s.sid for s in front] == [0, 1, 2]
    assert len(both[1].lps) == len(front[1].lps) + 3 and np.allclose(both[1].lps[-1], [-1.25, 0.5])
    assert np.allclose(front[2].lps, recs[2]["front"], rtol=0, atol=1e-8 * 30)
    assert abs(front[2].pose.th - recs[2]["th"]) < 1e-6
    # a record cut off by the end of the file is not returned
    open(path, "w").write("\n".join(text[:4 + 4 + 2]))
    assert len(replay.read_log(path)) == 1


def test_resampler_rules():
    # a straight line sampled every 2 cm -> one point every 5 cm exactly (interpolated)
    line = np.stack([np.arange(0, 1.0001, 0.02), np.zeros(51)], 1)
    out = replay.resample_points(line, 0.05, 0.25)
    assert np.allclose(np.diff(out[:, 0]), 0.05, atol=1e-12) and np.allclose(out[0], line[0])
    # a jump of 1 m is kept as it is, not bridged by interpolated points
    jump = np.array([[0, 0], [0.02, 0], [1.02, 0], [1.04, 0], [1.10, 0]], float)
    out = replay.resample_points(jump, 0.05, 0.25)
    assert out.tolist() == [[0, 0], [1.02, 0], [1.07, 0]]
    # sparse points (every 10 cm) get interpolated in between: 0.05 spacing
    sparse = np.stack([np.arange(0, 0.5001, 0.1), np.zeros(6)], 1)
    out = replay.resample_points(sparse, 0.05, 0.25)
    assert np.allclose(out[:, 0], np.arange(0, 0.5001, 0.05))
    assert len(replay.resample_points(np.zeros((0, 2)), 0.05, 0.25)) == 0


def test_pose_dump_and_pcd(tmp_path):
    poses = [Pose2D(0.1 * i, -1e-5 * i, 179.99 - i) for i in range(25)]
    replay.write_poses(tmp_path / "p.txt", poses)
    lines = open(tmp_path / "p.txt").read().split("\n")
    assert lines[0] == "25" and len(lines) == 1 + 3 + 1
    assert lines[2] == "1 -0.0001 169.99 " and lines[3] == "2 -0.0002 159.99 "
    xy = np.array([[1.5, -2.25], [0.1, 1e-3], [123456.789, 0.0]], np.float32)
    replay.save_pcd_ascii(tmp_path / "m.pcd", xy)
    head = open(tmp_path / "m.pcd").read().split("\n")
    assert head[0].startswith("# .PCD v0.7") and head[6] == "WIDTH 3" and head[9] == "POINTS 3" and head[10] == "DATA ascii"
    assert head[11] == "1.5 -2.25 0" and head[13] == "123456.79 0 0"
    assert np.array_equal(replay.load_pcd_ascii(tmp_path / "m.pcd"), xy)


def test_angles_and_fuser_match_the_oracle(oracle):
    rng = np.random.default_rng(2)
    pf = replay.PoseFuser(0.1, 0.5, 0.5)
    prm = oracle.default_fuse_params(coe_vel=0.1, coe_omega=0.5, del_time=0.5, score_thre=0.5) \
        if hasattr(oracle, "default_fuse_params") else None
    for _ in range(20):
        cur = Pose2D(*rng.uniform(-5, 5, 2), rng.uniform(-180, 180))
        prev = Pose2D(cur.tx + rng.normal(0, 0.2), cur.ty + rng.normal(0, 0.2), replay.add_angle(cur.th, rng.normal(0, 3)))
        last = Pose2D(*rng.uniform(-5, 5, 2), rng.uniform(-180, 180))
        mo = replay.calMotion(cur, prev)
        pr = replay.calPredPose(mo, last)
        o_mo, o_pr = oracle.predict([cur.tx, cur.ty, cur.th], [prev.tx, prev.ty, prev.th], [last.tx, last.ty, last.th])
        assert np.allclose([mo.tx, mo.ty, mo.th], o_mo, atol=1e-12) and np.allclose([pr.tx, pr.ty, pr.th], o_pr, atol=1e-12)
        assert -180 <= pr.th < 180


def test_replay_on_the_oracle(oracle, tmp_path):
    """The whole loop with the oracle as the device: the estimate follows the true drive while the raw odometry
    drifts away, a second submap is opened, the moving cart is not in the map."""
    recs, truth = synth.replay_records(n_frames=16, n_beams=181, step=0.8)
    replay.write_log(tmp_path / "log.txt", recs)
    scans = replay.read_log(tmp_path / "log.txt", sidelidar=False)
    params = dict(replay.LAUNCH_PARAMS, end_frame=16, keyframe_skip=5, sepThre=5.0)
    ops = OracleOps(oracle)
    sl = replay.SlamLauncher(ops, estim=OracleEstimator(oracle, params), **params)
    poses = sl.run(scans, poses_name=tmp_path / "poses.txt", map_name=str(tmp_path / "map.pcd"),
                   separated_map_name=str(tmp_path / "sep"))
    assert len(poses) == 16 and len(sl.pcmap.submaps) >= 2
    est = np.array([[p.tx, p.ty] for p in poses])
    odo = np.array([[r["x"], r["y"]] for r in recs])
    err_est = np.linalg.norm(est - truth[:, :2], axis=1)
    err_odo = np.linalg.norm(odo - truth[:, :2], axis=1)
    assert err_est.max() < 0.08 and err_odo[-1] > 0.1 and sum(sl.smat.accepted) >= 13, (err_est, err_odo, sl.smat.accepted)
    assert open(tmp_path / "poses.txt").read().split("\n")[0] == "16"
    g = replay.load_pcd_ascii(tmp_path / "map.pcd")
    assert len(g) > 500 and (tmp_path / "sep0.pcd").exists() and (tmp_path / "sep1.pcd").exists()
