"""SURVEY.md 8f row f4 on the device: a synthetic log replayed through the harness with every heavy step on the
MI355X (pre-filter, NDT target build and match, Submap::makeMap, filterPoints) against the same replay with the
oracle in their place.  Tolerance of the path: 1e-4 m / 1e-4 rad per pose (the runs are in fact identical)."""
import math

import numpy as np
import pytest

from ndt_slam_amd import replay, synth
from replay_helpers import OracleEstimator, OracleOps

pytestmark = pytest.mark.gpu


def test_replay_matches_the_oracle_pipeline(oracle, tmp_path):
    import torch
    assert torch.cuda.is_available()
    from ndt_slam_amd import capi
    recs, truth = synth.replay_records(n_frames=30, n_beams=361, step=0.5)
    replay.write_log(tmp_path / "log.txt", recs)
    params = dict(replay.LAUNCH_PARAMS, end_frame=30, sepThre=5.0)
    ctx = capi.Context(0)
    dev = replay.SlamLauncher(ctx, **params)
    p_dev = dev.run(replay.read_log(tmp_path / "log.txt", sidelidar=False), poses_name=tmp_path / "dev.txt",
                    map_name=str(tmp_path / "dev.pcd"), separated_map_name=str(tmp_path / "dev_sep"))
    ref = replay.SlamLauncher(OracleOps(oracle), estim=OracleEstimator(oracle, params), **params)
    p_ref = ref.run(replay.read_log(tmp_path / "log.txt", sidelidar=False), poses_name=tmp_path / "ref.txt",
                    map_name=str(tmp_path / "ref.pcd"), separated_map_name=str(tmp_path / "ref_sep"))
    assert len(p_dev) == len(p_ref) == 30 and len(dev.pcmap.submaps) == len(ref.pcmap.submaps) >= 2
    for a, b in zip(p_dev, p_ref):
        assert abs(a.tx - b.tx) <= 1e-4 and abs(a.ty - b.ty) <= 1e-4
        assert abs(math.radians(replay.sub_angle(a.th, b.th))) <= 1e-4
    assert dev.smat.accepted == ref.smat.accepted and sum(dev.smat.accepted) >= 25
    # the files the reference writes: same poses file, same maps
    assert open(tmp_path / "dev.txt").read() == open(tmp_path / "ref.txt").read()
    assert open(tmp_path / "dev.pcd").read() == open(tmp_path / "ref.pcd").read()
    for i in range(len(dev.pcmap.maps)):
        assert open(tmp_path / ("dev_sep%d.pcd" % i)).read() == open(tmp_path / ("ref_sep%d.pcd" % i)).read()
    # and the estimate is a good one: it follows the drive while the odometry drifts
    est = np.array([[p.tx, p.ty] for p in p_dev])
    odo = np.array([[r["x"], r["y"]] for r in recs])
    assert np.linalg.norm(est - truth[:, :2], axis=1).max() < 0.08
    assert np.linalg.norm(odo - truth[:, :2], axis=1).max() > 0.1
    # the cart that crossed the hall (y = -5.5 +- 0.25, x from -9 to 4) left almost nothing in the submap clouds
    g = np.concatenate([s.p_cloud for s in dev.pcmap.submaps[:-1]])
    cart = (np.abs(g[:, 1] + 5.5) < 0.3) & (g[:, 0] > -9.5) & (g[:, 0] < 5)
    assert cart.sum() < 0.02 * len(g)
