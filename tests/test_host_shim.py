"""The C++ PoseEstimator mirror (ndt_slam_amd/host) -- the class a maintainer links instead of the
reference's src/PoseEstimator.cpp -- against the Python mirror and the oracle."""
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ndt_slam_amd", "host")


def test_host_shim_builds_and_exports_the_reference_interface():
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    syms = subprocess.check_output(["nm", "-DC", os.path.join(HOST, "libndt_pose_estimator.so")], text=True)
    for name in ("ndt_amd::PoseEstimator::setScanPair", "ndt_amd::PoseEstimator::estimatePose",
                 "ndt_amd::PoseEstimator::PoseEstimator", "ndt_amd::approximateVoxelGrid"):
        assert name in syms, name


def test_host_shim_reports_sentinel_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    m = np.random.default_rng(0).uniform(0, 5, (200, 2)).astype(np.float32)
    s = m[:50].astype(np.float64)
    m.tofile(tmp_path / "m.f32"); s.tofile(tmp_path / "s.f64")
    out = subprocess.check_output([os.path.join(HOST, "shim_main"), str(tmp_path / "m.f32"), "200",
                                   str(tmp_path / "s.f64"), "50", "0", "0", "0", "0.5", "0.05"], text=True)
    assert float(out.split()[0]) == 10000000.0      # no device: the reference's failure sentinel, no CPU path


@pytest.mark.gpu
def test_host_shim_matches_python_mirror_and_oracle(tmp_path, oracle, c1_world):
    from ndt_slam_amd import capi
    from ndt_slam_amd.pose_estimator import PoseEstimator, Pose2D, Scan2D, RAD2DEG
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    m, sf, cfg = c1_world
    scan, truth, init = sf.make(4)
    m.astype(np.float32).tofile(tmp_path / "m.f32")
    scan.astype(np.float64).tofile(tmp_path / "s.f64")
    th = RAD2DEG(init[2])
    out = subprocess.check_output([os.path.join(HOST, "shim_main"), str(tmp_path / "m.f32"), str(len(m)),
                                   str(tmp_path / "s.f64"), str(len(scan)), repr(float(init[0])), repr(float(init[1])),
                                   repr(float(th)), "0.3", "0.05"], text=True).split()
    v = [float(x) for x in out[:13]]
    est = PoseEstimator(capi.Context(0), Resolution=0.3, LeafSize=0.05)
    est.setScanPair(Scan2D(scan.astype(np.float64)), m)
    cost, pose, cov = est.estimatePose(Pose2D(init[0], init[1], th))
    assert v[0] == cost and (v[1], v[2], v[3]) == (pose.tx, pose.ty, pose.th)
    assert np.array(v[4:13]).reshape(3, 3) == pytest.approx(cov, rel=1e-9)
    filtered = oracle.approx_voxel_filter(scan, 0.05)
    ref = oracle.Map(m, oracle.default_params(resolution=0.3)).align(filtered, [init[0], init[1], th * math.pi / 180])
    assert v[0] == pytest.approx(ref["fitness"], rel=1e-10)
    assert abs(v[1] - ref["pose"][0]) <= 1e-4 and abs(v[2] - ref["pose"][1]) <= 1e-4
    assert abs(v[3] * math.pi / 180 - ref["pose"][2]) <= 1e-4


def test_host_map_mirror_exports_the_reference_interface():
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    syms = subprocess.check_output(["nm", "-DC", os.path.join(HOST, "libndt_pose_estimator.so")], text=True)
    for name in ("ndt_amd::Submap::makeMap", "ndt_amd::Submap::filterPoints", "ndt_amd::PointCloudMap::addPose",
                 "ndt_amd::PointCloudMap::addPoints", "ndt_amd::PointCloudMap::makeLocalMap",
                 "ndt_amd::PointCloudMap::makeGlobalMap"):
        assert name in syms, name


@pytest.mark.gpu
def test_host_map_mirror_matches_the_python_harness_and_the_oracle(tmp_path, oracle):
    """The C++ Submap / PointCloudMap over the C ABI fed like ScanMatcher::growMap feeds it: same local and global
    maps as the Python mirror on the device and as the same bookkeeping on the oracle (two submaps are opened)."""
    from ndt_slam_amd import capi, replay, synth
    from ndt_slam_amd.pose_estimator import Pose2D
    from replay_helpers import OracleOps
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    recs, truth = synth.replay_records(n_frames=14, n_beams=361, step=0.6)
    scans = [replay.resample_points(r["front"], 0.05, 0.25) for r in recs]
    reg = []                                              # registered with the true poses (double arithmetic)
    for s, t in zip(scans, truth):
        p = Pose2D(*t)
        reg.append(np.stack([p.Rmat[0][0] * s[:, 0] + p.Rmat[0][1] * s[:, 1] + p.tx,
                             p.Rmat[1][0] * s[:, 0] + p.Rmat[1][1] * s[:, 1] + p.ty], 1))
    off = np.zeros(len(reg) + 1, np.uint64)
    off[1:] = np.cumsum([len(r) for r in reg])
    np.concatenate(reg).astype(np.float64).tofile(tmp_path / "xy.f64")
    off.tofile(tmp_path / "off.u64")
    truth.astype(np.float64).tofile(tmp_path / "poses.f64")
    kw = dict(sepThre=3.0, removeMoving=True, LeafSize=0.05, resol=0.05, thre_neighbor=0.2)
    out = subprocess.check_output([os.path.join(HOST, "localmap_main"), str(tmp_path / "xy.f64"), str(tmp_path / "off.u64"),
                                   str(len(reg)), str(tmp_path / "poses.f64"), "3.0", "1", "0.05", "0.05", "0.2",
                                   str(tmp_path / "out.bin")], text=True).split()
    nl, ng, nsub = int(out[0]), int(out[1]), int(out[2])
    raw = open(tmp_path / "out.bin", "rb").read()
    loc = np.frombuffer(raw, np.float32, 2 * nl, 24).reshape(-1, 2)
    glo = np.frombuffer(raw, np.float32, 2 * ng, 24 + 8 * nl).reshape(-1, 2)
    for ops in (capi.Context(0), OracleOps(oracle)):
        pm = replay.PointCloudMap(ops, sepThre=kw["sepThre"], removeMoving=True, LeafSize=0.05, resol=0.05,
                                  thre_neighbor=0.2)
        for r, t in zip(reg, truth):
            pose = Pose2D(*t)
            pm.addPose(pose); pm.addPoints(r); pm.setLastPose(pose); pm.makeLocalMap()
        pm.makeGlobalMap()
        assert len(pm.submaps) == nsub >= 2 and abs(pm.atd - float(out[3])) < 1e-12
        assert pm.localMap_cloud.tobytes() == loc.tobytes() and pm.globalMap_cloud.tobytes() == glo.tobytes()
