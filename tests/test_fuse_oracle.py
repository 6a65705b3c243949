"""CPU checks of the row-f2 restatement (odometry prediction + EKF fusion; oracle/ndt_oracle.c):
properties the reference's equations imply (src/Pose2D.cpp:5-37, src/PoseFuser.cpp:3-61,
src/MyUtil.cpp:4-23, src/ScanMatcher.cpp:50-67) and an independent numpy evaluation of them."""
import numpy as np
import pytest


def result_record(oracle, pose, H, fitness=1e-4, converged=1):
    r = np.zeros(1, dtype=oracle.RESULT_DTYPE)[0]
    r["pose"] = pose; r["H"] = np.asarray(H, float).ravel(); r["fitness"] = fitness; r["converged"] = converged
    return r


def rot(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])


def test_prediction_is_the_odometry_increment_applied_to_the_last_pose(oracle):
    rng = np.random.default_rng(1)
    for _ in range(50):
        prev = np.array([*rng.uniform(-50, 50, 2), rng.uniform(-180, 180)])
        cur = np.array([*(prev[:2] + rng.uniform(-1, 1, 2)), rng.uniform(-180, 180)])
        last = np.array([*rng.uniform(-50, 50, 2), rng.uniform(-180, 180)])
        motion, pred = oracle.predict(cur, prev, last)
        assert motion[:2] == pytest.approx(rot(prev[2]).T @ (cur[:2] - prev[:2]), abs=1e-12)
        assert pred[:2] == pytest.approx(last[:2] + rot(last[2]) @ motion[:2], abs=1e-12)
        assert -180 <= motion[2] < 180 and -180 <= pred[2] < 180
        assert ((motion[2] - (cur[2] - prev[2])) % 360) == pytest.approx(0, abs=1e-9) or \
               ((motion[2] - (cur[2] - prev[2])) % 360) == pytest.approx(360, abs=1e-9)
        # with the last pose equal to the previous odometry pose the prediction is the current one
        _, p2 = oracle.predict(cur, prev, prev)
        assert p2[:2] == pytest.approx(cur[:2], abs=1e-12)
    m, p = oracle.predict([0, 0, 179.0], [0, 0, -179.0], [0, 0, 179.5])     # wrap: -2 deg, then 177.5
    assert m[2] == pytest.approx(-2.0) and p[2] == pytest.approx(177.5)
    m, p = oracle.predict([0, 0, 10.0], [0, 0, 0.0], [0, 0, 170.0])         # 180 wraps to -180 (sum >= 180)
    assert p[2] == -180.0


def test_fusion_matches_a_numpy_kalman_update(oracle):
    rng = np.random.default_rng(2)
    prm = oracle.default_fuse_params(score_thre=0.5, coe_ndt_cov=0.7)
    for k in range(40):
        last = np.array([*rng.uniform(-20, 20, 2), rng.uniform(-180, 180)])
        motion = np.array([*rng.uniform(-0.5, 0.5, 2), rng.uniform(-5, 5)])
        pred = np.array([*(last[:2] + rot(last[2]) @ motion[:2]), last[2] + motion[2]])
        A = rng.normal(size=(3, 3)); lc = A @ A.T * 1e-3
        Bm = rng.normal(size=(3, 3)); H = -(Bm @ Bm.T + np.eye(3)) * 50.0
        est_rad = np.array([pred[0] + 0.03, pred[1] - 0.02, np.deg2rad(pred[2] + 0.4)])
        est_rad[2] = (est_rad[2] + np.pi) % (2 * np.pi) - np.pi
        ok, fused, cov = oracle.fuse(result_record(oracle, est_rad, H), pred, motion, last, lc, prm)
        assert ok == 1
        dt = prm.del_time
        v = np.hypot(motion[0], motion[1]) / dt; om = np.deg2rad(motion[2] / dt)
        th = np.deg2rad(last[2])
        F = np.array([[1, 0, -v * dt * np.sin(th)], [0, 1, v * dt * np.cos(th)], [0, 0, 1]])
        Am = np.array([[dt * np.cos(th), 0], [dt * np.sin(th), 0], [0, dt]])
        M = np.diag([prm.coe_vel * v * v, prm.coe_omega * om * om])
        ch = F @ lc @ F.T + Am @ M @ Am.T
        Q = np.linalg.inv(-H) * prm.coe_ndt_cov
        K = ch @ np.linalg.inv(Q + ch)
        dth = (np.rad2deg(est_rad[2]) - pred[2] + 180) % 360 - 180
        mu = K @ np.array([est_rad[0] - pred[0], est_rad[1] - pred[1], np.deg2rad(dth)]) + \
            np.array([pred[0], pred[1], np.deg2rad(pred[2])])
        assert cov == pytest.approx((np.eye(3) - K) @ ch, rel=1e-9, abs=1e-15)
        assert fused[:2] == pytest.approx(mu[:2], abs=1e-10)
        assert fused[2] == pytest.approx(np.rad2deg(mu[2]), abs=1e-8)


def test_rejected_and_unconverged_matches_keep_the_prediction(oracle):
    prm = oracle.default_fuse_params(score_thre=0.5)
    last, motion, pred = [1.0, 2.0, 30.0], [0.2, 0.0, 1.0], [1.17, 2.1, 31.0]
    lc = np.eye(3) * 1e-4
    H = -np.eye(3) * 100
    for rec in (result_record(oracle, [1.2, 2.1, 0.5], H, fitness=0.7),               # cost above the threshold
                result_record(oracle, [1.2, 2.1, 0.5], H, fitness=1e-4, converged=0)):   # 1e7 sentinel
        ok, fused, cov = oracle.fuse(rec, pred, motion, last, lc, prm)
        assert ok == 0 and list(fused) == pred
        assert cov[0, 0] > lc[0, 0] and np.allclose(cov, cov.T)
    # a confident match (huge -H => tiny Q) pulls the fused pose onto the estimate, a vague one leaves the prediction
    est = np.array([1.25, 2.05, np.deg2rad(31.5)])
    ok, fused, _ = oracle.fuse(result_record(oracle, est, -np.eye(3) * 1e12), pred, motion, last, lc, prm)
    assert ok == 1 and fused[:2] == pytest.approx(est[:2], abs=1e-6) and fused[2] == pytest.approx(31.5, abs=1e-4)
    ok, fused, _ = oracle.fuse(result_record(oracle, est, -np.eye(3) * 1e-9), pred, motion, last, lc, prm)
    assert ok == 1 and fused == pytest.approx(pred, abs=1e-4)
