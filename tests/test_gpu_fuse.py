"""GPU parity tests of row f2: odometry prediction and EKF fusion for a batch on the
device against the CPU oracle.  fp64 throughout; the two sides differ only in libm vs device
sin/cos (1 ulp), hence a 1e-11 relative tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-11


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from ndt_slam_amd import capi
    return capi, capi.Context(0)


def to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def test_predict_and_fuse_match_the_oracle(gpu, oracle):
    import torch
    capi, ctx = gpu
    rng = np.random.default_rng(5)
    B = 300
    prev = np.column_stack([rng.uniform(-50, 50, (B, 2)), rng.uniform(-180, 180, B)])
    cur = np.column_stack([prev[:, :2] + rng.uniform(-1, 1, (B, 2)), rng.uniform(-180, 180, B)])
    last = np.column_stack([rng.uniform(-50, 50, (B, 2)), rng.uniform(-180, 180, B)])
    last[:4, 2] = [179.9, -180.0, 170.0, -179.5]; cur[:4, 2] = [-179.0, 179.0, 10.0, 0.0]; prev[:4, 2] = [179.0, -179.0, 0.0, 0.4]
    d_cur, d_prev, d_last = to_dev(cur), to_dev(prev), to_dev(last)
    d_mo = torch.zeros(B, 3, dtype=torch.float64, device="cuda:0"); d_pred = torch.zeros_like(d_mo); d_init = torch.zeros_like(d_mo)
    ctx.predict_batch_dev(d_cur.data_ptr(), d_prev.data_ptr(), d_last.data_ptr(), B, d_mo.data_ptr(), d_pred.data_ptr(),
                          d_init.data_ptr())
    torch.cuda.synchronize()
    mo, pred, init = d_mo.cpu().numpy(), d_pred.cpu().numpy(), d_init.cpu().numpy()
    for b in range(B):
        m_ref, p_ref = oracle.predict(cur[b], prev[b], last[b])
        assert mo[b] == pytest.approx(m_ref, rel=TOL, abs=1e-12) and pred[b] == pytest.approx(p_ref, rel=TOL, abs=1e-12)
        assert init[b, :2] == pytest.approx(p_ref[:2], rel=TOL, abs=1e-12) and init[b, 2] == pytest.approx(np.deg2rad(p_ref[2]), rel=TOL)

    # results: accepted, above the threshold, not converged, near-singular Hessian
    res = np.zeros(B, dtype=capi.RESULT_DTYPE)
    A = rng.normal(size=(B, 3, 3))
    H = -(A @ np.transpose(A, (0, 2, 1)) + np.eye(3)) * rng.uniform(1, 1e4, (B, 1, 1))
    res["H"] = H.reshape(B, 9)
    res["pose"][:, :2] = pred[:, :2] + rng.normal(0, 0.05, (B, 2))
    res["pose"][:, 2] = (np.deg2rad(pred[:, 2] + rng.normal(0, 1.0, B)) + np.pi) % (2 * np.pi) - np.pi
    res["fitness"] = rng.uniform(0, 1.0, B); res["converged"] = 1
    res["converged"][::7] = 0
    res["H"][5] = (-np.ones((3, 3))).ravel()                     # singular: inf / nan covariance, as in the reference
    Lc = rng.normal(size=(B, 3, 3)); last_cov = (Lc @ np.transpose(Lc, (0, 2, 1))) * 1e-3
    prm_g = capi.default_fuse_params(score_thre=0.5, coe_ndt_cov=0.8)
    prm_o = oracle.default_fuse_params(score_thre=0.5, coe_ndt_cov=0.8)
    d_res = to_dev(np.frombuffer(res.tobytes(), np.uint8).copy()); d_lc = to_dev(last_cov.reshape(B, 9))
    d_fused = torch.zeros(B, 3, dtype=torch.float64, device="cuda:0"); d_cov = torch.zeros(B, 9, dtype=torch.float64, device="cuda:0")
    d_ok = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    ctx.fuse_batch_dev(d_res.data_ptr(), d_pred.data_ptr(), d_mo.data_ptr(), d_last.data_ptr(), d_lc.data_ptr(), B, prm_g,
                       d_fused.data_ptr(), d_cov.data_ptr(), d_ok.data_ptr())
    torch.cuda.synchronize()
    fused, cov, ok = d_fused.cpu().numpy(), d_cov.cpu().numpy().reshape(B, 3, 3), d_ok.cpu().numpy()
    n_acc = 0
    for b in range(B):
        ok_ref, f_ref, c_ref = oracle.fuse(res[b], pred[b], mo[b], last[b], last_cov[b], prm_o)
        assert ok[b] == ok_ref
        n_acc += ok_ref
        if b == 5 and ok_ref:
            assert not np.all(np.isfinite(c_ref)) and not np.all(np.isfinite(cov[b]))
            continue
        scale = np.abs(c_ref).max()
        assert cov[b] == pytest.approx(c_ref, rel=1e-9, abs=1e-12 * scale)
        assert fused[b] == pytest.approx(f_ref, rel=1e-10, abs=1e-10)
    assert 50 < n_acc < B - 50


def test_whole_front_end_step_on_the_device(gpu, oracle, c1_world):
    """predict -> pre-filter -> match -> fuse for a batch without leaving the device, against the same
    chain on the CPU (the flow of ScanMatcher::matchScan, src/ScanMatcher.cpp:22-67)."""
    import torch
    capi, ctx = gpu
    m, sf, cfg = c1_world
    B = 24
    prm = capi.default_params(resolution=cfg["resolution"])
    gm = capi.Map(ctx, m, prm)
    om = oracle.Map(m, oracle.default_params(resolution=cfg["resolution"]))
    rng = np.random.default_rng(9)
    raws, lasts, prevs, curs = [], [], [], []
    for b in range(B):
        scan, truth, init = sf.make(b % 16)
        raws.append(np.repeat(scan, 2, axis=0) + rng.normal(0, 0.003, (2 * len(scan), 2)).astype(np.float32))
        # odometry: previous odometry pose arbitrary; the last estimate + increment lands near `init`
        last = np.array([init[0] - 0.3, init[1] + 0.1, np.rad2deg(init[2]) - 2.0])
        prev = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-180, 180)])
        a, al = np.deg2rad(prev[2]), np.deg2rad(last[2])
        d = np.array([[np.cos(al), np.sin(al)], [-np.sin(al), np.cos(al)]]) @ (np.array(init[:2]) - last[:2])   # motion in the robot frame
        cur = np.array([*(prev[:2] + np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]) @ d), prev[2] + 2.0])
        lasts.append(last); prevs.append(prev); curs.append(cur)
    lens = [len(r) for r in raws]
    raw_all = np.concatenate(raws).astype(np.float32); raw_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    last_cov = np.tile(np.eye(3) * 1e-4, (B, 1, 1))
    dev = "cuda:0"
    d_raw, d_roff = to_dev(raw_all), to_dev(raw_off)
    d_f = torch.empty_like(d_raw); d_foff = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    d_cur, d_prev, d_last, d_lc = to_dev(np.array(curs)), to_dev(np.array(prevs)), to_dev(np.array(lasts)), to_dev(last_cov.reshape(B, 9))
    d_mo = torch.zeros(B, 3, dtype=torch.float64, device=dev); d_pred = torch.zeros_like(d_mo); d_init = torch.zeros_like(d_mo)
    d_res = torch.zeros(B * capi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    d_fused = torch.zeros_like(d_mo); d_cov = torch.zeros(B, 9, dtype=torch.float64, device=dev); d_ok = torch.zeros(B, dtype=torch.int32, device=dev)
    fp = capi.default_fuse_params(score_thre=0.5)
    torch.cuda.synchronize()
    ctx.predict_batch_dev(d_cur.data_ptr(), d_prev.data_ptr(), d_last.data_ptr(), B, d_mo.data_ptr(), d_pred.data_ptr(), d_init.data_ptr())
    ctx.prefilter_batch_dev(d_raw.data_ptr(), 8, d_roff.data_ptr(), B, len(raw_all), 0.05, d_f.data_ptr(), d_foff.data_ptr())
    gm.align_batch_dev(d_f.data_ptr(), d_foff.data_ptr(), B, len(raw_all), d_init.data_ptr(), d_res.data_ptr())
    ctx.fuse_batch_dev(d_res.data_ptr(), d_pred.data_ptr(), d_mo.data_ptr(), d_last.data_ptr(), d_lc.data_ptr(), B, fp,
                       d_fused.data_ptr(), d_cov.data_ptr(), d_ok.data_ptr())
    torch.cuda.synchronize()
    fused, ok = d_fused.cpu().numpy(), d_ok.cpu().numpy()
    fo = oracle.default_fuse_params(score_thre=0.5)
    for b in range(B):
        mo_ref, pred_ref = oracle.predict(curs[b], prevs[b], lasts[b])
        filt = oracle.approx_voxel_filter(raws[b], 0.05)
        r_ref = om.align(filt, [pred_ref[0], pred_ref[1], np.deg2rad(pred_ref[2])])
        ok_ref, f_ref, _ = oracle.fuse(r_ref, pred_ref, mo_ref, lasts[b], last_cov[b], fo)
        assert ok[b] == ok_ref
        assert fused[b, :2] == pytest.approx(f_ref[:2], abs=1e-4) and abs(fused[b, 2] - f_ref[2]) < np.rad2deg(1e-4)
    assert ok.sum() >= B // 2
