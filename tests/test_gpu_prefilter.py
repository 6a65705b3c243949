"""GPU parity tests of the source pre-filter: the device replay of
pcl::ApproximateVoxelGrid against the CPU oracle -- float32 centroids and output ORDER must match
bit for bit (the filter is order dependent; src/PoseEstimator.cpp:6-10)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from ndt_slam_amd import capi
    return capi, capi.Context(0)


def lidar_like(rng, n, radius=30.0):
    """Points along a few walls in angular order: long runs inside one voxel, then jumps."""
    ang = np.sort(rng.uniform(-np.pi, np.pi, n))
    r = radius * (0.6 + 0.4 * np.abs(np.sin(3 * ang))) + rng.normal(0, 0.01, n)
    return np.stack([r * np.cos(ang), r * np.sin(ang)], 1).astype(np.float32)


@pytest.mark.parametrize("leaf", [0.05, 0.1, 0.5])
def test_prefilter_matches_oracle_bit_for_bit(gpu, oracle, leaf):
    capi, ctx = gpu
    rng = np.random.default_rng(7)
    for n in (1, 2, 63, 64, 65, 360, 5000, 8192, 8193, 20000, 50000, 65535, 65536):    # up to 65535: by slot (tiles of 8192); above: step by step
        for cloud in (lidar_like(rng, n), rng.uniform(-40, 40, (n, 2)).astype(np.float32)):
            ref = oracle.approx_voxel_filter(cloud, leaf)
            got = ctx.prefilter(cloud, leaf)
            assert got.shape == ref.shape
            assert got.tobytes() == ref.tobytes()


def test_prefilter_scan_longer_than_the_flush_bitmap(gpu, oracle):
    """More than 2^18 points in one scan: the kernel's single-wave path."""
    capi, ctx = gpu
    rng = np.random.default_rng(17)
    cloud = lidar_like(rng, 300_000, radius=60.0)
    assert ctx.prefilter(cloud, 0.05).tobytes() == oracle.approx_voxel_filter(cloud, 0.05).tobytes()


def test_prefilter_batch_of_mixed_lengths(gpu, oracle):
    """One batch whose scans go down both kernels (by slot up to 65535 points, step by step above), empty scan included."""
    import torch
    capi, ctx = gpu
    rng = np.random.default_rng(23)
    lens = [100, 65535, 0, 70000, 30000, 1, 65536, 8192]
    clouds = [lidar_like(rng, n) if n else np.zeros((0, 2), np.float32) for n in lens]
    raw = np.concatenate(clouds).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dev = torch.device("cuda", 0)
    d_raw = torch.from_numpy(raw).to(dev); d_off = torch.from_numpy(off).to(dev)
    d_out = torch.empty_like(d_raw); d_ooff = torch.zeros(len(lens) + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.prefilter_batch_dev(d_raw.data_ptr(), 8, d_off.data_ptr(), len(lens), len(raw), 0.05, d_out.data_ptr(), d_ooff.data_ptr())
    torch.cuda.synchronize()
    ooff = d_ooff.cpu().numpy(); out = d_out.cpu().numpy()
    for b, c in enumerate(clouds):
        ref = oracle.approx_voxel_filter(c, 0.05) if len(c) else np.zeros((0, 2), np.float32)
        got = out[int(ooff[b]):int(ooff[b + 1])]
        assert got.shape == ref.shape and got.tobytes() == ref.tobytes(), b


def test_prefilter_edge_cases(gpu, oracle):
    capi, ctx = gpu
    one_voxel = np.full((300, 2), 0.012, np.float32) + np.linspace(0, 1e-3, 300, dtype=np.float32)[:, None]
    assert ctx.prefilter(one_voxel, 0.05).tobytes() == oracle.approx_voxel_filter(one_voxel, 0.05).tobytes()
    # two voxels that share a hash slot, alternating: every point flushes the other one
    inv = np.float32(1.0) / np.float32(0.1)
    a = np.array([0.05, 0.05], np.float32)
    k = next(k for k in range(1, 4000) if (k * 7171) % 512 == 0)
    bpt = np.array([0.05 + 0.1 * k, 0.05], np.float32)
    assert ((int(np.floor(bpt[0] * inv)) * 7171) & 511) == ((int(np.floor(a[0] * inv)) * 7171) & 511)
    alt = np.stack([a if i % 2 == 0 else bpt for i in range(257)]).astype(np.float32)
    got, ref = ctx.prefilter(alt, 0.1), oracle.approx_voxel_filter(alt, 0.1)
    assert len(ref) == 257 and got.tobytes() == ref.tobytes()
    neg = np.array([[-0.01, -0.01], [-0.06, 0.3], [-1e3, 1e3], [0.0, -0.0]], np.float32)
    assert ctx.prefilter(neg, 0.05).tobytes() == oracle.approx_voxel_filter(neg, 0.05).tobytes()
    with pytest.raises(capi.NdtError):
        ctx.prefilter(np.zeros((0, 2), np.float32), 0.05)
    with pytest.raises(capi.NdtError):
        ctx.prefilter(np.zeros((4, 2), np.float32), 0.0)


def test_prefilter_batch_feeds_the_matcher(gpu, oracle):
    """Raw scans in HBM -> filter -> matches, all on the device, against filter + match on the CPU."""
    import torch
    capi, ctx = gpu
    from ndt_slam_amd import synth
    cfg = synth.CONFIGS["C1"]
    m = synth.make_map(cfg["n_map"], cfg["half"])
    sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
    B = 37
    raws, inits = [], []
    rng = np.random.default_rng(3)
    for b in range(B):
        scan, truth, init = sf.make(b % 16)
        raw = np.repeat(scan, 3, axis=0) + rng.normal(0, 0.004, (3 * len(scan), 2)).astype(np.float32)   # oversampled
        raws.append(raw[: 1 + (b * 131) % len(raw)] if b % 4 == 1 else raw); inits.append(init)
    lens = [len(r) for r in raws]
    raw_all = np.concatenate(raws).astype(np.float32)
    raw_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dev = torch.device("cuda", 0)
    d_raw = torch.from_numpy(raw_all).to(dev); d_roff = torch.from_numpy(raw_off).to(dev)
    d_out = torch.empty_like(d_raw); d_ooff = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    d_init = torch.from_numpy(np.array(inits)).to(dev)
    d_res = torch.zeros(B * capi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    leaf = 0.05
    prm = capi.default_params(resolution=cfg["resolution"])
    gm = capi.Map(ctx, m, prm)
    ctx.prefilter_batch_dev(d_raw.data_ptr(), 8, d_roff.data_ptr(), B, len(raw_all), leaf, d_out.data_ptr(), d_ooff.data_ptr())
    gm.align_batch_dev(d_out.data_ptr(), d_ooff.data_ptr(), B, len(raw_all), d_init.data_ptr(), d_res.data_ptr())
    torch.cuda.synchronize()                               # device-wide: includes the context's own stream
    ooff = d_ooff.cpu().numpy(); out = d_out.cpu().numpy()
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)
    om = oracle.Map(m, oracle.default_params(resolution=cfg["resolution"]))
    for b in range(B):
        ref_f = oracle.approx_voxel_filter(raws[b], leaf)
        got_f = out[ooff[b]:ooff[b + 1]]
        assert got_f.tobytes() == ref_f.tobytes()
        ref = om.align(ref_f, inits[b])
        assert int(res[b]["status"]) == 0 and int(res[b]["iters"]) == int(ref["iters"])
        assert (res[b]["T00"], res[b]["T10"], res[b]["T03"], res[b]["T13"]) == (ref["T00"], ref["T10"], ref["T03"], ref["T13"])
