"""SURVEY.md 8f row f3 on the device: PCFilter::difference_extraction (include/ndt_slam/PCFilter.h:58-94) and
Submap::makeMap (src/PointCloudMap.cpp:15-39) through the C ABI against the oracle's literal two-buffer octree
(oracle/ndt_oracle_octree.c).  Exact: same point set for the difference (the device returns it in input order),
same bytes for the assembled submap cloud."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    from ndt_slam_amd import capi
    return capi.Context(0)


def scene(rng, nb, nt, centre=(0.0, 0.0), spread=6.0):
    """A base cloud and a test cloud that partly repeats it (same surfaces seen again) and partly does not."""
    c = np.asarray(centre)
    base = (rng.normal(size=(nb, 2)) * spread + c).astype(np.float32)
    if nb:
        again = base[rng.integers(0, nb, nt // 2)] + rng.normal(size=(nt // 2, 2)) * 0.02
    else:
        again = np.zeros((0, 2))
    fresh = rng.normal(size=(nt - len(again), 2)) * spread * 1.3 + c
    return base, np.concatenate([again, fresh]).astype(np.float32)


def expect_difference(oracle, base, test, r):
    return test[np.sort(oracle.difference_indices(base, test, r))]


@pytest.mark.parametrize("nb,nt,centre,r", [
    (1, 1, (0, 0), 0.05), (0, 300, (3, -2), 0.05), (700, 64, (0, 0), 0.05), (1023, 1025, (40, -70), 0.05),
    (5000, 4000, (-800.5, 1200.25), 0.05), (20000, 20000, (0, 0), 0.05), (3000, 3000, (0, 0), 0.5),
    (3000, 2000, (10, 10), 0.013)])
def test_difference_extraction_matches_the_octree(ctx, oracle, nb, nt, centre, r):
    rng = np.random.default_rng(nb * 7 + nt)
    base, test = scene(rng, nb, nt, centre)
    got = ctx.difference_extraction(base, test, r)
    ref = expect_difference(oracle, base, test, r)
    assert got.shape == ref.shape and got.tobytes() == ref.tobytes()
    assert 0 < len(got) <= nt


@pytest.mark.parametrize("resol,span", [(0.03, 5.0), (0.3, 20.0), (0.02, 40.0), (0.07, 9.0), (0.03, 60.0), (0.3, 500.0)])
def test_leaf_sizes_whose_z_key_depends_on_the_frame(ctx, oracle, resol, span):
    """The octree's z key of a z = 0 point differs between frames of the growing box for these leaf sizes (the
    `resol` ROS parameter, include/ndt_slam/PCFilter.h:21-23): columns split into two leaves, the difference set
    grows.  Difference extraction and the whole assembly against the pointer octree."""
    rng = np.random.default_rng(int(resol * 1000) + int(span))
    base = (rng.uniform(-span, span, size=(6000, 2))).astype(np.float32)
    near = base[rng.integers(0, len(base), 2500)] + (rng.normal(size=(2500, 2)) * resol * 0.2).astype(np.float32)
    test = np.concatenate([near, rng.uniform(-1.5 * span, 1.5 * span, size=(800, 2)).astype(np.float32)])
    got = ctx.difference_extraction(base, test, resol)
    assert got.tobytes() == expect_difference(oracle, base, test, resol).tobytes()
    scans = [base[:3000], test, base[3000:], near[:1500], base[1000:4000]]
    for first, newest in ((True, True), (False, False)):
        assert ctx.make_map(scans, first, newest, True, resol, 2.0 * resol).tobytes() == \
               oracle.make_map(scans, first, newest, True, resol, 2.0 * resol).tobytes()


def test_points_on_the_voxel_lattice(ctx, oracle):
    """Coordinates that are exact multiples of the voxel size away from the first point sit on cell borders,
    where the fp64 key of PCL's genOctreeKeyforPoint decides; the box also grows several times in between."""
    rng = np.random.default_rng(11)
    r = 0.05
    k = rng.integers(-400, 400, (4000, 2))
    base = (np.float32(1.25) + k[:2500].astype(np.float32) * np.float32(r)).astype(np.float32)
    test = (np.float32(1.25) + k[1500:].astype(np.float32) * np.float32(r)).astype(np.float32)
    got = ctx.difference_extraction(base, test, r)
    ref = expect_difference(oracle, base, test, r)
    assert got.tobytes() == ref.tobytes() and len(got) > 0


def test_growth_in_every_direction(ctx, oracle):
    """The first point is in the middle; later points walk outwards so the box doubles up, down, left and right
    while both clouds are being added (keys of early points are moved by the later shifts)."""
    t = np.linspace(0, 40, 3000)
    spiral = np.stack([t * np.cos(t), t * np.sin(t)], 1).astype(np.float32)
    base, test = spiral[::2], (spiral[1::2] + np.float32(0.004)).astype(np.float32)
    for r in (0.05, 0.2):
        got = ctx.difference_extraction(base, test, r)
        assert got.tobytes() == expect_difference(oracle, base, test, r).tobytes()


def test_non_finite_points_and_span_limit(ctx, oracle):
    base = np.array([[0, 0], [np.nan, 1], [1, 1]], np.float32)
    test = np.array([[np.inf, 0], [0.01, 0.01], [5, 5], [np.nan, np.nan]], np.float32)
    assert ctx.difference_extraction(base, test, 0.05).tolist() == [[5.0, 5.0]]
    assert len(ctx.difference_extraction(np.zeros((0, 2), np.float32), test, 0.05)) == 2
    assert len(ctx.difference_extraction(base, np.zeros((0, 2), np.float32), 0.05)) == 0
    with pytest.raises(RuntimeError):
        ctx.difference_extraction(base, np.array([[1e9, 0]], np.float32), 0.05)
    with pytest.raises(RuntimeError):
        ctx.difference_extraction(base, test, 0.0)


def submap_scans(rng, n_scans, n_wall, n_mover, jitter=0.003):
    """Scans of one submap in the map frame: static walls re-observed with noise + an object that moves."""
    th = np.linspace(0, 2 * np.pi, n_wall, endpoint=False)
    room = np.stack([8 * np.cos(th) / np.maximum(abs(np.cos(th)), abs(np.sin(th))),
                     6 * np.sin(th) / np.maximum(abs(np.cos(th)), abs(np.sin(th)))], 1)
    scans = []
    for k in range(n_scans):
        m = int(n_mover * (0.5 + rng.random()))
        mover = np.stack([rng.normal(-3 + 0.6 * k, 0.1, m), rng.normal(0.5, 0.15, m)], 1)
        keep = rng.random(n_wall) > 0.1                       # ragged: not every beam returns
        scans.append((np.concatenate([room[keep], mover]) + rng.normal(size=(keep.sum() + m, 2)) * jitter)
                     .astype(np.float32))
    return scans


@pytest.mark.parametrize("n_scans,n_wall,n_mover", [(3, 500, 20), (5, 1500, 60), (12, 4000, 150), (30, 700, 30)])
def test_make_map_matches_the_reference_assembly(ctx, oracle, n_scans, n_wall, n_mover):
    rng = np.random.default_rng(n_scans)
    scans = submap_scans(rng, n_scans, n_wall, n_mover)
    for first, newest in ((True, True), (False, True), (False, False)):
        got = ctx.make_map(scans, first, newest, True, 0.05, 0.1)
        ref = oracle.make_map(scans, first, newest, True, 0.05, 0.1)
        assert got.shape == ref.shape and got.tobytes() == ref.tobytes()
    total = sum(len(s) for s in scans)
    assert 0 < len(ref) < total                               # something was removed, something stayed
    for first in (True, False):
        got = ctx.make_map(scans, first, True, False, 0.05, 0.1)
        assert got.tobytes() == oracle.make_map(scans, first, True, False, 0.05, 0.1).tobytes()


def test_make_map_short_and_ragged_submaps(ctx, oracle):
    rng = np.random.default_rng(5)
    scans = submap_scans(rng, 4, 300, 15)
    empty = np.zeros((0, 2), np.float32)
    cases = [scans[:1], scans[:2], scans[:3], [scans[0], empty, scans[2], scans[3]], [empty, scans[1], empty],
             [scans[0], scans[1], empty], [empty, empty, scans[2]]]
    for sc in cases:
        for first, newest in ((True, True), (False, False), (True, False)):
            got = ctx.make_map(sc, first, newest, True, 0.05, 0.1)
            ref = oracle.make_map(sc, first, newest, True, 0.05, 0.1)
            assert got.shape == ref.shape and got.tobytes() == ref.tobytes(), (len(sc), first, newest)


def test_make_map_feeds_the_target_on_the_device(ctx, oracle):
    """makeMap -> filterPoints (ApproximateVoxelGrid) -> NDT target, device to device
    (src/PointCloudMap.cpp:119-134, src/PoseEstimator.cpp:22-28), against the same chain on the oracle."""
    import torch
    from ndt_slam_amd import capi
    rng = np.random.default_rng(9)
    scans = submap_scans(rng, 8, 3000, 100)
    off = np.zeros(len(scans) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in scans])
    allp = torch.from_numpy(np.concatenate(scans)).cuda()
    out = torch.empty((len(allp) + 1, 2), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()            # (a null stream handle would mean "the context's own stream")
    ctx.make_map_dev(allp.data_ptr(), 8, off, True, True, True, 0.05, 0.1, out.data_ptr(), cnt.data_ptr(),
                     stream=stream.cuda_stream)
    stream.synchronize()
    n = int(cnt.item())
    ref = oracle.make_map(scans, True, True, True, 0.05, 0.1)
    assert n == len(ref) and out[:n].cpu().numpy().tobytes() == ref.tobytes()
    filt_gpu = ctx.prefilter(out[:n].cpu().numpy(), 0.05)
    filt_ref = oracle.approx_voxel_filter(ref, 0.05)
    assert filt_gpu.tobytes() == filt_ref.tobytes()
    gm = capi.Map(ctx, filt_gpu, capi.default_params(resolution=0.5))
    assert gm.info().n_valid > 20
    gm.close()


def test_wide_spans_and_degenerate_clouds(ctx, oracle):
    """Tree depths far beyond a room (a point 3e6 m away: 26 levels), every point in one voxel (all inserts hit
    one set entry), exact duplicates, and a base cloud much larger than the test cloud."""
    rng = np.random.default_rng(17)
    base, test = scene(rng, 4000, 3000)
    far = np.array([[3.0e6, -2.5e6]], np.float32)
    for b, t in ((np.concatenate([base, far]), test), (base, np.concatenate([test[:100], far, test[100:]])),
                 (np.concatenate([far, base]), test)):
        got = ctx.difference_extraction(b, t, 0.05)
        assert got.tobytes() == expect_difference(oracle, b, t, 0.05).tobytes()
    one = (np.float32(2.0) + rng.random((5000, 2)).astype(np.float32) * np.float32(0.01)).astype(np.float32)
    got = ctx.difference_extraction(one[:3000], one[3000:], 0.05)
    assert got.tobytes() == expect_difference(oracle, one[:3000], one[3000:], 0.05).tobytes()
    dup = np.repeat(base[:50], 40, axis=0)
    got = ctx.difference_extraction(dup, np.concatenate([dup[::7], test[:200]]), 0.05)
    assert got.tobytes() == expect_difference(oracle, dup, np.concatenate([dup[::7], test[:200]]), 0.05).tobytes()
    big, small = scene(rng, 150000, 2000, spread=40.0)
    got = ctx.difference_extraction(big, small, 0.05)
    assert got.tobytes() == expect_difference(oracle, big, small, 0.05).tobytes()


def test_make_map_is_repeatable_and_reuses_its_buffers(ctx, oracle):
    """Calls of different sizes back to back on one context (scratch grows and is reused, the pinned job table
    of the previous call is waited for) give the same clouds as fresh calls."""
    rng = np.random.default_rng(23)
    sets = [submap_scans(rng, n, w, m) for n, w, m in ((4, 300, 10), (9, 2500, 80), (3, 100, 5), (14, 900, 30))]
    refs = [oracle.make_map(s, True, True, True, 0.05, 0.2) for s in sets]
    for _ in range(2):
        for s, r in zip(sets, refs):
            assert ctx.make_map(s, True, True, True, 0.05, 0.2).tobytes() == r.tobytes()
