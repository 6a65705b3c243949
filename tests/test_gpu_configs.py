"""GPU parity at the sizes of BASELINE.json configs[3] and configs[4] (SURVEY.md 8 shorthand C4, C5).

C4: a batch of 2048 scans x 10k points vs the shared 1M-point map -- one launch (workgroups take further scans from
the queue), against the oracle on EVERY scan and against eight 256-scan launches (what eight GPUs would run).
C5: multi-hypothesis relocalisation, 4096 seed poses x one 10k-point scan vs the 5M-point map, `shared_scan`
(one ordered copy of the scan per workgroup) -- against the oracle on every seed, against eight 512-seed shards
(the per-GPU share), arg-max equality; plus the HBM fall-back of the evaluation (scan larger than the LDS window,
poses that leave the staged voxels), reported in ndt_result.flags.
Tolerance: north_star's 1e-4 m / 1e-4 rad on poses; float32 / integer quantities exact (assert_result_parity)."""
import os

import numpy as np
import pytest

from test_gpu_parity import assert_result_parity, gpu  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

NTHREADS = min(16, os.cpu_count() or 1)


@pytest.fixture(scope="module")
def world_1m():
    from ndt_slam_amd import synth
    cfg = synth.CONFIGS["C4"]
    m = synth.make_map(cfg["n_map"], cfg["half"])
    return m, synth.ScanFactory(m, cfg["half"], cfg["n_scan"]), cfg


@pytest.fixture(scope="module")
def world_5m():
    from ndt_slam_amd import synth
    cfg = synth.CONFIGS["C5"]
    m = synth.make_map(cfg["n_map"], cfg["half"])
    return m, synth.ScanFactory(m, cfg["half"], cfg["n_scan"]), cfg


# Both settings of `stale_h_ang` at full size (the default, 0, is what PCL's computeDerivatives does --
# the angle terms of the Hessian are refreshed on every trial; 1 is kept as a switch).
STALE = [0, 1]


@pytest.mark.parametrize("preset", ["default", "pcl18", "pcl_new"])
@pytest.mark.parametrize("stale", STALE)
def test_c3_full_batch_every_scan_vs_oracle(gpu, oracle, world_1m, stale, preset):
    """configs[2]: all 256 scans of the bench batch against the oracle, under both settings of stale_h_ang and under
    every PCL-version preset (the presets change the voxel statistics and the float32 transform, i.e. every pass)."""
    capi, ctx = gpu
    m, sf, cfg = world_1m
    scans, off, truths, inits = sf.batch(0, 256)
    gm = capi.Map(ctx, m, capi.default_params(preset, resolution=cfg["resolution"], stale_h_ang=stale))
    om = oracle.Map(m, oracle.default_params(preset, resolution=cfg["resolution"], stale_h_ang=stale))
    res = gm.align_batch(scans, off, inits)
    ref = om.align_batch(scans, off, inits, nthreads=NTHREADS, run_stats=True)
    for b in range(256):
        assert_result_parity(res[b], ref[b])
    assert np.array_equal(res["iters"], ref["iters"]) and np.array_equal(res["ref_evals"], ref["ref_evals"])


@pytest.mark.parametrize("stale", STALE)
def test_c4_2048_scans_one_launch_vs_oracle_and_shards(gpu, oracle, world_1m, stale):
    capi, ctx = gpu
    m, sf, cfg = world_1m
    B = cfg["batch"]
    assert B == 2048
    scans, off, truths, inits = sf.batch(0, B)
    gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"], stale_h_ang=stale))
    res = gm.align_batch(scans, off, inits)                       # one launch, 2048 scans on 256 workgroups
    assert np.all(res["status"] == 0) and np.all(res["converged"] == 1)
    # configs[3] shards: eight launches of 256 scans give the same records, byte for byte
    for r in range(8):
        lo, hi = 256 * r, 256 * (r + 1)
        part = gm.align_batch(scans[int(off[lo]):int(off[hi])], off[lo:hi + 1] - off[lo], inits[lo:hi])
        assert part.tobytes() == res[lo:hi].tobytes(), "shard %d differs from the 2048-scan launch" % r
    # the oracle on every scan
    om = oracle.Map(m, oracle.default_params(resolution=cfg["resolution"], stale_h_ang=stale))
    ref = om.align_batch(scans, off, inits, nthreads=NTHREADS, run_stats=True)
    for b in range(B):
        assert_result_parity(res[b], ref[b])
    dp = np.abs(res["pose"] - ref["pose"])
    assert dp[:, :2].max() <= 1e-4 and dp[:, 2].max() <= 1e-4     # north_star tolerance, whole batch
    assert np.array_equal(res["iters"], ref["iters"])


@pytest.mark.parametrize("stale", STALE)
def test_c5_4096_seeds_shared_scan_vs_oracle_and_shards(gpu, oracle, world_5m, stale):
    capi, ctx = gpu
    from ndt_slam_amd import synth
    m, sf, cfg = world_5m
    scan, truth, _ = sf.make(0)
    seeds = synth.hypothesis_seeds(truth, cfg["seeds"])
    assert len(seeds) == 4096
    prm = capi.default_params(resolution=cfg["resolution"], stale_h_ang=stale)
    gm = capi.Map(ctx, m, prm)
    om = oracle.Map(m, oracle.default_params(resolution=cfg["resolution"], stale_h_ang=stale))
    # the 5M-point voxel grid itself
    gi, oi = gm.info(), om.info()
    assert (gi.min_bx, gi.min_by, gi.div_x, gi.div_y, gi.n_cells, gi.n_valid) == \
           (oi.min_bx, oi.min_by, oi.div_x, oi.div_y, oi.n_cells, oi.n_valid)
    g, o = gm.export(), om.export()
    for k in ("idx", "npts", "cent", "mean"):
        assert np.array_equal(g[k], o[k]), k
    assert g["icov"] == pytest.approx(o["icov"], rel=1e-12, abs=1e-300)
    off = np.array([0, len(scan)], np.uint64)
    res = gm.align_batch(scan, off, seeds, shared_scan=True)      # 4096 matches of the one scan, one launch
    assert np.all(res["status"] == 0)
    # per-GPU share of configs[4]: 512 seeds per launch
    for r in range(8):
        part = gm.align_batch(scan, off, seeds[512 * r:512 * (r + 1)], shared_scan=True)
        assert part.tobytes() == res[512 * r:512 * (r + 1)].tobytes(), "seed shard %d" % r
    ref = om.align_batch(scan, off, seeds, shared_scan=True, nthreads=NTHREADS, run_stats=True)
    for b in range(len(seeds)):
        assert_result_parity(res[b], ref[b])
    # relocalisation: the best hypothesis is the same one and it is the true pose
    best, best_ref = int(np.argmax(res["trans_prob"])), int(np.argmax(ref["trans_prob"]))
    assert best == best_ref
    assert np.hypot(*(res["pose"][best][:2] - truth[:2])) < 0.02      # within the line search's dither of the truth
    # the flags say which data path was taken; nothing else may depend on it
    spilled = (res["flags"] & capi.FLAG_WINDOW_SPILL) != 0
    print("C5: %d of 4096 windows with voxels left in HBM, evals mean %.1f max %d" %
          (spilled.sum(), res["evals"].mean(), res["evals"].max()))


def test_hbm_fallback_paths_match_oracle(gpu, oracle, world_5m):
    """The evaluation reads voxels from the LDS window; a point whose 3x3 neighbourhood leaves the window, or
    touches an occupied voxel that got no LDS record, reads the dense cell table in HBM instead
    (ndt_point.hip.h, slow path).  Both cases on the 5M-point map."""
    capi, ctx = gpu
    from ndt_slam_amd import synth
    m, _, cfg = world_5m
    gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
    om = oracle.Map(m, oracle.default_params(resolution=cfg["resolution"]))
    # (1) a scan whose voxel bounding box (2 x 48 m at 0.5 m = 192 cells + margin) exceeds the 16384-cell window
    wide = synth.ScanFactory(m, cfg["half"], cfg["n_scan"], radius=48.0)
    for k in (0, 1, 2):
        scan, truth, init = wide.make(k)
        r = gm.align(scan, init)
        assert int(r["flags"]) & capi.FLAG_REGION_CLIPPED, "scan %d was expected not to fit the window" % k
        assert_result_parity(r, om.align(scan, init, run_stats=True))
    # (2) poses that start 2-3 m away and walk: the window is staged around the first pose (+ 2 cells around
    # every cell a point fell in), later poses reach voxels outside that set
    near = synth.ScanFactory(m, cfg["half"], cfg["n_scan"], radius=12.0)
    n_spill = 0
    for k in range(6):
        scan, truth, init = near.make(k)
        far = truth + np.array([2.4 * np.cos(k), 2.4 * np.sin(k), np.radians(4.0 * (k - 2.5))])
        r = gm.align(scan, far)
        n_spill += int((int(r["flags"]) & capi.FLAG_WINDOW_SPILL) != 0)
        assert_result_parity(r, om.align(scan, far, run_stats=True))
    assert n_spill > 0, "no window left voxels in HBM: the test does not reach the fall-back"


def test_scan_above_the_lds_sort_limit(gpu, oracle, world_1m):
    """Scans above 20000 points are not re-ordered (NDT_FLAG_UNSORTED) and above the staging limits read from HBM;
    30k and 70k points against the oracle.  Scans of 10241 .. 20000 points are ordered by the streaming routines
    (compute_region + sort_points) instead of the register-resident set-up of the 10k-point scans: 15k points."""
    capi, ctx = gpu
    from ndt_slam_amd import synth
    m, _, cfg = world_1m
    gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
    om = oracle.Map(m, oracle.default_params(resolution=cfg["resolution"]))
    for n in (30_000, 70_000):
        sf = synth.ScanFactory(m, cfg["half"], n, radius=40.0)
        scan, truth, init = sf.make(3)
        r = gm.align(scan, init)
        assert int(r["flags"]) & capi.FLAG_UNSORTED
        assert_result_parity(r, om.align(scan, init, run_stats=True))
    sf = synth.ScanFactory(m, cfg["half"], 15_000, radius=35.0)
    for k in (1, 2):
        scan, truth, init = sf.make(k)
        r = gm.align(scan, init)
        assert not (int(r["flags"]) & capi.FLAG_UNSORTED)
        assert_result_parity(r, om.align(scan, init, run_stats=True))
    # a mixed batch: both set-up routines and the unordered path side by side in one launch
    parts = [synth.ScanFactory(m, cfg["half"], n, radius=40.0).make(5)[:3:2] for n in (9_000, 15_000, 10_240, 10_241, 25_000)]
    scans = np.concatenate([p[0] for p in parts]); inits = np.stack([p[1] for p in parts])
    off = np.concatenate([[0], np.cumsum([len(p[0]) for p in parts])]).astype(np.uint64)
    res = gm.align_batch(scans, off, inits)
    for b in range(len(parts)):
        assert_result_parity(res[b], om.align(parts[b][0], parts[b][1], run_stats=True))
