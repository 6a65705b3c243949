"""The HIP path against the reference's own vendored Eigen (tests/golden/eigen_golden.npz, made in the build container by
tests/golden/make_eigen_golden.{cpp,py}): `map_finalize_kernel`'s Sigma^-1 against VoxelGridCovariance's leaf block run
on Eigen, and the device's Newton step against matches the oracle replayed with JacobiSVD's delta_p in every step."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eigen_golden.npz")


@pytest.fixture(scope="module")
def world():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from ndt_slam_amd import capi
    z = np.load(GOLD)
    ctx = capi.Context(0)
    gm = capi.Map(ctx, z["replay_map"], capi.default_params(resolution=float(z["replay_resolution"])))
    return z, capi, ctx, gm


def test_device_cell_table_against_eigens_leaf_block(world):
    """include/Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h:405-553 + LU/InverseImpl.h:140-200 behind src/PoseEstimator.cpp:19."""
    z, capi, ctx, gm = world
    g = gm.export()
    assert np.array_equal(g["npts"], z["replay_cells_npts_eig"])           # same accept / reject decision per voxel
    assert g["mean"].tobytes() == z["replay_cells_mean_eig"].tobytes()     # Eigen's mean_ / n, bit for bit
    ok = g["npts"] > 0
    e = z["replay_cells_icov_eig"][ok]
    rel = np.abs(g["icov"][ok] - e) / np.abs(e).max(axis=1, keepdims=True)
    assert rel.max() < 2e-10 and np.median(rel) < 1e-15, (rel.max(), np.median(rel))


def test_device_newton_steps_against_matches_replayed_with_jacobisvd(world):
    """include/Eigen/src/SVD/JacobiSVD.h:488,664 behind src/PoseEstimator.cpp:28: the device's 3x3 solve drives the same
    sequence of step lengths, the same float32 transforms and iteration counts as the replay in which every delta_p came
    from JacobiSVD<Matrix<double,6,6>>::solve (and as the replay on Eigen's cell table)."""
    z, capi, ctx, gm = world
    scans, off, inits = z["replay_scans"], z["replay_offsets"], z["replay_inits"]
    res, traces = gm.align_batch(scans, off, inits, trace_cap=512)
    for name in ("solve", "cells"):
        rr, tt = z["replay_%s_results" % name], z["replay_%s_trace" % name]
        for b in range(len(inits)):
            assert int(res[b]["status"]) == 0
            assert int(res[b]["iters"]) == int(rr[b]["iters"]) and int(res[b]["converged"]) == int(rr[b]["converged"])
            for k in ("T00", "T10", "T03", "T13"):
                assert res[b][k] == rr[b][k], (name, b, k)
            rows = tt[b][~np.isnan(tt[b][:, 0])]
            assert len(traces[b]) == len(rows), (name, b)
            assert traces[b][:, 0] == pytest.approx(rows[:, 0], rel=1e-8, abs=1e-12)        # step lengths
            assert traces[b][:, 5:8] == pytest.approx(rows[:, 5:8], rel=0, abs=1e-9)        # trial poses = p + a * delta_p / |delta_p|
