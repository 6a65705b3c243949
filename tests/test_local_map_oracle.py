"""the oracle's literal two-buffer octree (oracle/ndt_oracle_octree.c) against an
independent numpy statement of the voxel lattice, and Submap::makeMap's bookkeeping (CPU only)."""
import numpy as np
import pytest

from oracle import ndt_oracle as O


def lattice_difference(base, test, r):
    """Indices of test points whose lattice cell (anchored at the first point added, minus r) is not occupied
    by base.  Valid away from exact cell boundaries, which random float data never hits."""
    first = base[0] if len(base) else test[0]
    a = first.astype(np.float64) - r
    kb = np.floor((base.astype(np.float64) - a) / r).astype(np.int64) if len(base) else np.zeros((0, 2), np.int64)
    kt = np.floor((test.astype(np.float64) - a) / r).astype(np.int64)
    occ = set(map(tuple, kb))
    return np.array([i for i, k in enumerate(map(tuple, kt)) if k not in occ], dtype=np.int64)


@pytest.mark.parametrize("seed", range(6))
def test_octree_matches_lattice(seed):
    rng = np.random.default_rng(seed)
    nb, nt = int(rng.integers(0, 3000)), int(rng.integers(1, 2000))
    base = (rng.normal(size=(nb, 2)) * 5 + rng.uniform(-50, 50, 2)).astype(np.float32)
    near = (base[rng.integers(0, nb, nt // 2)] + rng.normal(size=(nt // 2, 2)) * 0.02) if nb else np.zeros((0, 2))
    test = np.concatenate([near, rng.normal(size=(nt - len(near), 2)) * 6]).astype(np.float32)
    idx = O.difference_indices(base, test, 0.05)
    assert np.array_equal(np.sort(idx), lattice_difference(base, test, 0.05))
    assert len(np.unique(idx)) == len(idx)


def test_octree_order_is_depth_first():
    """The detector walks children in (x, y, z) bit order.  The first test point lies beyond the box in +x and
    +y at every doubling, so the lattice origin stays one cell below the first base point and the new leaves
    must come back sorted by the Morton code of their cell (x the high bit).  (A point that violates only one
    axis makes the box grow DOWNWARDS along the other: adoptBoundingBoxToPoint's child index is built from the
    upper-bound flags alone.)"""
    r = 0.5
    base = np.array([[0.0, 0.0]], np.float32)
    cells = [(6, 6), (3, 1), (1, 3), (2, 2), (1, 1), (3, 3), (2, 0)]
    test = np.array([[cx * r + 0.1, cy * r + 0.1] for cx, cy in cells], np.float32)
    idx = O.difference_indices(base, test, r)

    def morton(c):            # cells are relative to the first point; the lattice origin is one cell below it
        x, y = c[0] + 1, c[1] + 1
        return sum((((x >> b) & 1) << (2 * b + 1)) | (((y >> b) & 1) << (2 * b)) for b in range(8))
    assert [cells[i] for i in idx] == sorted(cells, key=morton)


def test_non_finite_and_empty():
    base = np.array([[0, 0], [np.nan, 1], [1, 1]], np.float32)
    test = np.array([[np.inf, 0], [0.01, 0.01], [5, 5], [np.nan, np.nan]], np.float32)
    idx = O.difference_indices(base, test, 0.05)
    assert list(idx) == [2]
    assert len(O.difference_indices(np.zeros((0, 2), np.float32), test, 0.05)) == 2      # every finite point is new
    assert len(O.difference_indices(base, np.zeros((0, 2), np.float32), 0.05)) == 0


def test_span_limit():
    base = np.array([[0, 0]], np.float32)
    test = np.array([[1e9, 0]], np.float32)
    with pytest.raises(ValueError):
        O.difference_indices(base, test, 0.05)


def test_make_map_bookkeeping():
    rng = np.random.default_rng(3)
    wall = np.stack([np.linspace(-5, 5, 400), np.full(400, 2.01)], 1)      # seen identically by every scan
    scans = []
    for k in range(5):
        mover = np.stack([np.linspace(-0.2, 0.2, 30) + k * 0.8 - 2, np.full(30, 1.0)], 1)    # an object walking by
        mover = mover + rng.normal(size=(30, 2)) * 0.002
        scans.append(np.concatenate([wall, mover]).astype(np.float32))
    # without removal: all scans (first submap) or scans 2.. (later submaps)
    m = O.make_map(scans, True, True, False, 0.05, 0.1)
    assert np.array_equal(m, np.concatenate(scans))
    m = O.make_map(scans, False, True, False, 0.05, 0.1)
    assert np.array_equal(m, np.concatenate(scans[2:]))
    # with removal: scans[0] + filtered middles + newest
    m = O.make_map(scans, True, True, True, 0.05, 0.1)
    mids = []
    for i in range(3):
        d = O.difference_extraction(np.concatenate([scans[i], scans[i + 2]]), scans[i + 1], 0.05)
        mids.append(O.remove_neighbors(scans[i + 1], d, 0.1))
    assert np.array_equal(m, np.concatenate([scans[0]] + mids + [scans[4]]))
    # the walking object is gone from the middle scans, the wall mostly stays
    for k, mid in enumerate(mids):
        mover_x = (k + 1) * 0.8 - 2
        assert not np.any((np.abs(mid[:, 0] - mover_x) < 0.15) & (np.abs(mid[:, 1] - 1.0) < 0.05))
        assert 390 <= len(mid) <= 400          # (the first wall point sits on a lattice corner and may flip voxel)
    assert len(O.make_map(scans[:1], True, True, True, 0.05, 0.1)) == 2 * len(scans[0])      # as the reference does
    assert len(O.make_map(scans[:2], False, False, True, 0.05, 0.1)) == 0
