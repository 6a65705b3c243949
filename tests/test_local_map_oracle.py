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


def replay_difference(base, test, r):
    """The octree's voxel lattice by replaying its bounding-box growth point by point (a second, loop-level
    statement of adoptBoundingBoxToPoint + genOctreeKeyforPoint in three dimensions, z = 0): every point's key is
    taken in the frame of the moment it is inserted and moved to the final frame by the shifts of the later
    doublings.  Returns the indices of the test points whose (x, y, z) key no base point has."""
    eps = float(np.finfo(np.float32).eps)
    mn = mx = None
    depth = 0
    shift = np.zeros(3, np.int64)
    recs = []                                  # (key in its frame, shift at insertion)
    for p in list(base) + list(test):
        p3 = np.array([float(p[0]), float(p[1]), 0.0])
        if not np.all(np.isfinite(p3)):
            recs.append(None)
            continue
        while True:
            if mn is None:
                mn, mx = p3 - r / 2, p3 + r / 2
                side = 2.0 * r
                for a in range(3):
                    over = (side - (mx[a] - mn[a])) / 2.0
                    if over > eps:
                        mn[a] -= over; mx[a] += over
                depth = 1
            up = p3 >= mx
            if not (np.any(p3 < mn) or np.any(up)):
                break
            side = float(1 << depth) * r
            for a in range(3):
                if not up[a]:
                    mn[a] -= side; shift[a] += 1 << depth
            depth += 1
            mx = mn + (float(1 << depth) * r - eps)
        recs.append((((p3 - mn) / r).astype(np.int64), shift.copy()))
    keys = [None if q is None else tuple(q[0] + (shift - q[1])) for q in recs]
    occ = set(k for k in keys[:len(base)] if k is not None)
    return np.array([i for i, k in enumerate(keys[len(base):]) if k is not None and k not in occ], dtype=np.int64)


@pytest.mark.parametrize("resol,span", [(0.05, 5.0), (0.03, 5.0), (0.3, 20.0), (0.02, 40.0), (0.07, 9.0), (0.013, 3.0)])
def test_octree_matches_the_frame_replay_in_three_dimensions(resol, span):
    """For most leaf sizes the z key of a z = 0 point is not the same in every frame of the growing box (min_z
    accumulates rounding, the fp64 quotient truncates to 2^depth - 2 instead of 2^depth - 1), so points of one
    (x, y) column inserted at different depths can fall into different leaves -- at 0.03 / 0.3 / 0.02 / 0.07 a
    lattice in x and y alone disagrees with the octree, the three-dimensional replay does not."""
    rng = np.random.default_rng(int(resol * 1000))
    base = (rng.uniform(-span, span, size=(1500, 2))).astype(np.float32)
    near = base[rng.integers(0, len(base), 600)] + (rng.normal(size=(600, 2)) * resol * 0.2).astype(np.float32)
    test = np.concatenate([near, rng.uniform(-1.5 * span, 1.5 * span, size=(300, 2)).astype(np.float32)])
    idx = np.sort(O.difference_indices(base, test, resol))
    assert np.array_equal(idx, replay_difference(base, test, resol))


def test_z_keys_split_columns_at_some_leaf_sizes():
    """The phenomenon itself: at 0.03 the octree reports more new-voxel points than an (x, y) lattice has."""
    rng = np.random.default_rng(30)
    base = (rng.uniform(-5, 5, size=(1500, 2))).astype(np.float32)
    test = base[rng.integers(0, len(base), 600)] + (rng.normal(size=(600, 2)) * 0.006).astype(np.float32)
    n3 = len(O.difference_indices(base, test, 0.03))
    n2 = len(lattice_difference(base, test, 0.03))
    assert n3 >= n2
    print("resol 0.03: octree %d new-voxel points, x-y lattice %d" % (n3, n2))


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
