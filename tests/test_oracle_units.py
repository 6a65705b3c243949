"""Unit / known-answer tests pinning the CPU oracle's pieces (SURVEY.md 4 'Unit' row).
The reference has no tests; every expectation here is hand-derived or from the literature."""
import math

import numpy as np
import pytest

from oracle import ndt_numpy as NP


def test_gauss_constants_match_survey_probe(oracle):
    # SURVEY.md 8a row a3 [probe]: res 0.3 -> d1=-0.199596, d2=0.924144; 0.5 -> -0.704447, 0.756363
    d1, d2 = oracle.gauss(oracle.default_params(resolution=0.3))
    assert d1 == pytest.approx(-0.199596, abs=2e-6) and d2 == pytest.approx(0.924144, abs=2e-6)
    d1, d2 = oracle.gauss(oracle.default_params(resolution=0.5))
    assert d1 == pytest.approx(-0.704447, abs=2e-6) and d2 == pytest.approx(0.756363, abs=2e-6)
    assert NP.gauss_constants(0.5) == pytest.approx((d1, d2), rel=1e-14)


@pytest.mark.parametrize("deg", [0.3, 45.0, 89.9, 90.2, 135.0, 179.8, -0.3, -45.0, -89.9, -90.2, -135.0, -179.8])
def test_yaw_extraction_branches(oracle, deg):
    # src/PoseEstimator.cpp:31-35 on float32 entries; intrinsic error <= 2.1e-4 rad + asin/acos float
    a = math.radians(deg)
    c, s = np.float32(math.cos(a)), np.float32(math.sin(a))
    y = oracle.yaw_from_T(c, s)
    assert y == pytest.approx(a, abs=5e-4)
    assert y == NP.yaw_from_T(c, s)
    # the value is a float32 (asinf/acosf result widened)
    assert float(np.float32(y)) == y


def test_yaw_extraction_degenerate(oracle):
    assert oracle.yaw_from_T(1.0, 0.0) == 0.0          # falls to the -acos branch: -acos(1) = -0
    assert oracle.yaw_from_T(-1.0, 0.0) == pytest.approx(-math.pi, abs=1e-6)


def test_mt_trial_quadratic_known_minimiser(oracle):
    # phi(a) = (a-2)^2, psi ignored: bracket [0, 5]; case 1 (f_t > f_l): cubic == quadratic minimiser 2
    f = lambda a: (a - 2.0) ** 2
    g = lambda a: 2.0 * (a - 2.0)
    a = oracle.mt_trial(0.0, f(0), g(0), 0.0, f(0), g(0), 5.0, f(5.0), g(5.0))
    assert a == pytest.approx(2.0, abs=1e-12)
    # case 2 (f_t <= f_l, derivative changes sign): a_t = 3 -> cubic and secant both give 2
    a = oracle.mt_trial(0.0, f(0), g(0), 5.0, f(5), g(5), 3.0, f(3.0), g(3.0))
    assert a == pytest.approx(2.0, abs=1e-12)
    # case 3 (same sign, derivative shrinking): a_t = 1 -> secant 2, limited by a_t + 0.66 (a_u - a_t)
    a = oracle.mt_trial(0.0, f(0), g(0), 5.0, f(5), g(5), 1.0, f(1.0), g(1.0))
    assert a == pytest.approx(2.0, abs=1e-12)
    a = oracle.mt_trial(0.0, f(0), g(0), 1.5, f(1.5), g(1.5), 1.0, f(1.0), g(1.0))
    assert a == pytest.approx(1.0 + 0.66 * 0.5, abs=1e-12)


def test_mt_trial_cubic_known_minimiser(oracle):
    # phi(a) = a^3 - 3a has its minimiser at 1; cubic interpolation is exact
    f = lambda a: a ** 3 - 3 * a
    g = lambda a: 3 * a * a - 3
    # case 1: cubic minimiser a_c = 1 (exact), quadratic a_q = 0.5; a_c is farther from a_l,
    # so More-Thuente takes the average (a_q + a_c) / 2
    a = oracle.mt_trial(0.0, f(0), g(0), 0.0, f(0), g(0), 3.0, f(3.0), g(3.0))
    assert a == pytest.approx(0.75, abs=1e-12)
    for args in [(0.0, f(0), g(0), 3.0, f(3), g(3), 1.5, f(1.5), g(1.5)),
                 (0.0, f(0), g(0), 3.0, f(3), g(3), 0.5, f(0.5), g(0.5)),
                 (0.2, f(.2), g(.2), 0.1, f(.1), g(.1), 0.05, f(.05), g(.05))]:
        assert oracle.mt_trial(*args) == pytest.approx(NP.mt_trial(*args), rel=1e-13, nan_ok=True)


def test_mt_update_cases(oracle):
    # U1: higher value -> new upper end point
    r, v = oracle.mt_update(0.0, 0.0, -1.0, 0.0, 0.0, -1.0, 1.0, 0.5, 1.0)
    assert r == 0 and v[3:] == [1.0, 0.5, 1.0] and v[:3] == [0.0, 0.0, -1.0]
    # U2: lower value, derivative still descending away from a_l -> new lower end point
    r, v = oracle.mt_update(0.0, 0.0, -1.0, 2.0, 1.0, 1.0, 1.0, -0.5, -0.2)
    assert r == 0 and v[:3] == [1.0, -0.5, -0.2] and v[3:] == [2.0, 1.0, 1.0]
    # U3: lower value, derivative points back to a_l -> old a_l becomes a_u
    r, v = oracle.mt_update(0.0, 0.0, -1.0, 2.0, 1.0, 1.0, 1.0, -0.5, 0.2)
    assert r == 0 and v[:3] == [1.0, -0.5, 0.2] and v[3:] == [0.0, 0.0, -1.0]
    # zero derivative: converged
    r, _ = oracle.mt_update(0.0, 0.0, -1.0, 2.0, 1.0, 1.0, 1.0, -0.5, 0.0)
    assert r == 1


def test_solve3_matches_lstsq_and_pinv(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        A = rng.normal(size=(3, 3)); H = A + A.T
        b = rng.normal(size=3)
        assert oracle.solve3(H, b) == pytest.approx(np.linalg.solve(H, b), rel=1e-9, abs=1e-12)
    # rank deficient: minimum-norm solution like JacobiSVD::solve
    v = np.array([1.0, 2.0, -1.0]); H = np.outer(v, v)
    b = np.array([0.3, -0.2, 0.9])
    assert oracle.solve3(H, b) == pytest.approx(np.linalg.pinv(H) @ b, rel=1e-9, abs=1e-12)
    assert np.all(oracle.solve3(np.zeros((3, 3)), b) == 0.0)


def test_voxel_statistics_hand_case(oracle):
    # six points in one 1 m voxel + a sparse voxel that must be dropped (min_pts = 6)
    pts = np.array([[0.1, 0.2], [0.3, 0.25], [0.5, 0.2], [0.7, 0.3], [0.9, 0.2], [0.5, 0.6], [3.5, 3.5]], np.float32)
    M = oracle.Map(pts, oracle.default_params(resolution=1.0, cov_init_identity=0))
    t = M.export()
    assert M.info().n_cells == 1 and t["npts"][0] == 6
    p = pts[:6].astype(np.float64)
    mu = p.mean(0)
    cov = ((p - mu).T @ (p - mu) / 6) * (5 / 6)          # PCL <= 1.10 normalisation
    w, V = np.linalg.eigh(cov)
    w = np.maximum(w, 0.01 * w[1])
    icov = np.linalg.inv(V @ np.diag(w) @ V.T)
    assert t["mean"][0] == pytest.approx(mu, rel=1e-14)
    assert t["icov"][0] == pytest.approx([icov[0, 0], icov[0, 1], icov[1, 1]], rel=1e-10)
    c32 = np.cumsum(pts[:6], axis=0, dtype=np.float32)[-1] / np.float32(6)
    assert np.array_equal(t["cent"][0], c32)


def _leaf_identity_by_hand(p, eig_mult=0.01):
    """PCL <= 1.10 voxel statistics written out in 3-D: Leaf() starts cov_ at the identity, applyFilter adds
    pt pt^T on top and normalises with the single-pass formula and (n-1)/n."""
    n = len(p)
    p3 = np.column_stack([p, np.zeros(n)])
    S = np.eye(3) + p3.T @ p3
    pt_sum = p3.sum(0)
    mu = pt_sum / n
    cov = (S - 2 * np.outer(pt_sum, mu)) / n + np.outer(mu, mu)
    cov *= (n - 1.0) / n
    w, V = np.linalg.eigh(cov)
    if w[0] < eig_mult * w[2]:
        w[0] = eig_mult * w[2]
        if w[1] < eig_mult * w[2]:
            w[1] = eig_mult * w[2]
        cov = V @ np.diag(w) @ np.linalg.inv(V)
    return mu[:2], np.linalg.inv(cov)[:2, :2], w, V


def test_voxel_statistics_identity_default(oracle):
    """Default preset (PCL 1.10): the identity start adds (n-1)/n^2 to every diagonal entry of the covariance;
    the z eigenvalue (n-1)/n^2 takes part in the ascending sort and in the 0.01 * max inflation floor."""
    pts = np.array([[0.1, 0.2], [0.3, 0.25], [0.5, 0.2], [0.7, 0.3], [0.9, 0.2], [0.5, 0.6]], np.float32)
    prm = oracle.default_params(resolution=1.0)
    assert prm.cov_init_identity == 1 and prm.cov_unbiased == 0
    t = oracle.Map(pts, prm).export()
    mu, icov, w, V = _leaf_identity_by_hand(pts.astype(np.float64))
    assert t["mean"][0] == pytest.approx(mu, rel=1e-14)
    assert t["icov"][0] == pytest.approx([icov[0, 0], icov[0, 1], icov[1, 1]], rel=1e-10)
    # a short, nearly collinear run of 7 points: without the identity its minor variance would be inflated to
    # 1 % of the major one; with it both are czz = 6/49 plus the spread, i.e. a nearly isotropic, fat Gaussian
    x = np.linspace(0.45, 0.55, 7)
    line = np.stack([x, 0.5 + 1e-5 * np.sin(37 * x)], 1).astype(np.float32)
    t = oracle.Map(line, prm).export()
    mu, icov, w, V = _leaf_identity_by_hand(line.astype(np.float64))
    czz = 6.0 / 49.0
    assert w[0] == pytest.approx(czz, rel=1e-12) and abs(V[2, 0]) == pytest.approx(1.0)   # the z pair sorts first
    ic = np.array([[t["icov"][0][0], t["icov"][0][1]], [t["icov"][0][1], t["icov"][0][2]]])
    ev = np.linalg.eigvalsh(np.linalg.inv(ic))
    assert ev[0] == pytest.approx(czz, rel=1e-3) and ev[1] == pytest.approx(czz + np.var(x) * 6 / 7, rel=1e-3)
    assert t["icov"][0] == pytest.approx([icov[0, 0], icov[0, 1], icov[1, 1]], rel=1e-9)
    # a long wall through a 4 m voxel with many points: (n-1)/n^2 = 0.005 is below the major variance, and the
    # minor one (wall noise + 0.005) is lifted to the 1 % floor set by the MAJOR axis -- the usual inflation
    rng = np.random.default_rng(3)
    wall = np.stack([np.linspace(0.1, 3.9, 200), 2.0 + rng.normal(0, 0.002, 200)], 1).astype(np.float32)
    t = oracle.Map(wall, oracle.default_params(resolution=4.0)).export()
    mu, icov, w, V = _leaf_identity_by_hand(wall.astype(np.float64))
    assert abs(V[2, 0]) == pytest.approx(1.0) and w[2] > 1.0 and w[1] == pytest.approx(0.01 * w[2], rel=1e-12)
    assert t["icov"][0] == pytest.approx([icov[0, 0], icov[0, 1], icov[1, 1]], rel=1e-9)


def test_voxel_inflation_and_rejection(oracle):
    # nearly collinear points: lambda_min inflated to 0.01 lambda_max
    x = np.linspace(0.05, 0.95, 10)
    pts = np.stack([x, 0.5 + 1e-4 * np.sin(37 * x)], 1).astype(np.float32)
    t = oracle.Map(pts, oracle.default_params(resolution=1.0, cov_init_identity=0)).export()
    assert t["npts"][0] == 10
    ic = np.array([[t["icov"][0][0], t["icov"][0][1]], [t["icov"][0][1], t["icov"][0][2]]])
    w = np.linalg.eigvalsh(np.linalg.inv(ic))
    assert w[0] / w[1] == pytest.approx(0.01, rel=1e-6)
    # all points identical: zero covariance -> rejected voxel kept searchable with icov = 0
    pts = np.tile(np.array([[0.5, 0.5]], np.float32), (8, 1))
    t = oracle.Map(pts, oracle.default_params(resolution=1.0, cov_init_identity=0)).export()
    assert t["npts"][0] == -8 and np.all(t["icov"][0] == 0)
    # with the PCL 1.10 identity start the same voxel has covariance (n-1)/n^2 I: accepted
    t = oracle.Map(pts, oracle.default_params(resolution=1.0)).export()
    assert t["npts"][0] == 8 and t["icov"][0] == pytest.approx([64 / 7, 0.0, 64 / 7], rel=1e-12)


def test_rejected_voxel_scores_minus_d1_only(oracle):
    pts = np.tile(np.array([[0.5, 0.5]], np.float32), (8, 1))
    prm = oracle.default_params(resolution=1.0, cov_init_identity=0)
    M = oracle.Map(pts, prm)
    d1, _ = oracle.gauss(prm)
    s, g, H, pairs = M.eval_at(np.array([[0.6, 0.4]], np.float32), [0.0, 0.0, 0.0])
    assert pairs == 1 and s == pytest.approx(-d1, rel=1e-15)
    assert np.all(g == 0) and np.all(H == 0)


def test_approx_voxel_filter_hand_case(oracle):
    # two points share a 0.1 m voxel -> averaged; a third lands alone
    pts = np.array([[0.01, 0.01], [0.03, 0.05], [0.55, 0.01]], np.float32)
    out = oracle.approx_voxel_filter(pts, 0.1)
    out = out[np.argsort(out[:, 0])]
    assert len(out) == 2
    assert out[0] == pytest.approx([0.02, 0.03], rel=1e-6)
    assert np.array_equal(out[1], pts[2])


def test_approx_voxel_filter_flush_on_collision(oracle):
    # voxels (0,0) and (512*k,...) collide in the 512-slot history when the hash matches:
    # ix*7171 & 511 == 0 for ix = 512 -> the first centroid is flushed, then re-opened
    leaf = np.float32(0.1)
    a = np.array([0.05, 0.05], np.float32); b = np.array([51.25, 0.05], np.float32)
    pts = np.stack([a, b, a])
    out = oracle.approx_voxel_filter(pts, float(leaf))
    assert len(out) == 3          # order dependent: the same voxel is emitted twice


def test_switches_change_covariance(oracle):
    rng = np.random.default_rng(1)
    pts = (rng.uniform(0.05, 0.95, size=(12, 2))).astype(np.float32)
    base = oracle.Map(pts, oracle.default_params(resolution=1.0, cov_unbiased=0, cov_init_identity=0)).export()["icov"][0]
    unb = oracle.Map(pts, oracle.default_params(resolution=1.0, cov_unbiased=1, cov_init_identity=0)).export()["icov"][0]
    idn = oracle.Map(pts, oracle.default_params(resolution=1.0, cov_unbiased=0, cov_init_identity=1)).export()["icov"][0]
    assert unb == pytest.approx(base * (11 / 12) ** 2, rel=1e-9)
    assert not np.allclose(idn, base)
    c = NP.Cells(pts, 1.0, init_identity=True)
    assert idn == pytest.approx(c.icov[0], rel=1e-9)
