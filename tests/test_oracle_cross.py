"""Cross-implementation and derivation tests ('Derivation', 'Cross-implementation',
'Known-answer' rows).  The C oracle, an independent NumPy restatement written from the
equations, sympy and finite differences must agree -- that is how parity is established
without PCL."""
import math

import numpy as np
import pytest

from oracle import ndt_numpy as NP


@pytest.fixture(scope="module")
def c1(oracle, c1_world):
    m, sf, cfg = c1_world
    prm = oracle.default_params(resolution=cfg["resolution"])
    return oracle.Map(m, prm), NP.Cells(m, cfg["resolution"]), prm


def test_cell_table_c_vs_numpy(oracle, c1):
    M, cells, _ = c1
    t = M.export()
    assert np.array_equal(t["idx"], cells.idx)
    assert np.array_equal(t["npts"], cells.npts)
    assert np.array_equal(t["cent"], cells.cent)            # float32 sequential sums: bit exact
    assert t["mean"] == pytest.approx(cells.mean, rel=1e-13, abs=1e-13)
    # closed-form 2x2 eigen path vs LAPACK 3x3 eigh + general inverse
    assert np.allclose(t["icov"], cells.icov, rtol=1e-8, atol=1e-8)


def test_eval_c_vs_numpy(oracle, c1, c1_world):
    M, cells, prm = c1
    _, sf, cfg = c1_world
    d1, d2 = oracle.gauss(prm)
    for k in range(4):
        scan, truth, init = sf.make(k)
        for p in (init, truth, truth + [0.02, -0.01, 0.003]):
            s, g, H, pairs = M.eval_at(scan, p)
            tr, _ = NP.transform32(scan, p)
            s2, g2, H2, n2 = NP.score_grad_hess(cells, scan, tr, p[2], d1, d2)
            assert pairs == n2
            assert s == pytest.approx(s2, rel=1e-9)
            assert g == pytest.approx(g2, rel=1e-8, abs=1e-9 * np.abs(g2).max())
            assert H == pytest.approx(H2, rel=1e-8, abs=1e-9 * np.abs(H2).max())


def test_derivatives_against_sympy():
    """Symbolic d/dp and d2/dp2 of  s(p) = -d1 exp(-d2/2 q^T S^-1 q),  q = R(yaw) x + t - mu."""
    sp = pytest.importorskip("sympy")
    tx, ty, th, x, y, mx, my, a, b, c, d1, d2 = sp.symbols("tx ty th x y mx my a b c d1 d2", real=True)
    q = sp.Matrix([sp.cos(th) * x - sp.sin(th) * y + tx - mx, sp.sin(th) * x + sp.cos(th) * y + ty - my])
    S = sp.Matrix([[a, b], [b, c]])
    s = -d1 * sp.exp(-d2 / 2 * (q.T * S * q)[0, 0])
    P = [tx, ty, th]
    grad = [sp.diff(s, v) for v in P]
    hess = [[sp.diff(s, u, v) for v in P] for u in P]
    f = sp.lambdify([tx, ty, th, x, y, mx, my, a, b, c, d1, d2], [s, grad, hess], "math")
    rng = np.random.default_rng(7)

    class OneCell:       # minimal stand-in for NP.Cells with a fixed pairing
        pass
    for _ in range(25):
        vals = dict(tx=rng.normal(), ty=rng.normal(), th=rng.uniform(-3, 3), x=rng.normal() * 5,
                    y=rng.normal() * 5, a=rng.uniform(1, 50), c=rng.uniform(1, 50), d1=-0.7, d2=0.75)
        vals["b"] = rng.uniform(-0.9, 0.9) * math.sqrt(vals["a"] * vals["c"])
        xt = math.cos(vals["th"]) * vals["x"] - math.sin(vals["th"]) * vals["y"] + vals["tx"]
        yt = math.sin(vals["th"]) * vals["x"] + math.cos(vals["th"]) * vals["y"] + vals["ty"]
        vals["mx"], vals["my"] = xt + rng.normal() * 0.2, yt + rng.normal() * 0.2
        es, eg, eh = f(*[vals[k] for k in ["tx", "ty", "th", "x", "y", "mx", "my", "a", "b", "c", "d1", "d2"]])
        cells = OneCell()
        cells.mean = np.array([[vals["mx"], vals["my"]]]); cells.icov = np.array([[vals["a"], vals["b"], vals["c"]]])
        scan = np.array([[vals["x"], vals["y"]]]); trans = np.array([[xt, yt]])
        s_, g_, H_, _ = NP.score_grad_hess(cells, scan, trans, vals["th"], vals["d1"], vals["d2"],
                                           pairs=(np.array([0]), np.array([0])))
        assert s_ == pytest.approx(es, rel=1e-12)
        assert g_ == pytest.approx(np.array(eg), rel=1e-10, abs=1e-12)
        assert H_ == pytest.approx(np.array(eh), rel=1e-10, abs=1e-10)


def test_gradient_hessian_finite_differences(oracle, c1, c1_world):
    """Central differences on the fp64 NumPy restatement with the neighbour set frozen."""
    M, cells, prm = c1
    _, sf, _ = c1_world
    d1, d2 = oracle.gauss(prm)
    scan, truth, _ = sf.make(3)
    p0 = truth + np.array([0.03, -0.02, 0.004])
    scan64 = scan.astype(np.float64)

    def f(p, pairs):
        c, s = math.cos(p[2]), math.sin(p[2])
        tr = np.stack([c * scan64[:, 0] - s * scan64[:, 1] + p[0], s * scan64[:, 0] + c * scan64[:, 1] + p[1]], 1)
        return NP.score_grad_hess(cells, scan64, tr, p[2], d1, d2, pairs=pairs)
    tr32, _ = NP.transform32(scan, p0)
    pairs = cells.neighbours(tr32)
    s0, g0, H0, _ = f(p0, pairs)
    h = 1e-6
    for i in range(3):
        e = np.zeros(3); e[i] = h
        sp_, gp, _, _ = f(p0 + e, pairs); sm, gm, _, _ = f(p0 - e, pairs)
        assert (sp_ - sm) / (2 * h) == pytest.approx(g0[i], rel=1e-6, abs=1e-6)
        assert (gp - gm) / (2 * h) == pytest.approx(H0[i], rel=1e-6, abs=1e-4)
    # and the C oracle agrees with the same quantities at p0 (float32 transform inside)
    s, g, H, _ = M.eval_at(scan, p0)
    assert g == pytest.approx(g0, rel=1e-3, abs=1e-3 * np.abs(g0).max())


def test_align_c_vs_numpy_same_path(oracle, c1, c1_world):
    """Same step lengths, same number of evaluations, same final float32 matrix."""
    M, cells, prm = c1
    _, sf, cfg = c1_world
    for k in (0, 3, 4, 5):
        scan, truth, init = sf.make(k)
        r, tr = M.align(scan, init, trace_cap=600)
        n = NP.align(cells, scan, init, cfg["resolution"])
        assert int(r["iters"]) == n["iters"] and bool(r["converged"]) == n["converged"]
        steps_c = tr[:, 0]
        # the C trace has no row for Hessian-only passes
        steps_n = np.array([a for a, _ in n["log"]])
        assert len(steps_c) == len(steps_n)
        assert steps_c == pytest.approx(steps_n, rel=1e-7, abs=1e-12)
        assert tr[:, 1] == pytest.approx(np.array([s for _, s in n["log"]]), rel=1e-8)
        assert (r["T00"], r["T10"], r["T03"], r["T13"]) == tuple(n["T"])
        assert r["pose"] == pytest.approx(n["pose"], abs=1e-12)
        assert r["H"].reshape(3, 3) == pytest.approx(n["H"], rel=1e-7, abs=1e-7 * np.abs(n["H"]).max())
        assert int(r["ref_evals"]) == n["evals"] + 0


def test_fitness_vs_brute_force(oracle, c1, c1_world):
    M, _, _ = c1
    m, sf, _ = c1_world
    scan, truth, init = sf.make(2)
    for p in (truth, init, truth + [3.0, -2.0, 0.5], [100.0, -80.0, 1.0]):   # incl. far outside the map
        _, T = NP.transform32(scan, p)
        got = M.fitness(scan, *[float(v) for v in T])
        assert got == pytest.approx(NP.fitness(m, scan, T), rel=1e-12)


def test_known_transform_recovery(oracle, c1_world):
    """Scan = map subset moved by a known SE(2): recovered pose within the line-search dither."""
    from ndt_slam_amd import synth
    cfg = synth.CONFIGS["C1"]
    m = synth.make_map(20000, 40.0, seed=11)
    sf = synth.ScanFactory(m, 40.0, 1500)
    M = oracle.Map(m, oracle.default_params(resolution=0.5))
    ok = 0
    for k in range(8):
        scan, truth, init = sf.make(k)
        r = M.align(scan, truth + 0.25 * (init - truth))
        err = r["pose"] - truth
        err[2] = (err[2] + math.pi) % (2 * math.pi) - math.pi
        if abs(err[0]) < 0.02 and abs(err[1]) < 0.02 and abs(err[2]) < 2e-3:
            ok += 1
        assert r["converged"] == 1 and r["fitness"] < 0.05
    assert ok >= 7


def test_batch_equals_single_and_threads(oracle, c1, c1_world):
    M, _, _ = c1
    _, sf, _ = c1_world
    scans, off, truths, inits = sf.batch(0, 6)
    r1 = M.align_batch(scans, off, inits, nthreads=1)
    r4 = M.align_batch(scans, off, inits, nthreads=4)
    assert r1.tobytes() == r4.tobytes()
    for b in range(6):
        single = M.align(scans[int(off[b]):int(off[b + 1])], inits[b])
        assert single.tobytes() == r1[b].tobytes()


def test_version_switches_are_live(oracle, c1_world):
    m, sf, cfg = c1_world
    scan, truth, init = sf.make(0)
    base = oracle.Map(m, oracle.default_params(resolution=0.3)).align(scan, init)
    for kw in (dict(transform_sse=0), dict(stale_h_ang=1), dict(cov_unbiased=1), dict(cov_init_identity=0)):
        r = oracle.Map(m, oracle.default_params(resolution=0.3, **kw)).align(scan, init)
        assert r["status"] == 0
        assert abs(r["score"] - base["score"]) > 0 or kw == dict(stale_h_ang=1)
