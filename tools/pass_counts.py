"""No GPU: the CPU checker's per-pass traces of the 256 bench matches.  (i) How predictable is the number of passes a match still
needs from what its owner knows after k passes (score, gradient, step)?  Hardly: 8-29 % of the variance.  (ii) What do the
13-20-pass matches have in common?  A last line search stuck at the minimum step length: the same trial ten times over
(LOG R4.9).   python tools/pass_counts.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from ndt_slam_amd import synth
from oracle import ndt_oracle as O
c = synth.CONFIGS["C3"]
m = synth.make_map(c["n_map"], c["half"])
sf = synth.ScanFactory(m, c["half"], c["n_scan"])
om = O.Map(m, O.default_params(resolution=c["resolution"]))
rows = []
for i in range(256):
    scan, truth, init = sf.make(i)
    r, tr = om.align(scan, init, trace_cap=64)
    E = len(tr)                                  # oracle passes (incl. Hessian-only ones): proxy for the kernel's evals
    rows.append((E, int(r["evals"]) if "evals" in r.dtype.names else E, int(r["iters"]), tr))
E = np.array([r[0] for r in rows]); it = np.array([r[2] for r in rows])
print("passes: mean %.2f max %d | iters mean %.2f max %d" % (E.mean(), E.max(), it.mean(), it.max()))
print("hist passes", np.bincount(E))
# features after k passes
for k in (2, 3, 4, 6):
    X = []; y = []
    for (e, _, _, tr) in rows:
        if e <= k: continue
        a = tr[k - 1]                            # a_t, score, g(3), p(3)
        gn = np.linalg.norm(a[2:5]); step = np.linalg.norm(tr[k - 1][5:8] - tr[k - 2][5:8]) if k >= 2 else 0
        X.append((a[0], a[1], np.log10(gn + 1e-12), np.log10(step + 1e-12))); y.append(e - k)
    X = np.array(X); y = np.array(y)
    print("after %d passes: %d scans unfinished, remaining mean %.2f sd %.2f" % (k, len(y), y.mean(), y.std()))
    for j, name in enumerate(("a_t", "score", "log|g|", "log|dp|")):
        print("   corr(remaining, %s) = %+.2f" % (name, np.corrcoef(X[:, j], y)[0, 1]))
    # simple predictor: linear fit on all four
    A = np.c_[X, np.ones(len(X))]
    w, *_ = np.linalg.lstsq(A, y, rcond=None)
    res = y - A @ w
    print("   linear fit: residual sd %.2f (vs %.2f)  -> explains %.0f %%" % (res.std(), y.std(), 100 * (1 - res.var() / y.var())))
print()
for (e, _, its, tr) in rows:
    if e >= 13:
        print(e, its, " ".join("%.3g" % a for a in tr[:, 0]))
