"""Diagnostic: does the order of the scans in a batch matter to the fitness kernels (tail of the poorly matched
scans)?  The bench batch as it is, worst matches first, worst matches last, good matches only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndt_slam_amd import capi, synth
cfg = synth.CONFIGS["C3"]
B = 256
m = synth.make_map(cfg["n_map"], cfg["half"])
sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
scans, off, truths, inits = sf.batch(0, B)
ctx = capi.Context(0)
gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
r = gm.align_batch(scans, off, inits)
fit = r["fitness"]
print("fitness pcts", np.percentile(fit, [10, 50, 80, 90, 100]), "bad (> 0.05):", int((fit > 0.05).sum()))
n = cfg["n_scan"]
def run(order, name):
    sc = np.concatenate([scans[int(off[b]):int(off[b + 1])] for b in order])
    of = np.arange(len(order) + 1, dtype=np.uint64) * n
    t = []
    for rep in range(5):
        gm.align_batch(sc, of, inits[order])
        t.append(ctx.kernel_timing(0))
    t = np.array(t)
    print("%-28s B=%3d match %.4f ms fitness %.4f ms" % (name, len(order), np.median(t[:, 0]), np.median(t[:, 1])))
idx = np.arange(B)
run(idx, "as generated")
run(np.argsort(-fit), "worst matches first")
run(np.argsort(fit), "worst matches last")
good = idx[fit <= 0.05]
run(good, "good matches only")
run(np.concatenate([good, good])[:B], "256 good")
