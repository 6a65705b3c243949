"""Diagnostic: BASELINE configs[4] per-GPU share -- 512 of the 4096 seed poses x one 10k-pt scan vs a 5M-pt map."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ndt_slam_amd import capi, synth
cfg = synth.CONFIGS["C5"]
t = time.time()
m = synth.make_map(cfg["n_map"], cfg["half"])
sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
scan, truth, _ = sf.make(0)
seeds = synth.hypothesis_seeds(truth, count=cfg["seeds"])
print("inputs in %.1f s" % (time.time() - t))
ctx = capi.Context(0)
gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
print("map build ms", ctx.last_timing()[0], "cells", gm.info().n_cells)
off = np.array([0, len(scan)], np.uint64)
for nb in (512, 512, 4096):
    r = gm.align_batch(scan, off, seeds[:nb], shared_scan=True)
    ms = ctx.last_timing()[1]
    best = int(np.argmax(r["trans_prob"]))
    err = r["pose"][best] - truth
    print("B=%d align ms %.3f  -> %.0f seeds/s | passes mean %.1f max %d | best seed %d err %.4f m %.5f rad" %
          (nb, ms, nb / ms * 1e3, r["evals"].mean(), r["evals"].max(), best, np.hypot(err[0], err[1]), err[2]))
