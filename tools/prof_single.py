"""Diagnostic workload for rocprofv3: B matches of the bench workload (no phase timers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ndt_slam_amd import capi, synth
cfg = synth.CONFIGS["C3"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = synth.make_map(cfg["n_map"], cfg["half"])
sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
scans, off, truths, inits = sf.batch(0, B)
ctx = capi.Context(0)
gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
for rep in range(reps):
    r = gm.align_batch(scans, off, inits)
    print("align ms", ctx.last_timing()[1], "evals", r["evals"].sum(), "kbar", r["kbar"].mean())
for rep in range(3):
    gm.eval_at(scans[:int(off[1])], inits[0])
