"""Diagnostic: a batch several times the chip (every workgroup owns several scans in turn)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ndt_slam_amd import capi, synth
cfg = synth.CONFIGS["C3"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = synth.make_map(cfg["n_map"], cfg["half"])
sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
scans, off, truths, inits = sf.batch(0, B)
ctx = capi.Context(0)
gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
r0 = None
for rep in range(3):
    r = gm.align_batch(scans, off, inits)
    ms = ctx.last_timing()[1]
    print("B=%d align ms %.3f -> %.0f matches/s, status ok %s, identical to first run %s" %
          (B, ms, B / ms * 1e3, bool(np.all(r["status"] == 0)), r0 is None or r.tobytes() == r0.tobytes()))
    r0 = r if r0 is None else r0
half = gm.align_batch(scans[:int(off[256])], off[:257], inits[:256])
print("first 256 alone == first 256 of the big batch:", half.tobytes() == r0[:256].tobytes())
