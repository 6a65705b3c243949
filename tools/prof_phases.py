"""Diagnostic: phase timing of the match kernel (NDT_PROF=1) on the bench workload."""
import os, sys, time
os.environ["NDT_PROF"] = "1"
os.makedirs("gpurun_out", exist_ok=True)
os.environ["NDT_PROF_DUMP"] = "gpurun_out/prof_dump.bin"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ndt_slam_amd import capi, synth
cfg = synth.CONFIGS["C3"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = synth.make_map(cfg["n_map"], cfg["half"])
sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
scans, off, truths, inits = sf.batch(0, B)
ctx = capi.Context(0)
gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
print("map build ms", ctx.last_timing()[0])
for rep in range(3):
    r = gm.align_batch(scans, off, inits)
    print("align ms", ctx.last_timing()[1], "evals mean/max", r["evals"].mean(), r["evals"].max())
np.save("gpurun_out/prof_res.npy", r)
os.rename("gpurun_out/prof_dump.bin", "gpurun_out/prof_dump_batch.bin")
r1 = gm.align_batch(scans[:int(off[1])], off[:2], inits[:1])
print("single: align ms", ctx.last_timing()[1], "evals", r1["evals"])
