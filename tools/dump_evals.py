import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from ndt_slam_amd import capi, synth
cfg = synth.CONFIGS["C3"]
m = synth.make_map(cfg["n_map"], cfg["half"])
sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
ctx = capi.Context(0)
gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
scans, off, truths, inits = sf.batch(0, 2048)
r = gm.align_batch(scans, off, inits)
os.makedirs("gpurun_out/ev", exist_ok=True)
np.save("gpurun_out/ev/evals2048.npy", r["evals"])
print(r["evals"][:256].mean(), r["evals"][:256].max())
