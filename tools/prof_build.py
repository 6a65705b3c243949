"""Diagnostic: the map build (row a2) alone, for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_build.py [reps]`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ndt_slam_amd import capi, synth
cfg = synth.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "C3"]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = synth.make_map(cfg["n_map"], cfg["half"])
ctx = capi.Context(0)
d_map = torch.from_numpy(m).cuda()
gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
torch.cuda.synchronize()
t = []
for r in range(reps):
    gm.rebuild(dev_ptr=d_map.data_ptr(), n=len(m), stride=8)
    torch.cuda.synchronize()
    t.append(ctx.last_timing()[0])
print("map build ms: median %.4f min %.4f" % (np.median(t), np.min(t)))
