"""Where does a moved bounding box cost its time?  Per-step wall times of the two-phase rebuild + one launch of 256 matches
with the cloud's box moving every 8th step, every step synchronised (tools: not the pipelined bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ndt_slam_amd import capi, synth
cfg = synth.CONFIGS["C3"]; B = 256
m = synth.make_map(cfg["n_map"], cfg["half"])
sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
scans, off, truths, inits = sf.batch(0, B)
dev = torch.device("cuda", 0)
mv = m.copy(); mv[0] = m.min(axis=0) - np.float32(cfg["resolution"])
clouds = [torch.from_numpy(m).to(dev), torch.from_numpy(mv).to(dev)]
d_s = torch.from_numpy(scans).to(dev); d_o = torch.from_numpy(off.astype(np.int64)).to(dev); d_i = torch.from_numpy(inits).to(dev)
d_r = torch.zeros(B * capi.RESULT_BYTES, dtype=torch.uint8, device=dev)
ctx = capi.Context(0); prm = capi.default_params(resolution=cfg["resolution"])
gm = capi.Map(ctx, params=prm, dev_ptr=clouds[0].data_ptr(), n=len(m), stride=8)
torch.cuda.synchronize()
for i in range(40):
    c = clouds[(i // 8) % 2]
    t0 = time.perf_counter()
    gm.rebuild_begin(c.data_ptr(), len(m), 8)
    t1 = time.perf_counter()
    gm.align_batch_dev(d_s.data_ptr(), d_o.data_ptr(), B, len(scans), d_i.data_ptr(), d_r.data_ptr())
    t2 = time.perf_counter()
    stale = gm.rebuild_end()
    t3 = time.perf_counter()
    if stale:
        gm.align_batch_dev(d_s.data_ptr(), d_o.data_ptr(), B, len(scans), d_i.data_ptr(), d_r.data_ptr())
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("step %2d stale %d | begin %.3f launch %.3f end %.3f rest %.3f ms | map/align ms %s" % (i, stale, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, ctx.last_timing()))
