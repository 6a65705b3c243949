"""Diagnostic: the batched pre-filter alone (256 scans x 30k raw points), time per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ndt_slam_amd import capi, synth
B = 256
cfg = synth.CONFIGS["C3"]
m = synth.make_map(cfg["n_map"], cfg["half"])
if len(sys.argv) > 1 and sys.argv[1] == "distinct":          # 30k distinct returns per scan: nearly every point flushes a voxel
    scans, off, _, _ = synth.ScanFactory(m, cfg["half"], 30000).batch(0, B)
else:                                                        # the bench's raw scans: every return three times, 4 mm of noise
    s10, o10, _, _ = synth.ScanFactory(m, cfg["half"], cfg["n_scan"]).batch(0, B)
    rng = np.random.default_rng(11)
    scans = (np.repeat(s10, 3, axis=0) + rng.normal(0, 0.004, (3 * len(s10), 2)).astype(np.float32)).astype(np.float32)
    off = o10.astype(np.int64) * 3
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
st = torch.cuda.Stream(device=dev); ctx.set_stream(st.cuda_stream)
d_raw = torch.from_numpy(scans).to(dev); d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
d_out = torch.zeros_like(d_raw); d_ooff = torch.zeros(B + 1, dtype=torch.int64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for it in range(12):
    e0.record(st)
    ctx.prefilter_batch_dev(d_raw.data_ptr(), 8, d_off.data_ptr(), B, len(scans), 0.05, d_out.data_ptr(), d_ooff.data_ptr(), stream=st.cuda_stream)
    e1.record(st); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("prefilter_batch_dev: median %.4f ms, kept %d of %d points" % (np.median(ts[2:]), int(d_ooff[-1].item()), len(scans)))
