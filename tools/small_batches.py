import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ndt_slam_amd import capi, synth
cfg = synth.CONFIGS["C3"]
m = synth.make_map(cfg["n_map"], cfg["half"])
sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
st = torch.cuda.Stream(device=dev); ctx.set_stream(st.cuda_stream)
gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
for B in (1, 4, 24, 64, 128):
    scans, off, truths, inits = sf.batch(0, B)
    d_scans = torch.from_numpy(scans).to(dev); d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_init = torch.from_numpy(inits).to(dev)
    d_res = torch.zeros(B * capi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for it in range(30):
        e0.record(st)
        gm.align_batch_dev(d_scans.data_ptr(), d_off.data_ptr(), B, len(scans), d_init.data_ptr(), d_res.data_ptr(), stream=st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    r = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)
    print("B=%3d  median %.4f ms  (evals mean %.1f max %d)  checksum %.9f" % (B, np.median(ts[10:]), r["evals"].mean(), r["evals"].max(), r["pose"].sum()))
