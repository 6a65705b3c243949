"""What lies between two match launches on one stream: back-to-back launches of the bench batch (no map rebuild) with
nothing in between / an event record / a wait for an already finished event of another stream / both.
Prints the time per launch next to the kernels' own durations (ndt_kernel_timing)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndt_slam_amd import capi, synth               # noqa: E402

cfg = synth.CONFIGS["C3"]
B = 256
m = synth.make_map(cfg["n_map"], cfg["half"])
sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
scans, off, truths, inits = sf.batch(0, B)
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
st, other = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
ctx.set_stream(st.cuda_stream)
gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
d_scans = torch.from_numpy(scans).to(dev); d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
d_init = torch.from_numpy(inits).to(dev)
out = torch.zeros(B * capi.RESULT_BYTES, dtype=torch.uint8, device=dev)
done = torch.cuda.Event(); done.record(other); torch.cuda.synchronize()
N = 200
for mode in ("nothing", "record", "wait", "record+wait"):
    evs = [torch.cuda.Event() for _ in range(N)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            if "wait" in mode:
                st.wait_event(done)
            gm.align_batch_dev(d_scans.data_ptr(), d_off.data_ptr(), B, len(scans), d_init.data_ptr(), out.data_ptr(),
                               stream=st.cuda_stream, ctx=ctx)
            if "record" in mode:
                evs[i].record(st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N * 1e3
    k = [ctx.kernel_timing(j) for j in range(40)]
    km, kf = np.mean([x[0] for x in k]), np.mean([x[1] for x in k])
    print("%-12s %.4f ms per launch | match %.4f + fitness %.4f = %.4f | between launches %.1f us" % (mode, dt, km, kf, km + kf, (dt - km - kf) * 1e3))
