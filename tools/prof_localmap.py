"""Times Submap::makeMap on the device (ndt_make_map_dev) against the oracle on the host.
Usage: python tools/prof_localmap.py [n_scans] [points_per_scan]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndt_slam_amd import capi, synth               # noqa: E402
from oracle import ndt_oracle as O                 # noqa: E402  (checker only)


def main():
    n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    scans = synth.submap_scans(n_scans, n_pts)
    off = np.zeros(n_scans + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in scans])
    ctx = capi.Context(0)
    allp = torch.from_numpy(np.concatenate(scans)).cuda()
    out = torch.empty((len(allp) + 1, 2), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    times = []
    for it in range(8):
        ev[0].record(stream)
        ctx.make_map_dev(allp.data_ptr(), 8, off, True, True, True, 0.05, 0.1, out.data_ptr(), cnt.data_ptr(),
                         stream=stream.cuda_stream)
        ev[1].record(stream)
        stream.synchronize()
        times.append(ev[0].elapsed_time(ev[1]))
    n = int(cnt.item())
    t0 = time.perf_counter()
    ref = O.make_map(scans, True, True, True, 0.05, 0.1)
    cpu = time.perf_counter() - t0
    same = n == len(ref) and out[:n].cpu().numpy().tobytes() == ref.tobytes()
    print("make_map %d scans x %d pts: device %.3f ms (best of 8, first %.3f), oracle on 1 core %.1f ms, kept %d of %d, "
          "identical %s" % (n_scans, n_pts, min(times[1:]), times[0], cpu * 1e3, n, len(allp), same))


if __name__ == "__main__":
    main()
