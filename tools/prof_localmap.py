"""Times Submap::makeMap on the device (ndt_make_map_dev) against the oracle on the host.
Usage: python tools/prof_localmap.py [n_scans] [points_per_scan]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndt_slam_amd import capi                      # noqa: E402
from oracle import ndt_oracle as O                 # noqa: E402  (checker only)


def scans_of(rng, n_scans, n_wall, n_mover):
    th = np.linspace(0, 2 * np.pi, n_wall, endpoint=False)
    d = np.maximum(abs(np.cos(th)), abs(np.sin(th)))
    room = np.stack([8 * np.cos(th) / d, 6 * np.sin(th) / d], 1)
    out = []
    for k in range(n_scans):
        mover = np.stack([rng.normal(-3 + 0.6 * k, 0.1, n_mover), rng.normal(0.5, 0.15, n_mover)], 1)
        out.append((np.concatenate([room, mover]) + rng.normal(size=(n_wall + n_mover, 2)) * 0.003).astype(np.float32))
    return out


def main():
    n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    rng = np.random.default_rng(0)
    scans = scans_of(rng, n_scans, n_pts - n_pts // 40, n_pts // 40)
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
