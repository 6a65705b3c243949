"""Soak of the work-sharing protocol: many launches of the bench batch (and a ragged one), every result compared
with the first launch bit for bit; any watchdog abort, hang or race shows up as a status or a mismatch.
Usage: python tools/soak.py [launches] [B] [shared]   (shared: B seed poses of ONE scan, `shared_scan`, as configs[4])"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndt_slam_amd import capi, synth               # noqa: E402


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    shared = len(sys.argv) > 3 and sys.argv[3] == "shared"
    cfg = synth.CONFIGS["C3"]
    m = synth.make_map(cfg["n_map"], cfg["half"])
    sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
    scans, off, truths, inits = sf.batch(0, B)
    if shared:                                       # one scan, B hypotheses around its true pose
        scans, truth, _ = sf.make(0)
        off = np.array([0, len(scans)], np.uint64)
        inits = synth.hypothesis_seeds(truth, 4096)[::max(1, 4096 // B)][:B]
        B = len(inits)
    if B % 2:                                        # ragged: shorten every third scan
        keep = np.ones(len(scans), bool)
        for b in range(0, B, 3):
            keep[int(off[b]) + 7000:int(off[b + 1])] = False
        lens = np.array([keep[int(off[b]):int(off[b + 1])].sum() for b in range(B)])
        scans = scans[keep]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
    d_scans = torch.from_numpy(scans).to(dev); d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_init = torch.from_numpy(inits).to(dev)
    d_res = torch.zeros(B * capi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    first = None
    bad = 0
    t0 = time.time()
    for it in range(launches):
        gm.align_batch_dev(d_scans.data_ptr(), d_off.data_ptr(), B, len(scans), d_init.data_ptr(), d_res.data_ptr(),
                           shared_scan=shared, stream=stream.cuda_stream)
        if it % 50 == 0 or it == launches - 1:
            stream.synchronize()
            r = d_res.cpu().numpy().tobytes()
            res = np.frombuffer(r, dtype=capi.RESULT_DTYPE)
            if not np.all(res["status"] == 0):
                print("launch", it, "status", np.unique(res["status"])); bad += 1
            if first is None:
                first = r
            elif r != first:
                print("launch", it, "differs from launch 0"); bad += 1
        if it % 1000 == 0:
            print("launch", it, "%.1f s" % (time.time() - t0), flush=True)
    print("soak: %d launches of B=%d%s, %d problems, %.1f s" % (launches, B, " (shared scan)" if shared else "", bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
