"""Soak of the map build (single-pass offsets scan with look-back, two-phase rebuild): many rebuilds of one map handle from
two clouds in turn, a third of them while a batch of matches runs on another stream (the build's workgroups then start late
and in any order); every build's exported cell table compared with the first build of that cloud, bit for bit.
Usage: python tools/soak_build.py [builds] [config]"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndt_slam_amd import capi, synth               # noqa: E402


def digest(e):
    h = hashlib.sha256()
    for k in ("idx", "npts", "cent", "mean", "icov"):
        h.update(np.ascontiguousarray(e[k]).tobytes())
    return h.hexdigest()[:16]


def main():
    builds = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    cfg = synth.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "C3"]
    m = synth.make_map(cfg["n_map"], cfg["half"])
    mv = m.copy()
    mv[0] = m.min(axis=0) - np.float32(3 * cfg["resolution"])          # another bounding box: every voxel index moves
    dev = torch.device("cuda", 0)
    clouds = [torch.from_numpy(m).to(dev), torch.from_numpy(mv).to(dev)]
    sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
    B = 96
    scans, off, truths, inits = sf.batch(0, B)
    d_s = torch.from_numpy(scans).to(dev); d_o = torch.from_numpy(off.astype(np.int64)).to(dev); d_i = torch.from_numpy(inits).to(dev)
    d_r = torch.zeros(B * capi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    bstream, mstream = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev)
    bctx, mctx = capi.Context(0), capi.Context(0)
    bctx.set_stream(bstream.cuda_stream); mctx.set_stream(mstream.cuda_stream)
    prm = capi.default_params(resolution=cfg["resolution"])
    gm = capi.Map(bctx, params=prm, dev_ptr=clouds[0].data_ptr(), n=len(m), stride=8)
    other = capi.Map(mctx, params=prm, dev_ptr=clouds[0].data_ptr(), n=len(m), stride=8)      # what the matches read
    torch.cuda.synchronize()
    want = {}
    bad = 0
    t0 = time.time()
    for i in range(builds):
        c = (i // 3) % 2
        if i % 3 == 0:                                   # keep the chip busy with a persistent match kernel meanwhile
            other.align_batch_dev(d_s.data_ptr(), d_o.data_ptr(), B, len(scans), d_i.data_ptr(), d_r.data_ptr(),
                                  stream=mstream.cuda_stream, ctx=mctx)
        gm.rebuild(dev_ptr=clouds[c].data_ptr(), n=len(m), stride=8)
        torch.cuda.synchronize()
        d = digest(gm.export())
        if c not in want:
            want[c] = d
        elif d != want[c]:
            bad += 1
            print("build %d of cloud %d: table differs (%s != %s)" % (i, c, d, want[c]), flush=True)
        if i % 200 == 0:
            print("build %d %.1f s" % (i, time.time() - t0), flush=True)
    print("soak_build: %d builds, %d problems, %.1f s" % (builds, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
