"""Result records of fixed workloads as hashes, to compare library builds bit for bit (run once per build with
NDT_LIB_PATH): C3 batch of 256, a ragged batch of 301, 24 scans (helpers from the start), 512 seeds of one scan on the
5M-point map (shared_scan), C1 x 24.  Usage: python tools/hash_results.py [--c5]"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndt_slam_amd import capi, synth               # noqa: E402


def h(a):
    return hashlib.sha256(a.tobytes()).hexdigest()[:16]


def main():
    ctx = capi.Context(0)
    cfg = synth.CONFIGS["C3"]
    m = synth.make_map(cfg["n_map"], cfg["half"])
    sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
    gm = capi.Map(ctx, m, capi.default_params(resolution=cfg["resolution"]))
    scans, off, truths, inits = sf.batch(0, 301)
    r = gm.align_batch(scans[:int(off[256])], off[:257], inits[:256])
    print("C3x256", h(r), "evals %.3f T %s" % (r["evals"].mean(), h(np.stack([r["T00"], r["T10"], r["T03"], r["T13"]]))), "iters", h(r["iters"]))
    keep = np.ones(len(scans), bool)
    for b in range(0, 301, 3):
        keep[int(off[b]) + 7000:int(off[b + 1])] = False
    lens = np.array([keep[int(off[b]):int(off[b + 1])].sum() for b in range(301)])
    r = gm.align_batch(scans[keep], np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64), inits)
    print("ragged301", h(r), "T", h(np.stack([r["T00"], r["T10"], r["T03"], r["T13"]])))
    r = gm.align_batch(scans[:int(off[24])], off[:25], inits[:24])
    print("C3x24", h(r), "T", h(np.stack([r["T00"], r["T10"], r["T03"], r["T13"]])))
    c1 = synth.CONFIGS["C1"]
    m1 = synth.make_map(c1["n_map"], c1["half"])
    s1 = synth.ScanFactory(m1, c1["half"], c1["n_scan"])
    g1 = capi.Map(ctx, m1, capi.default_params(resolution=c1["resolution"]))
    sc, of, tr, ini = s1.batch(0, 24)
    r = g1.align_batch(sc, of, ini)
    print("C1x24", h(r), "T", h(np.stack([r["T00"], r["T10"], r["T03"], r["T13"]])))
    if "--c5" in sys.argv:
        c5 = synth.CONFIGS["C5"]
        m5 = synth.make_map(c5["n_map"], c5["half"])
        s5 = synth.ScanFactory(m5, c5["half"], c5["n_scan"])
        scan, truth, _ = s5.make(0)
        seeds = synth.hypothesis_seeds(truth, c5["seeds"])[::8]
        g5 = capi.Map(ctx, m5, capi.default_params(resolution=c5["resolution"]))
        r = g5.align_batch(scan, np.array([0, len(scan)], np.uint64), seeds, shared_scan=True)
        print("C5x512", h(r), "T", h(np.stack([r["T00"], r["T10"], r["T03"], r["T13"]])), "iters", h(r["iters"]))


if __name__ == "__main__":
    main()
