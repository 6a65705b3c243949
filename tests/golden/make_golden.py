#!/usr/bin/env python3
"""Generates tests/golden/c1_golden.npz (default preset = PCL 1.10 semantics) and
tests/golden/c1_golden_pcl_new.npz (the PCL >= 1.11 preset: zero-initialised, unbiased voxel covariance).

The reference ships no golden vectors, and its hot path (PCL) cannot be built
or imported here, so these vectors come from this repo's CPU restatement (oracle/ndt_oracle.c,
cross-checked against the independent NumPy restatement by tests/test_oracle_cross.py).  They pin
the oracle and the HIP path against silent drift; they are NOT outputs of the reference.

Workload: BASELINE.json configs[0] -- 360-point scans vs a 5k-point map, launch-file parameters
(Resolution 0.3, StepSize 0.1, TransformationEpsilon 0.01, MaximumIterations 35).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ndt_slam_amd import synth          # noqa: E402
from oracle import ndt_oracle as O      # noqa: E402


def main(preset="default", name="c1_golden.npz"):
    cfg = synth.CONFIGS["C1"]
    m = synth.make_map(cfg["n_map"], cfg["half"])
    sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
    scans, off, truths, inits = sf.batch(0, 8)
    M = O.Map(m, O.default_params(preset, resolution=cfg["resolution"]))
    res = M.align_batch(scans, off, inits)
    _, trace0 = M.align(scans[int(off[0]):int(off[1])], inits[0], trace_cap=256)
    t = M.export()
    info = M.info()
    evals = []
    for b in range(3):
        sc = scans[int(off[b]):int(off[b + 1])]
        s, g, H, pairs = M.eval_at(sc, inits[b])
        evals.append(np.concatenate([[s], g, H.ravel(), [pairs]]))
    filt = O.approx_voxel_filter(scans[int(off[0]):int(off[1])], 0.05)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
    np.savez_compressed(
        out, map_xy=m, scans=scans, offsets=off, truths=truths, inits=inits, results=res,
        trace0=trace0, cell_idx=t["idx"], cell_cent=t["cent"], cell_mean=t["mean"], cell_icov=t["icov"],
        cell_npts=t["npts"], grid=np.array([info.min_bx, info.min_by, info.div_x, info.div_y, info.n_cells, info.n_valid]),
        evals=np.stack(evals), filtered0=filt, resolution=np.float32(cfg["resolution"]), preset=np.array(preset))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
    main("pcl_new", "c1_golden_pcl_new.npz")
