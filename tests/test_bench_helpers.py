"""CPU checks of bench.py's bookkeeping (the bench itself needs a GPU)."""
import json
import os

import numpy as np

import bench


def test_algorithmic_bytes_follow_survey_8d():
    res = np.zeros(2, dtype=[("evals", "i4"), ("kbar", "f8")])
    res["evals"] = [10, 20]; res["kbar"] = [2.0, 3.0]
    n = 1000
    want = 10 * n * (8 + 20 * 2.0) + 16 * n + 20 * n * (8 + 20 * 3.0) + 16 * n
    assert bench.algorithmic_bytes(res, n) == want


def test_traffic_summary_is_consistent_with_the_pmc_files():
    traffic, src = bench.measured_traffic()
    root = os.path.dirname(os.path.abspath(bench.__file__))
    if traffic is None:
        assert not os.path.isdir(os.path.join(root, "profiles")) or not [f for f in os.listdir(os.path.join(root, "profiles")) if f.endswith("_traffic.json")]
        return
    t = json.load(open(os.path.join(root, src)))
    # rocprofv3 reports KB; FETCH_SIZE counts half the bytes fetched on gfx950 (calibrated for this kernel's access
    # widths: profiles/r02_calib_fetch.txt), WRITE_SIZE is exact (MI355X_MICROARCH.md, HBM section)
    fetch_factor = 2.0 if src.endswith("r02_traffic.json") or "calib_fetch" in t.get("note", "") else 1.0
    assert traffic == (fetch_factor * t["fetch_size_kb"] + t["write_size_kb"]) * 1024.0
    assert 1e6 < traffic < 1e11
