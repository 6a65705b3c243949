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


def test_gpus_n_without_a_launcher_starts_the_ranks_itself(monkeypatch, capsys):
    """`python bench.py --gpus 8` is one command (the form the driver uses at N = 1): without WORLD_SIZE the process becomes
    the launcher -- a child `python -m torch.distributed.run`, one rank per GPU on 127.0.0.1 -- relays rank 0's JSON line,
    passes everything else to stderr and exits with the ranks' status; it never reaches the rank code (no GPU here)."""
    import subprocess
    import sys
    import pytest
    seen = {}

    class FakeProc:
        def __init__(self, cmd, **kw):
            seen["cmd"], seen["kw"] = cmd, kw
            self.stdout = iter(["noise from a rank\n", '{"metric": "m", "n_gpus": 8}\n'])

        def wait(self):
            return 3

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8", "--steps", "2", "--warmup", "1"])
    assert e.value.code == 3                                         # a failing rank fails the command
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["kw"]["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert out.out == '{"metric": "m", "n_gpus": 8}\n' and "noise from a rank" in out.err


def test_a_rank_under_a_launcher_of_the_wrong_size_says_so(monkeypatch):
    import pytest
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.init_env(bench.parse_args(["--gpus", "4"]))
    assert "WORLD_SIZE 2 != --gpus 4" in str(e.value)
