"""The oracle's accounting of the passes the DEVICE path runs (ndt_oracle_run_stats) and its memoise switch
(bench.py's cpu_baseline.memoised): a line-search trial at the step length of the pass just run -- More-Thuente's clamp
at trans_eps / 2 (PCL `a_t = std::max (a_t, step_min)`) repeats it up to ten times -- re-uses that pass's totals."""
import numpy as np


def test_memoised_matches_are_the_full_matches(oracle, c1_world):
    m, sf, cfg = c1_world
    scans, off, _, inits = sf.batch(0, 24)
    om = oracle.Map(m, oracle.default_params(resolution=cfg["resolution"]))
    full = om.align_batch(scans, off, inits, run_stats=True)
    memo = om.align_batch(scans, off, inits, memoise=True, run_stats=True)
    for k in ("T00", "T10", "T03", "T13", "iters", "ref_evals", "converged", "score", "fitness", "H", "pose", "p",
              "evals_run", "pairs_run", "kbar_run"):
        assert np.array_equal(full[k], memo[k]), k
    # the switch is off again behind the call, and a call without it counts what the reference runs
    again = om.align_batch(scans, off, inits, run_stats=True)
    assert again.tobytes() == full.tobytes()
    # passes: reference >= run by the full port = ref_evals; the memoised port runs fewer; the device runs evals_run
    assert np.all(full["evals"] == full["ref_evals"]) and np.all(memo["evals"] <= full["evals"])
    assert np.all(memo["evals_run"] <= memo["evals"]) and (memo["evals"] < full["evals"]).any()
    # every pass the device runs has a gradient: at most one per trace row, and the first pass is always one of them
    assert np.all(full["evals_run"] >= 1) and np.all(full["evals_run"] <= full["flags"])


def test_run_stats_count_the_repeated_trials_of_the_traces(oracle, c1_world):
    """evals_run = trace rows minus the rows that repeat the step length of the row before them inside a line search."""
    m, sf, cfg = c1_world
    om = oracle.Map(m, oracle.default_params(resolution=cfg["resolution"]))
    seen_repeat = False
    for b in range(24):
        scan, _, init = sf.make(b)
        ref, tr = om.align(scan, init, trace_cap=512, run_stats=True)
        memo, tr_m = om.align(scan, init, trace_cap=512, memoise=True, run_stats=True)
        assert np.array_equal(tr, tr_m)                         # the log of the passes is the same log
        a = tr[:, 0]
        # a row with a_t == the previous row's a_t and the same trial point is a repeat (row 0 is the pass at p0, a_t = 0)
        rep = np.zeros(len(tr), bool)
        rep[2:] = (a[2:] == a[1:-1]) & np.all(tr[2:, 5:8] == tr[1:-1, 5:8], axis=1)
        seen_repeat |= bool(rep.any())
        assert int(ref["evals_run"]) == len(tr) - int(rep.sum())
        n = len(scan)
        assert ref["kbar_run"] == ref["pairs_run"] / (float(ref["evals_run"]) * n)
    assert seen_repeat
