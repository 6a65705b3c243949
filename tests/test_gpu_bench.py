"""The bench line keeps its contract (one JSON object on stdout with the driver's keys)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-sample", "8",
                          "--cpu-reps", "1"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert d["value"] > 100 * c["value"]
    assert 0 < d["map_build_ms"] < d["ms_per_step"]
    assert d["parity"]["max_dpos_m"] <= 1e-4 and d["parity"]["max_dyaw_rad"] <= 1e-4
    assert d["parity"]["same_iters"] is True and d["parity"]["same_pairs_run"] is True
    # the like-for-like CPU figure: the same port with the repeated line-search trials skipped, as the GPU path skips them
    cm = c["memoised"]
    assert cm["identical_to_the_full_run"] is True and cm["value"] >= 0.8 * c["value"] and cm["passes_run_mean"] <= cm["passes_reference_mean"]      # (8 matches, one repetition: the timing is noise-bound here)
    # the two genuinely HBM-bound kernel groups of SURVEY 8d carry their own roofline figures
    mb, ft = r["map_build"], r["fitness"]
    assert abs(mb["algorithmic_bytes_per_build"] - (8.0 * mb["map_points"] + 24.0 * mb["voxels"])) < 1 and 0 < mb["frac"] < 1
    assert abs(mb["frac"] - mb["achieved_GBps"] / r["peak"]) < 1e-12 and 0 < ft["frac"] < 1
    # the path the reference takes: a local map whose voxel bounding box moves.  The timed region never moves it ...
    assert d["rebuilt_steps"] == 0
    # ... the side leg does: every move is found (two map buffers: two stale grids per move), the step is repeated, and the
    # records are those of a build from scratch
    mv = d["moving_map"]
    assert mv["steps"] == 48 and mv["box_moves_every"] == 8 and mv["rebuilt_steps"] == 2 * (48 // 8 - 1)
    assert mv["identical_to_a_fresh_build"] is True and 0 < mv["ms_per_step_moving"] < 3 * d["ms_per_step"]
    # ... and with ndt_params::grid_margin the grid queued ahead survives the move: same records, (almost) no repeated builds
    mm = d["moving_map_margin"]
    assert mm["grid_margin"] == 8 and mm["rebuilt_steps"] <= 2 and mm["identical_to_a_fresh_build"] is True
    assert 0 < mm["ms_per_step_moving"] < mv["ms_per_step_moving"]
    assert d["steady_state"]["steps"] == 200 and d["steady_state"]["value"] > 0


def test_two_rank_path_walks_through_on_one_gpu():
    """N > 1 code path (sharded scans, gather of the result records on a side stream, MAX over ranks, rank 0
    prints) rehearsed with both ranks on device 0 over gloo; the real run uses RCCL, one rank per GPU."""
    env = dict(os.environ, NDT_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    # ONE command, as the driver types it at N = 1: bench.py starts the two ranks itself (bench.spawn_ranks)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")          # the launcher relays rank 0's line and nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["scans_per_gpu"] == 256 and "x2" in d["config"]["parallelism"]
    assert "scatter from rank 0" in d["config"]["parallelism"] and d["comm"]["scatter_ms"] > 0
    assert len(d["per_rank"]["kernel_ms"]) == 2 and min(d["per_rank"]["kernel_ms"]) > 0
    assert d["comm"]["broadcast_map_ms"] > 0 and d["comm"]["broadcast_map_bytes"] == 8_000_000      # the target cloud's fan-out
    assert d["comm"]["broadcast_map_identical_on_every_rank"] is True and "broadcast_map_error" not in d["comm"]
    assert "cpu_baseline" not in d                     # the CPU leg runs at N = 1 only
    mh = d["multi_hypothesis"]                         # the configs[4] leg of the default run, both ranks
    assert "error" not in mh, mh
    assert mh["broadcast_scan_ms"] > 0 and 0 <= mh["best_over_all_ranks"]["seed"] < 4096 and mh["seeds_per_s"] > 0


def test_two_rank_multi_hypothesis_path_walks_through_on_one_gpu():
    """configs[4] at N > 1 (scan broadcast, seeds sharded by stride, arg-max of the scores over the ranks) rehearsed
    with both ranks on device 0 over gloo."""
    env = dict(os.environ, NDT_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--config", "C5"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["matches_per_gpu"] == 512 and d["config"]["map_points"] == 5_000_000
    g = d["best_hypothesis"]["global"]
    assert g["trans_prob"] >= d["best_hypothesis"]["trans_prob"] and 0 <= g["seed"] < 4096


def test_bench_c5_and_two_launches_in_flight():
    """`--config C5` at N = 1 and `--inflight 2` (alternating contexts) keep the contract and the results."""
    for extra in (["--config", "C5"], ["--inflight", "2"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-sample", "8",
                              "--cpu-reps", "1", "--no-single-scan"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert d["parity"]["max_dpos_m"] <= 1e-4 and d["parity"]["max_dyaw_rad"] <= 1e-4 and d["parity"]["same_iters"]
        assert 0 < d["roofline"]["frac"] < 1 and d["value"] > 0


def test_four_rank_path_with_the_ranks_contending_for_one_gpu():
    """Four ranks over gloo on the ONE device of the box: the scatter / gather bookkeeping at world size 4, and four
    processes whose persistent match kernels compete for the same CUs -- no launch has all its workgroups resident, so
    this only finishes in time because workgroups take over the scans of workgroups that are not running (claims)."""
    env = dict(os.environ, NDT_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                          "--master-addr", "127.0.0.1", "--master-port", "29523", os.path.join(ROOT, "bench.py"),
                          "--gpus", "4", "--steps", "4", "--warmup", "2", "--no-single-scan", "--no-cpu-baseline", "--map-from-rank0"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 4 and d["value"] > 0 and len(d["per_rank"]["kernel_ms"]) == 4
    assert "scatter_error" not in d["comm"] and "gather_error" not in d["comm"]
    assert "broadcast_map_error" not in d["comm"] and d["comm"]["broadcast_map_ms"] > 0     # --map-from-rank0: ranks 1-3 matched on the copy
    assert max(d["per_rank"]["kernel_ms_max"]) < 50.0              # ms: no launch waited for another process's kernels to end
    assert d["converged"] == 256 and d["accepted"] == 256


_RCCL_ONE = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from ndt_slam_amd import shard
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)                  # "nccl" IS RCCL on ROCm
rng = np.random.default_rng(1)
off = np.array([0, 5, 5, 12], np.uint64); scans = rng.standard_normal((12, 2)).astype(np.float32); inits = rng.standard_normal((3, 3))
sc, of, ini = shard.scatter_batch(scans, off, inits, src=0, device=dev)
assert sc.is_cuda and sc.cpu().numpy().tobytes() == scans.tobytes() and of.cpu().tolist() == [0, 5, 5, 12]
pay = torch.arange(2003, dtype=torch.float64, device=dev); dist.broadcast(pay, src=0)
cloud = rng.standard_normal((5000, 2)).astype(np.float32)
mt = shard.broadcast_map(cloud, src=0, device=dev, chunk_bytes=16384)              # int64 count + chunked float32 broadcasts
assert mt.is_cuda and mt.cpu().numpy().tobytes() == cloud.tobytes()
got = shard.gather_results(torch.arange(216, dtype=torch.uint8, device=dev), dst=0)
assert len(got) == 1 and got[0].cpu().tolist() == list(range(216))
t, i = shard.best_hypothesis_t(torch.tensor([0.5, 2.5, 2.5], dtype=torch.float64, device=dev), torch.tensor([8, 16, 24], dtype=torch.int64, device=dev))
assert float(t) == 2.5 and int(i) == 16
m = torch.tensor([1.25], dtype=torch.float64, device=dev); dist.all_reduce(m, op=dist.ReduceOp.MAX)
lst = [torch.empty(3, dtype=torch.float64, device=dev)]; dist.all_gather(lst, torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device=dev))
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_the_collectives_of_the_sharded_path_execute_on_rccl():
    """Every collective bench.py and shard.py issue at N > 1 (object broadcast, float64 broadcast, uint8 gather, float64 MAX /
    int64 MIN all-reduce, all-gather, barrier), issued once on the real backend -- RCCL -- with device tensors, in a world of
    one (the box has one GPU and RCCL refuses two ranks on a device): the library loads, the communicator comes up under this
    image's IPC settings, and the dtypes / reduce ops are ones RCCL implements.  Point-to-point transfers need a peer and stay
    covered by gloo (tests/test_shard_gloo.py, test_shard_units.py)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    out = subprocess.run([sys.executable, "-c", _RCCL_ONE % (ROOT, port)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
