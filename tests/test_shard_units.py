"""Host logic of the N > 1 path without the oracle: slicing, the scatter from any source rank, the gather's
all-gather fall-back, the arg-max with ties / empty shards, and a world of one.  `gloo` on CPU tensors; the same
calls run over RCCL on the GPU box (ndt_slam_amd/shard.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ndt_slam_amd import shard


def _batch(B, seed=3):
    """Ragged batch: scan b has (b * 5) % 11 points -- some scans are empty."""
    rng = np.random.default_rng(seed)
    lens = np.array([(b * 5) % 11 for b in range(B)], np.int64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    scans = rng.standard_normal((int(off[-1]), 2)).astype(np.float32)
    inits = rng.standard_normal((B, 3))
    return scans, off, inits


@pytest.mark.parametrize("B,world", [(7, 2), (7, 8), (3, 8), (0, 4), (256, 8), (1, 1)])
def test_shard_batch_rebases_offsets(B, world):
    scans, off, inits = _batch(B)
    seen_pts, seen_in = [], []
    for r in range(world):
        sc, of, ini = shard.shard_batch(scans, off, inits, world, r)
        lo, hi = shard.shard_bounds(B, world, r)
        assert of.dtype == np.uint64 and len(of) == hi - lo + 1 and int(of[0]) == 0 and int(of[-1]) == len(sc)
        assert np.array_equal(np.diff(of.astype(np.int64)), np.diff(off[lo:hi + 1].astype(np.int64)))
        seen_pts.append(sc); seen_in.append(ini)
    assert np.concatenate(seen_pts).tobytes() == scans.tobytes()          # every point exactly once, in batch order
    assert np.concatenate(seen_in).tobytes() == inits.tobytes()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _verdict(outdir, rank, ok):
    open(os.path.join(outdir, "ok%d" % rank), "w").write("1" if ok else "0")


def _scatter_worker(rank, world, port, outdir, src, B):
    _init(rank, world, port)
    scans, off, inits = _batch(B)                                          # every rank can rebuild the truth
    got = shard.scatter_batch(*((scans, off, inits) if rank == src else (None, None, None)), src=src)
    want = shard.shard_batch(scans, off, inits, world, rank)
    ok = (got[0].numpy().tobytes() == np.ascontiguousarray(want[0]).tobytes()
          and np.array_equal(got[1].numpy(), want[1].astype(np.int64))
          and got[2].numpy().tobytes() == np.ascontiguousarray(want[2]).tobytes()
          and tuple(got[0].shape[1:]) == (2,) and tuple(got[2].shape[1:]) == (3,))
    _verdict(outdir, rank, ok)
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world,src,B", [(3, 2, 7), (2, 1, 1), (4, 0, 2)])
def test_scatter_from_any_source_rank(tmp_path, world, src, B):
    """The batch may start on any rank; shards of ranks beyond the batch are empty tensors of the right shape."""
    mp.spawn(_scatter_worker, args=(world, _free_port(), str(tmp_path), src, B), nprocs=world, join=True)
    assert all(open(os.path.join(str(tmp_path), "ok%d" % r)).read() == "1" for r in range(world))


def _gather_worker(rank, world, port, outdir, force_fallback):
    _init(rank, world, port)
    if force_fallback:
        shard._GATHER_OK = False                                           # what a backend without gather leaves behind
    mine = torch.full((24,), rank + 1, dtype=torch.uint8)
    got = shard.gather_results(mine, dst=1)
    if rank == 1:
        ok = got is not None and len(got) == world and all(bool((got[r] == r + 1).all()) for r in range(world))
    else:
        ok = got is None
    _verdict(outdir, rank, ok)
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("force_fallback", [False, True])
def test_gather_and_its_all_gather_fallback(tmp_path, force_fallback):
    world = 3
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path), force_fallback), nprocs=world, join=True)
    assert all(open(os.path.join(str(tmp_path), "ok%d" % r)).read() == "1" for r in range(world))


def _argmax_worker(rank, world, port, outdir):
    _init(rank, world, port)
    checks = []
    # (i) only the last rank has hypotheses at all
    sc = torch.tensor([0.25, 0.75, 0.5], dtype=torch.float64) if rank == world - 1 else torch.zeros(0, dtype=torch.float64)
    gi = torch.tensor([40, 41, 42], dtype=torch.int64) if rank == world - 1 else torch.zeros(0, dtype=torch.int64)
    t, i = shard.best_hypothesis_t(sc, gi)
    checks.append(float(t) == 0.75 and int(i) == 41)
    # (ii) the same best score on every rank: the lowest global index wins wherever it lives
    t, i = shard.best_hypothesis_t(torch.tensor([1.5, 1.5], dtype=torch.float64),
                                   torch.tensor([100 - rank, 200 + rank], dtype=torch.int64))
    checks.append(float(t) == 1.5 and int(i) == 100 - (world - 1))
    # (iii) nobody has a hypothesis: -inf and the sentinel index, no hang and no exception
    t, i = shard.best_hypothesis_t(torch.zeros(0, dtype=torch.float64), torch.zeros(0, dtype=torch.int64))
    checks.append(float(t) == float("-inf") and int(i) == torch.iinfo(torch.int64).max)
    # (iv) negative scores (NDT scores of bad seeds): the maximum, not the largest magnitude
    t, i = shard.best_hypothesis(np.array([-3.0 - rank, -2.0 - rank]), 2 * rank)
    checks.append(t == -2.0 and i == 1)
    _verdict(outdir, rank, all(checks))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_arg_max_ties_empty_shards_and_negative_scores(tmp_path, world):
    mp.spawn(_argmax_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(open(os.path.join(str(tmp_path), "ok%d" % r)).read() == "1" for r in range(world))


def _bcast_map_worker(rank, world, port, outdir, src, n, chunk):
    _init(rank, world, port)
    cloud = np.random.default_rng(11).standard_normal((n, 2)).astype(np.float32)     # every rank can rebuild the truth
    calls = []
    real = dist.broadcast

    def counting(t, src=0, **kw):
        calls.append(t.numel() * t.element_size())
        return real(t, src=src, **kw)
    dist.broadcast = counting
    try:
        if rank == src and n % 2:                                                # the source may hand a tensor over as well
            got = shard.broadcast_map(torch.from_numpy(cloud), src=src, chunk_bytes=chunk)
        else:
            got = shard.broadcast_map(cloud if rank == src else None, src=src, chunk_bytes=chunk)
    finally:
        dist.broadcast = real
    pieces = calls[1:]                                                           # calls[0]: the point count
    ok = (got.dtype == torch.float32 and tuple(got.shape) == (n, 2) and got.numpy().tobytes() == cloud.tobytes()
          and calls[0] == 8 and sum(pieces) == 8 * n and all(b <= max(chunk, 8) and b % 8 == 0 for b in pieces)
          and len(pieces) == (0 if n == 0 else -(-8 * n // max(8, chunk // 8 * 8))))
    _verdict(outdir, rank, ok)
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world,src,n,chunk", [(2, 0, 5000, 4096), (3, 2, 777, 1000), (4, 1, 0, 4096), (2, 1, 3, 1 << 20)])
def test_broadcast_map_from_any_rank_in_bounded_pieces(tmp_path, world, src, n, chunk):
    """shard.broadcast_map: the cloud arrives byte for byte on every rank, from any source rank, in collectives of at
    most `chunk_bytes` of whole points; an empty cloud is an empty [0, 2] tensor."""
    mp.spawn(_bcast_map_worker, args=(world, _free_port(), str(tmp_path), src, n, chunk), nprocs=world, join=True)
    assert all(open(os.path.join(str(tmp_path), "ok%d" % r)).read() == "1" for r in range(world))


def _solo_worker(rank, world, port, outdir):
    _init(rank, world, port)
    scans, off, inits = _batch(5)
    sc, of, ini = shard.scatter_batch(scans, off, inits, src=0)
    ok = sc.numpy().tobytes() == scans.tobytes() and np.array_equal(of.numpy(), off.astype(np.int64)) and ini.numpy().tobytes() == inits.tobytes()
    got = shard.gather_results(torch.arange(8, dtype=torch.uint8), dst=0)
    ok = ok and len(got) == 1 and bool((got[0] == torch.arange(8, dtype=torch.uint8)).all())
    ok = ok and shard.best_hypothesis(np.array([0.1, 0.7, 0.7]), 10) == (0.7, 11)
    ok = ok and shard.broadcast_map(scans, src=0).numpy().tobytes() == scans.tobytes()
    _verdict(outdir, rank, ok)
    dist.destroy_process_group()


def test_world_of_one_is_the_identity(tmp_path):
    """`bench.py --gpus 1` never initialises a process group, but a caller may: every helper must be the identity."""
    mp.spawn(_solo_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert open(os.path.join(str(tmp_path), "ok0")).read() == "1"
