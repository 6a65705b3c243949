"""The N > 1 path on CPU: world_size 2 / 4 / 8 `gloo` processes shard a batch, match their shards (with the
CPU oracle standing in for the kernel -- this test is about the scatter / gather bookkeeping) and
gather the result records; rank 0 must end up with exactly the single-process results.  The batch (7 scans) is
never divisible by the world size, and with 8 ranks the last shard is EMPTY.  The target cloud exists on rank 0 only and
reaches the others through shard.broadcast_map (chunked)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ndt_slam_amd import shard


def test_shard_bounds_partition():
    for total in (0, 1, 7, 256, 2048, 4096):
        for world in (1, 2, 3, 4, 8):
            cuts = [shard.shard_bounds(total, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ndt_slam_amd import synth
    from oracle import ndt_oracle as O
    cfg = synth.CONFIGS["C1"]
    # the target cloud lives on rank 0 only (src/ScanMatcher.cpp:40: one PointCloudMap hands it over); the others get it
    # through shard.broadcast_map -- in pieces of 4096 bytes here, so that the 40 KB cloud takes ten collectives
    m0 = synth.make_map(cfg["n_map"], cfg["half"]) if rank == 0 else None
    mt = shard.broadcast_map(m0, src=0, chunk_bytes=4096)
    assert isinstance(mt, torch.Tensor) and mt.dtype == torch.float32 and tuple(mt.shape) == (cfg["n_map"], 2)
    m = mt.numpy()
    map_ok = m.tobytes() == synth.make_map(cfg["n_map"], cfg["half"]).tobytes()
    om = O.Map(m, O.default_params(resolution=cfg["resolution"]))
    B = 7                                                         # ragged split: 4 + 3
    if rank == 0:
        sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
        scans, off, truths, inits = sf.batch(0, B)
        # ragged scans too: drop a few points of scan 2
        keep = np.ones(len(scans), bool); keep[int(off[2]):int(off[2]) + 17] = False
        lens = np.diff(off.astype(np.int64)); lens[2] -= 17
        scans = scans[keep]; off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    else:
        scans = off = inits = None
    sc, of, ini = shard.scatter_batch(scans, off, inits, src=0)
    assert isinstance(sc, torch.Tensor) and sc.dtype == torch.float32 and of.dtype == torch.int64 and ini.dtype == torch.float64
    lo, hi = shard.shard_bounds(B, world, rank)
    assert len(ini) == hi - lo and len(of) == hi - lo + 1 and int(of[0]) == 0 and int(of[-1]) == len(sc)
    sc, of, ini = sc.numpy(), of.numpy().astype(np.uint64), ini.numpy()      # (device tensors on the GPU box: data_ptr())
    res = om.align_batch(sc, of, ini) if hi > lo else np.zeros(0, O.RESULT_DTYPE)
    most = shard.shard_bounds(B, world, 0)[1]                     # the largest shard: equal-size records for gather
    buf = np.zeros(most * O.RESULT_DTYPE.itemsize, np.uint8)
    buf[:len(res) * O.RESULT_DTYPE.itemsize] = np.frombuffer(res.tobytes(), np.uint8)
    got = shard.gather_results(torch.from_numpy(buf), dst=0)
    best = shard.best_hypothesis(res["trans_prob"], shard.shard_bounds(B, world, rank)[0])
    # configs[4] numbering: rank r holds every `world`-th hypothesis of a lattice (global index r + world * k)
    lat = np.array([0.3, 0.9, 0.1, 0.9, 0.5, 0.2, 0.8, 0.4])        # 0.9 twice: the lower global index (1) must win
    mine = torch.from_numpy(lat[rank::world].copy())
    t, gi = shard.best_hypothesis_t(mine, rank + world * torch.arange(len(mine), dtype=torch.int64))
    strided_ok = float(t.item()) == 0.9 and int(gi.item()) == 1
    flags = torch.tensor([1 if map_ok else 0], dtype=torch.int64)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)                 # every rank received the cloud byte for byte
    strided_ok = strided_ok and int(flags.item()) == 1
    if rank == 0:
        parts = []
        for r in range(world):
            lo, hi = shard.shard_bounds(B, world, r)
            parts.append(np.frombuffer(got[r].numpy().tobytes(), O.RESULT_DTYPE)[:hi - lo])
        allres = np.concatenate(parts)
        ref = om.align_batch(scans, off, inits)
        ok = allres.tobytes() == ref.tobytes() and best[1] == int(np.argmax(ref["trans_prob"])) and strided_ok
        open(os.path.join(outdir, "ok"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_scatter_match_gather(tmp_path, oracle, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert open(os.path.join(str(tmp_path), "ok")).read() == "1"
