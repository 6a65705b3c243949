"""Sharding of independent scan-to-map matches over the GPUs of one node.

Every match is independent given the read-only map, so the path shards by scan with no exchange
during the optimisation: rank g owns the contiguous scans [g*B/G, (g+1)*B/G).  The only
collectives are the scatter of a batch that starts on one rank, the gather of the ~200-byte
result records, and (multi-hypothesis relocalisation) an arg-max over per-seed scores.
`torch.distributed` backend "nccl" is RCCL over xGMI on the GPU box; the same code runs on "gloo"
in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(total, world, rank):
    """Contiguous, balanced partition: the first (total % world) ranks get one extra item."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(scans, offsets, inits, world, rank):
    """Slice a whole batch (numpy) down to this rank's shard, offsets rebased to 0."""
    lo, hi = shard_bounds(len(inits), world, rank)
    p0, p1 = int(offsets[lo]), int(offsets[hi])
    return scans[p0:p1], (offsets[lo:hi + 1] - offsets[lo]).astype(np.uint64), inits[lo:hi]


def scatter_batch(scans, offsets, inits, src=0, device="cpu"):
    """Rank `src` holds the batch (numpy arrays; None elsewhere); every rank returns its own shard as TENSORS on
    `device` -- (scans [n, 2] float32, offsets [b + 1] int64 rebased to 0, inits [b, 3] float64) -- ready to be
    handed to ndt_align_batch_dev by data_ptr().  The shards travel as grouped point-to-point sends of exactly the
    shard bytes (RCCL over xGMI on the GPU box: the root fans out over its links; no host round trip on the
    receiving side); gloo on CPU tensors in the tests."""
    world, rank = dist.get_world_size(), dist.get_rank()
    device = torch.device(device)
    meta = [None]
    if rank == src:
        cuts = [shard_bounds(len(inits), world, r) for r in range(world)]
        meta[0] = [(int(offsets[lo]), int(offsets[hi]), (lo, hi)) for lo, hi in cuts]
    dist.broadcast_object_list(meta, src=src)
    p0, p1, (lo, hi) = meta[0][rank]
    nb = hi - lo
    if rank == src:
        ops, keep = [], []
        for r in range(world):
            if r == src:
                continue
            q0, q1, (l, h) = meta[0][r]
            pay = [torch.from_numpy(np.ascontiguousarray(scans[q0:q1], dtype=np.float32)).to(device),
                   torch.from_numpy((np.asarray(offsets[l:h + 1]).astype(np.int64) - int(offsets[l]))).to(device),
                   torch.from_numpy(np.ascontiguousarray(inits[l:h], dtype=np.float64)).to(device)]
            keep += pay
            ops += [dist.P2POp(dist.isend, t, r) for t in pay]
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        sc, of, ini = shard_batch(scans, offsets, inits, world, rank)
        return (torch.from_numpy(np.ascontiguousarray(sc, dtype=np.float32)).to(device),
                torch.from_numpy(of.astype(np.int64)).to(device),
                torch.from_numpy(np.ascontiguousarray(ini, dtype=np.float64)).to(device))
    t_sc = torch.empty((p1 - p0, 2), dtype=torch.float32, device=device)
    t_of = torch.empty(nb + 1, dtype=torch.int64, device=device)
    t_in = torch.empty((nb, 3), dtype=torch.float64, device=device)
    for q in dist.batch_isend_irecv([dist.P2POp(dist.irecv, t, src) for t in (t_sc, t_of, t_in)]):
        q.wait()
    return t_sc, t_of, t_in


def broadcast_map(points, src=0, device="cpu", chunk_bytes=8 << 20):
    """The target cloud lives on ONE rank (the reference's local map is assembled by one PointCloudMap,
    src/PointCloudMap.cpp:119-134, and handed over at src/ScanMatcher.cpp:40): rank `src` passes it as an [n, 2] float32
    numpy array or tensor (None elsewhere); every rank returns it as a float32 tensor [n, 2] on `device`, ready for
    ndt_map_build_dev by data_ptr() -- each rank then builds its own cell table from it ("build
    redundantly from a broadcast of raw points": 8 MB for the 1M-point map, 40 MB for configs[4]'s).  One broadcast of
    the point count, then the points in pieces of at most `chunk_bytes` (collectives of bounded size: the 40 MB cloud goes
    as five 8 MB broadcasts; RCCL over xGMI on the GPU box, gloo in the CPU tests)."""
    rank = dist.get_rank()
    device = torch.device(device)
    if rank == src:
        t = points if torch.is_tensor(points) else torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32))
        if t.dim() != 2 or t.shape[1] != 2 or t.dtype != torch.float32:
            raise ValueError("expected an [n, 2] float32 cloud")
        t = t.to(device).contiguous()
        n = torch.tensor([t.shape[0]], dtype=torch.int64, device=device)
    else:
        t = None
        n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    count = int(n.item())
    if t is None:
        t = torch.empty((count, 2), dtype=torch.float32, device=device)
    flat = t.view(-1)
    step = max(2, (int(chunk_bytes) // 8) * 2)               # whole points per piece
    for a in range(0, flat.numel(), step):
        dist.broadcast(flat[a:a + step], src=src)
    return t


_GATHER_OK = True


def gather_results(res_bytes, dst=0):
    """Gather equally sized uint8 result tensors to rank `dst` (returns the list there, else None)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    out = [torch.empty_like(res_bytes) for _ in range(world)] if rank == dst else None
    global _GATHER_OK
    if _GATHER_OK:
        try:
            dist.gather(res_bytes, out, dst=dst)
            return out
        except (RuntimeError, NotImplementedError):     # a backend without gather: every rank gets all records
            _GATHER_OK = False
    full = [torch.empty_like(res_bytes) for _ in range(world)]
    dist.all_gather(full, res_bytes)
    return full if rank == dst else None


def best_hypothesis_t(scores, global_index):
    """Arg-max of a score over all ranks without leaving the device: `scores` a 1-D float64 tensor of this rank's
    hypotheses, `global_index` the int64 tensor of their global numbers (ascending), both on the device the process
    group communicates from.  Returns (best score, its global index) as 1-element tensors on every rank; ties go
    to the lowest global index.  Two all-reduces of 8 bytes."""
    if scores.numel() == 0:                          # an empty shard takes part in the collectives with -inf
        best = torch.full((), float("-inf"), dtype=scores.dtype, device=scores.device)
        k = None
    else:
        best, k = torch.max(scores, dim=0)
    t = best.reshape(1).clone()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    big = torch.full((1,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=scores.device)
    mine = global_index[k].reshape(1) if k is not None else big
    cand = torch.where(best.reshape(1) == t, mine, big)
    dist.all_reduce(cand, op=dist.ReduceOp.MIN)
    return t, cand


def best_hypothesis(scores, first_index, device="cpu"):
    """numpy front of best_hypothesis_t for a contiguous shard: local scores + the global index of its first seed;
    returns (best score, global index) on every rank."""
    sc = torch.from_numpy(np.ascontiguousarray(scores, dtype=np.float64)).to(device)
    gi = torch.arange(first_index, first_index + len(scores), dtype=torch.int64, device=device)
    t, i = best_hypothesis_t(sc, gi)
    return float(t.item()), int(i.item())
