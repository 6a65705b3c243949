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
    """Rank `src` holds the batch (numpy arrays); every rank returns its own shard as numpy arrays.
    Point-to-point sends of exactly the shard bytes (on xGMI the root fans out over its links)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = [(int(offsets[shard_bounds(len(inits), world, r)[0]]),
                    int(offsets[shard_bounds(len(inits), world, r)[1]]),
                    shard_bounds(len(inits), world, r)) for r in range(world)]
    dist.broadcast_object_list(meta, src=src)
    p0, p1, (lo, hi) = meta[0][rank]
    nb = hi - lo
    if rank == src:
        reqs = []
        for r in range(world):
            if r == src:
                continue
            q0, q1, (l, h) = meta[0][r]
            pay = [torch.from_numpy(np.ascontiguousarray(scans[q0:q1])).to(device),
                   torch.from_numpy((offsets[l:h + 1] - offsets[l]).astype(np.int64)).to(device),
                   torch.from_numpy(np.ascontiguousarray(inits[l:h])).to(device)]
            reqs += [dist.isend(t, dst=r) for t in pay]
        for q in reqs:
            q.wait()
        return shard_batch(scans, offsets, inits, world, rank)
    t_sc = torch.empty((p1 - p0, 2), dtype=torch.float32, device=device)
    t_of = torch.empty(nb + 1, dtype=torch.int64, device=device)
    t_in = torch.empty((nb, 3), dtype=torch.float64, device=device)
    for t in (t_sc, t_of, t_in):
        dist.recv(t, src=src)
    return t_sc.cpu().numpy(), t_of.cpu().numpy().astype(np.uint64), t_in.cpu().numpy()


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


def best_hypothesis(scores, first_index, device="cpu"):
    """Arg-max of a score over all ranks' seeds: each rank passes its local scores and the global
    index of its first seed; returns (best score, global index) on every rank."""
    local = int(np.argmax(scores))
    t = torch.tensor([float(scores[local])], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    best = float(t.item())
    cand = first_index + local if float(scores[local]) == best else np.iinfo(np.int64).max
    i = torch.tensor([cand], dtype=torch.int64, device=device)
    dist.all_reduce(i, op=dist.ReduceOp.MIN)       # ties: lowest global index
    return best, int(i.item())
