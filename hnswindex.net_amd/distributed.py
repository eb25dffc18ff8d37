"""
Multi-GPU batched KnnQuery: shard the QUERY set across ranks, one exchange step.

The reference's BatchKnnQuery is a Parallel.For over independent read-only searches
(/root/reference/src/HNSWIndex/HNSWIndex.cs:129-137), so the query set shards with no
data-path dependency: every rank holds a replica of the vector matrix and graph (10M x 128
f32 = 5.1 GB, far below one MI355X's 288 GB), rank r searches queries
[r*nq/W, (r+1)*nq/W), and ONE all-gather of the per-shard top-k -- ids (int32) and
distances (float32 bit patterns) packed in a single int32 tensor -- leaves the full result
on every rank.  Over RCCL that is one ncclAllGather on xGMI (config C4: 12 500 x 10 x 8 B
= 1 MB per rank: latency-bound, not link-bound).  Each query's traversal is unchanged, so
results are bit-identical to the single-GPU run.

Sharding the DATASET (sub-index per GPU + merge) would change results versus the single
reference graph and is deliberately not done.  Add() does not shard: replicas only.
"""
from typing import Callable, Tuple

import numpy as np


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous near-equal split; the first (n % world_size) ranks take one extra."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def knn_query_sharded(search: Callable[[np.ndarray, int], Tuple[np.ndarray, np.ndarray]], queries: np.ndarray, k: int,
                      group=None, device=None) -> Tuple[np.ndarray, np.ndarray]:
    """
    search(queries_shard, k) -> (ids int32 [m,k], dists float32 [m,k]) is this rank's local
    searcher (an `Index.knn_query` bound method).  Returns the full (ids, dists) for all
    queries on every rank.  Without an initialised process group this is a plain call.
    """
    import torch
    import torch.distributed as dist

    q = np.ascontiguousarray(queries, dtype=np.float32)
    nq = q.shape[0]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return search(q, k)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(nq, world, rank)
    per = -(-nq // world)  # every rank contributes the same number of rows to the gather
    ids, d = search(q[lo:hi], k) if hi > lo else (np.empty((0, k), np.int32), np.empty((0, k), np.float32))
    packed = np.full((per, 2 * k), -1, dtype=np.int32)
    packed[: hi - lo, :k] = ids
    packed[: hi - lo, k:] = np.ascontiguousarray(d, dtype=np.float32).view(np.int32)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(packed).to(device)
    gathered = torch.empty((world * per, 2 * k), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(gathered, mine, group=group)  # RCCL ncclAllGather over xGMI
    g = gathered.cpu().numpy().reshape(world, per, 2 * k)
    out_ids = np.empty((nq, k), dtype=np.int32)
    out_d = np.empty((nq, k), dtype=np.float32)
    for r in range(world):
        rlo, rhi = shard_bounds(nq, world, r)
        out_ids[rlo:rhi] = g[r, : rhi - rlo, :k]
        out_d[rlo:rhi] = np.ascontiguousarray(g[r, : rhi - rlo, k:]).view(np.float32)
    return out_ids, out_d
