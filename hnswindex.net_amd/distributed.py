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


_buffers = {}


def _buffer(key, shape, device, pinned=False):
    """Cached exchange buffers: the gather runs every step with the same shapes."""
    import torch
    t = _buffers.get(key)
    if t is None or tuple(t.shape) != tuple(shape) or t.device != torch.device(device):
        t = torch.empty(shape, dtype=torch.int32, device=device, pin_memory=pinned)
        _buffers[key] = t
    return t


def knn_query_sharded(search: Callable[[np.ndarray, int], Tuple[np.ndarray, np.ndarray]], queries: np.ndarray, k: int,
                      group=None, device=None, dst_rank=None, copy: bool = True, _always_exchange: bool = False):
    """
    search(queries_shard, k) -> (ids int32 [m,k], dists float32 [m,k]) is this rank's local
    searcher (an `Index.knn_query` bound method).  Returns the full (ids, dists) for all
    queries; without an initialised process group this is a plain call.

    dst_rank=None: every rank copies the gathered result to its host.  dst_rank=r: the gathered
    tensor stays on the device everywhere, only rank r copies it out (the others return
    (None, None)) -- SURVEY.md 8e: "full result on every rank (rank 0 copies to host)".
    copy=False returns views of a reused pinned buffer (valid until the next call).
    """
    import torch
    import torch.distributed as dist

    q = np.ascontiguousarray(queries, dtype=np.float32)
    nq = q.shape[0]
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not _always_exchange):
        return search(q, k)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(nq, world, rank)
    per = -(-nq // world)  # every rank contributes the same number of rows to the gather
    ids, d = search(q[lo:hi], k) if hi > lo else (np.empty((0, k), np.int32), np.empty((0, k), np.float32))
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    device = torch.device(device)
    on_gpu = device.type == "cuda"
    # ids and distance bit patterns side by side in one int32 tensor; staged through pinned memory
    stage = _buffer(("stage", per, k), (per, 2 * k), "cpu", pinned=on_gpu)
    st = stage.numpy()
    st[: hi - lo, :k] = ids
    st[: hi - lo, k:] = np.ascontiguousarray(d, dtype=np.float32).view(np.int32)
    if hi - lo < per:
        st[hi - lo:] = -1
    if on_gpu:
        mine = _buffer(("mine", per, k), (per, 2 * k), device)
        mine.copy_(stage, non_blocking=True)
    else:
        mine = stage
    gathered = _buffer(("gathered", world, per, k), (world * per, 2 * k), device)
    dist.all_gather_into_tensor(gathered, mine, group=group)  # RCCL ncclAllGather over xGMI
    if dst_rank is not None and rank != dst_rank:
        if on_gpu:
            torch.cuda.current_stream(device).synchronize()
        return None, None
    if on_gpu:
        out = _buffer(("out", world, per, k), (world * per, 2 * k), "cpu", pinned=True)
        out.copy_(gathered, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
        g = out.numpy()
    else:
        g = gathered.numpy()
    if nq == world * per:  # equal shards: the gathered rows are already in query order
        out_ids, out_d = g[:, :k], g[:, k:].view(np.float32)
        if copy or not on_gpu:
            out_ids, out_d = out_ids.copy(), out_d.copy()
        return out_ids, out_d
    g = g.reshape(world, per, 2 * k)
    out_ids = np.empty((nq, k), dtype=np.int32)
    out_d = np.empty((nq, k), dtype=np.float32)
    for r in range(world):
        rlo, rhi = shard_bounds(nq, world, r)
        out_ids[rlo:rhi] = g[r, : rhi - rlo, :k]
        out_d[rlo:rhi] = np.ascontiguousarray(g[r, : rhi - rlo, k:]).view(np.float32)
    return out_ids, out_d


def replicate_index(ix, rows, max_edges: int, src: int = 0, group=None, device=None, broadcast_rows: bool = False):
    """
    Build once, broadcast, import: rank `src` holds a built `Index`; on every other rank `ix` is a configured but
    still EMPTY `Index` (same setters applied, nothing added).  The graph -- levels, entry point, adjacency lists of
    every layer -- is broadcast (RCCL `ncclBroadcast` over xGMI with backend "nccl"; 150 MB at C2) and imported
    (`Index.import_graph`), after which every replica is identical to the source: same graph hash, same answers, and
    later Adds continue identically (the level generator is advanced as on the source).  `rows` are the indexed
    vectors, which every rank normally holds already (same file / same generator); with `broadcast_rows=True` only
    `src` needs them and they travel too (512 MB at C2).

    The alternative -- every rank building the same deterministic graph itself, in parallel -- costs the wall time of
    one build and no communication; this routine saves the N - 1 redundant builds instead.  Returns `ix`.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return ix
    rank = dist.get_rank(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    device = torch.device(device)

    def bcast(arr, shape, dtype):
        """arr: the numpy array on src (None elsewhere); returns it on every rank."""
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype)).to(device) if rank == src else torch.empty(shape, dtype=getattr(torch, np.dtype(dtype).name), device=device)
        dist.broadcast(t, src=src, group=group)
        return arr if rank == src else t.cpu().numpy()

    stride0, strideU = 2 * max_edges + 2, max_edges + 2
    if rank == src:
        lv = ix.levels()
        # A replica is rebuilt from levels, entry point and out-edges alone: removed slots, the slot-reuse stack and
        # the draws they cost cannot travel that way, so a source that ever removed anything (length != count) or is
        # empty is refused -- by a status word every rank reads before any payload, so that all of them raise together.
        status = 0 if (lv.size > 0 and ix.count == lv.size) else (1 if lv.size == 0 else 2)
        head = np.array([lv.size, ix.entry_point, (int(lv.max()) + 1) if lv.size else 0, ix.dim, status], dtype=np.int64)
    else:
        lv, head = None, None
    head = bcast(head, (5,), np.int64)
    n, entry, nlayers, dim, status = (int(v) for v in head)
    if status != 0:
        raise RuntimeError("replicate_index: the source index is empty" if status == 1 else
                           "replicate_index: the source index has removed slots (Length != Count); a replica built from "
                           "levels and edges would differ in Count, graph hash and later Add ids")
    lv = bcast(lv, (n,), np.int32)
    if broadcast_rows:
        rows = bcast(rows if rank == src else None, (n, dim), np.float32)
    layers = []
    for layer in range(nlayers):
        stride = stride0 if layer == 0 else strideU
        counts, edges = ix.export_edges(layer, stride) if rank == src else (None, None)
        layers.append((bcast(counts, (n,), np.int32), bcast(edges, (n, stride), np.int32)))
    if rank != src:
        ix.import_graph(np.asarray(rows)[:n], lv, entry, layers)
    return ix
