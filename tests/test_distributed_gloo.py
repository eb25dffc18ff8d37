"""CPU tier: the multi-GPU query-sharding path (hnswindex.net_amd/distributed.py) on two
gloo ranks.  The local searcher is the oracle here (test tier only); on the GPU box the
same function is handed `Index.knn_query`."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nq, k, out_dir, dst_rank=None):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import hnswindex
    from common import uniform
    x, q = uniform(1500, 32, 1), uniform(nq, 32, 2)
    ix = oracle.OracleIndex(32)  # every rank holds a replica, built identically
    ix.add(x)
    calls = []

    def search(qs, kk):
        calls.append(qs.shape[0])
        return ix.knn_query(qs, kk)

    ids, d = hnswindex.net_amd.distributed.knn_query_sharded(search, q, k, dst_rank=dst_rank)
    full_ids, full_d = ix.knn_query(q, k)
    if dst_rank is not None and rank != dst_rank:
        ok = ids is None and d is None  # the gathered result stays where it is on the other ranks
    else:
        ok = (ids == full_ids).all() and np.ascontiguousarray(d).tobytes() == full_d.tobytes()
    lo, hi = hnswindex.net_amd.distributed.shard_bounds(nq, world, rank)
    ok = ok and calls == ([hi - lo] if hi > lo else [])
    np.save(Path(out_dir) / f"ok_{rank}.npy", np.array([int(ok), hi - lo]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nq", [101, 100, 2, 1])
def test_sharded_query_equals_single_rank(tmp_path, nq):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, nq, 5, str(tmp_path)), nprocs=world, join=True)
    sizes = []
    for r in range(world):
        ok, m = np.load(tmp_path / f"ok_{r}.npy")
        assert ok == 1
        sizes.append(int(m))
    assert sum(sizes) == nq and max(sizes) - min(sizes) <= 1


def test_eight_ranks_like_config_4(tmp_path):
    # BASELINE config 4's partitioning: the query set over EIGHT ranks, one all-gather of the per-shard top-k (gloo stands in
    # for RCCL here; the 8-GPU run is the driver's)
    import torch.multiprocessing as mp
    world, port, nq = 8, _free_port(), 1003
    mp.spawn(_worker, args=(world, port, nq, 10, str(tmp_path)), nprocs=world, join=True)
    sizes = []
    for r in range(world):
        ok, m = np.load(tmp_path / f"ok_{r}.npy")
        assert ok == 1
        sizes.append(int(m))
    assert sum(sizes) == nq and max(sizes) - min(sizes) <= 1


def test_single_destination_rank(tmp_path):
    # the all-gather leaves the result on every rank, rank 0 copies it out
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, 64, 5, str(tmp_path), 0), nprocs=world, join=True)
    for r in range(world):
        assert np.load(tmp_path / f"ok_{r}.npy")[0] == 1


def test_shard_bounds_cover_without_overlap():
    import hnswindex
    sb = hnswindex.net_amd.distributed.shard_bounds
    for n in (0, 1, 7, 8, 100000):
        for w in (1, 2, 4, 8):
            b = [sb(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
