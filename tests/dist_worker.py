"""Worker of tests/test_gpu_distributed_nccl.py::test_two_ranks_share_the_gpu_with_the_product_searcher:
one rank of a world of two (gloo), both on the same MI355X, each holding a replica built by the PRODUCT
and answering its shard of the query set through hnswindex.net_amd.distributed.knn_query_sharded."""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    out_dir = Path(sys.argv[1])
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import hnswindex
    from common import uniform
    x, q = uniform(20000, 32, 11), uniform(1001, 32, 12)
    ix = hnswindex.Index(32)
    ix.set_collection_size(20000); ix.set_max_candidates(64); ix.set_min_nn(32)
    ix.add(x)                                   # replicas only: every rank builds the same deterministic graph
    hashes = [None] * world
    dist.all_gather_object(hashes, int(ix.graph_hash()))
    calls = []

    def search(qs, k):
        calls.append(int(qs.shape[0]))
        return ix.knn_query(qs, k)
    res = {"rank": rank, "replicas_identical": len(set(hashes)) == 1}
    for dst in (None, 0):
        ids, d = hnswindex.net_amd.distributed.knn_query_sharded(search, q, 10, dst_rank=dst)
        if dst is not None and rank != dst:
            res[f"dst{dst}"] = ids is None and d is None
        else:
            full_ids, full_d = ix.knn_query(q, 10)   # the whole set on this rank alone
            res[f"dst{dst}"] = bool((ids == full_ids).all() and np.ascontiguousarray(d).tobytes() == full_d.tobytes())
    lo, hi = hnswindex.net_amd.distributed.shard_bounds(q.shape[0], world, rank)
    res["shard_calls"] = calls == [hi - lo, hi - lo]
    # build once on rank 0, broadcast the graph, import on rank 1: the replica equals the one rank 1 built itself
    rep = hnswindex.Index(32)
    rep.set_collection_size(20000); rep.set_max_candidates(64); rep.set_min_nn(32)
    hnswindex.net_amd.distributed.replicate_index(ix if rank == 0 else rep, x, 16, src=0)
    if rank != 0:
        r_ids, r_d = rep.knn_query(q[:200], 10)
        o_ids, o_d = ix.knn_query(q[:200], 10)
        extra = uniform(300, 32, 13)
        res["replicated"] = bool(rep.graph_hash() == ix.graph_hash() and (r_ids == o_ids).all() and r_d.tobytes() == o_d.tobytes()
                                 and (rep.add(extra) == ix.add(extra)).all() and rep.graph_hash() == ix.graph_hash())
    else:
        res["replicated"] = True
    res["native_lib"] = str(hnswindex.net_amd.LIB_PATH)
    (out_dir / f"rank{rank}.json").write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
