"""GPU tier: RCCL (torch.distributed backend "nccl") and the HIP library share one process and
one HIP runtime.  A single-GPU box cannot run two NCCL ranks on one card, so this is a
world_size-1 group: it exercises process-group creation, the packed all-gather of
hnswindex.net_amd.distributed on CUDA tensors and the library's own stream side by side; the
2-rank logic itself is covered on gloo in tests/test_distributed_gloo.py."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle
from common import default_cap, uniform

pytestmark = pytest.mark.gpu


def test_nccl_group_and_library_coexist():
    import torch
    import torch.distributed as dist
    import hnswindex
    assert hnswindex.net_amd.lib.hnswdev_device_count() > 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        x, q = uniform(3000, 64, 151), uniform(257, 64, 152)
        ix = hnswindex.Index(64); ix.set_collection_size(3000)
        ix.add(x)
        ids, d = ix.knn_query(q, 10)
        # the exchange step of knn_query_sharded, as it runs on every rank (world = 1 here)
        packed = np.concatenate([ids, np.ascontiguousarray(d).view(np.int32)], axis=1)
        mine = torch.from_numpy(packed).cuda()
        out = torch.empty_like(mine)
        dist.all_gather_into_tensor(out, mine)
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        got = out.cpu().numpy()
        assert (got[:, :10] == ids).all() and got[:, 10:].view(np.float32).tobytes() == d.tobytes()
        # and the library still answers correctly after collectives ran on the same device
        ids2, d2 = ix.knn_query(q, 10)
        assert (ids2 == ids).all() and d2.tobytes() == d.tobytes()
        r_ids, r_d = hnswindex.net_amd.distributed.knn_query_sharded(ix.knn_query, q, 10)
        assert (r_ids == ids).all()
        # the exchange path itself (pinned staging, cached device buffers, views of the pinned result),
        # which a world of one would otherwise skip
        for kw in (dict(), dict(dst_rank=0, copy=False)):
            for _ in range(2):
                e_ids, e_d = hnswindex.net_amd.distributed.knn_query_sharded(ix.knn_query, q, 10, _always_exchange=True, **kw)
                assert (e_ids == ids).all() and np.ascontiguousarray(e_d).tobytes() == d.tobytes()
        ref = oracle.OracleIndex(64, collection_size=3000); ref.add_batched(x, default_cap())
        assert (ref.knn_query(q, 10)[0] == ids).all()
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_the_gpu_with_the_product_searcher(tmp_path):
    # The sharded KnnQuery with the PRODUCT as every rank's local searcher: two ranks (gloo -- two NCCL ranks
    # cannot share one card) on this one GPU, started as child processes of a launcher.  Each builds its
    # replica, answers its shard, and the gathered result must equal one rank answering everything.
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "tests" / "dist_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for rank in (0, 1):
        res = json.loads((tmp_path / f"rank{rank}.json").read_text())
        assert res["replicas_identical"] and res["dstNone"] and res["dst0"] and res["shard_calls"] and res["replicated"], res
        assert res["native_lib"].endswith("HNSWIndex.Native.so")

