#!/usr/bin/env python3
"""
bench.py -- BASELINE.json metric on MI355X: batched KnnQuery/sec (and Add/sec) on
1M x 128 float32, sq_euclid, M=16, efConstruction=200, efSearch=128, k=10, recall@10.

A "step" is one pass of the hot path over one batch of synthetic queries: every rank runs
`Index.knn_query` on its shard of the batch (nq_per_gpu queries), then one all-gather of the
per-shard top-k (RCCL) when N > 1.  The vector matrix, graph and query set are resident
before the timed region.  Timed: exactly K steps between (barrier + cuda.synchronize) pairs,
max over ranks.  One JSON line on rank 0.

    python bench.py [--gpus N --steps K --warmup W] [--n 1000000 --nq 10000 ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--n", "--index-size", dest="n", type=int, default=1_000_000, help="indexed vectors (BASELINE C2: 1M)")
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--metric", default="sq_euclid")
    p.add_argument("--nq", type=int, default=65_536, help="queries per GPU per step (one batched knn_query call)")
    p.add_argument("--small-batch", type=int, default=10_000, help="also time batches of this many queries (N=1 only; 0 = skip)")
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--max-edges", type=int, default=16)
    p.add_argument("--ef-construction", type=int, default=200)
    p.add_argument("--ef-search", type=int, default=128)
    p.add_argument("--insert-batch", type=int, default=65536, help="cap of a snapshot batch (a batch is also <= linked/16)")
    p.add_argument("--slots", type=int, default=0, help="lock-step search slots (0 = library default)")
    p.add_argument("--threads", type=int, default=0, help="host threads of the driver (0 = library default)")
    p.add_argument("--recall-queries", type=int, default=1000)
    p.add_argument("--cpu-queries", type=int, default=4000, help="bounded cpu_baseline sample (all-cores leg)")
    p.add_argument("--cpu-adds", type=int, default=3000, help="bounded cpu_baseline sample of sequential inserts")
    p.add_argument("--data", choices=["uniform", "clustered"], default="uniform",
                   help="uniform: i.i.d. U[0,1) (the reference's test data, BASELINE.md); clustered: 1000-centre Gaussian "
                        "mixture, only to show recall on data that has neighbourhood structure")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--traversal", choices=["device", "host"], default="device",
                   help="device: graph-resident search kernel (default); host: lock-step traversal on host threads")
    return p.parse_args()


def make_data(n, dim, seed, metric, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "clustered":
        centres = np.random.default_rng(4242).random((1000, dim), dtype=np.float32)
        x = centres[rng.integers(0, 1000, n)] + (0.05 * rng.standard_normal((n, dim))).astype(np.float32)
    else:
        x = rng.random((n, dim), dtype=np.float32)  # Utils.cs:35-49: uniform [0,1)
    if metric == "ucosine":
        x = (x / np.sqrt((x * x).sum(axis=1, dtype=np.float32, keepdims=True))).astype(np.float32)
    return x


def brute_force_topk(x_t, q, k, metric):
    """Exact ground truth on the GPU in float32 (measurement harness, not the product path)."""
    import torch
    qt = torch.from_numpy(q).to(x_t.device)
    out = []
    for i in range(0, qt.shape[0], 256):
        qc = qt[i:i + 256]
        if metric == "sq_euclid":
            d = (x_t * x_t).sum(1, keepdim=True) - 2.0 * (x_t @ qc.T) + (qc * qc).sum(1)[None, :]
        else:
            d = 1.0 - (x_t @ qc.T) / (x_t.norm(dim=1, keepdim=True) * qc.norm(dim=1)[None, :])
        out.append(torch.topk(d, k, dim=0, largest=False).indices.T.cpu())
    return torch.cat(out).numpy()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    # one process per GPU; BENCH_BACKEND=gloo + several ranks on one card is only a rehearsal of
    # the multi-rank code path on a single-GPU box
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)  # "nccl" is RCCL on ROCm
    import hnswindex
    from hnswindex import Index
    dmod = hnswindex.net_amd.distributed

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- setup (untimed): data, index build, resident queries ----------------
    x = make_data(a.n, a.dim, 65537, a.metric, a.data)
    ix = Index(a.dim, a.metric)
    ix.set_collection_size(a.n)            # avoid the doubling resize (GraphData.cs:98-111)
    ix.set_max_edges(a.max_edges)
    ix.set_max_candidates(a.ef_construction)
    ix.set_min_nn(a.ef_search)             # ef = max(MinNN, k)  (HNSWIndex.cs:115)
    ix.set_allow_removals(False)           # build-rate runs drop in-edge upkeep
    ix.set_device(dev_index)
    ix.set_insert_batch(a.insert_batch)
    ix.set_device_traversal(a.traversal == "device")
    if a.slots:
        ix.set_search_slots(a.slots)
    if a.threads:
        ix.set_host_threads(a.threads)
    barrier()
    t0 = time.perf_counter()
    ids = ix.add(x)
    barrier()
    build_s = time.perf_counter() - t0
    assert ids.size == a.n
    build_stats = ix.stats()

    nq_total = a.nq * world
    q_all = make_data(nq_total, a.dim, 65538, a.metric, a.data)  # queries distinct from the base vectors

    # this rank's shard of the query set is uploaded ONCE, before the timed region: the timed steps
    # start with their inputs resident in HBM (per-step PCIe traffic: the k ids + distances back)
    lo, hi = dmod.shard_bounds(nq_total, world, rank)
    ix.set_resident_queries(q_all[lo:hi])

    # one all-gather of the packed top-k leaves the full result on every GPU; rank 0 copies it to its
    # host (pinned buffer, returned as views) -- SURVEY.md 8e
    def step():
        return dmod.knn_query_sharded(lambda qs, k: ix.knn_query_resident(k), q_all, a.k, dst_rank=0, copy=False)

    def step_pcie():  # same work with the queries handed over as host buffers every step
        return dmod.knn_query_sharded(ix.knn_query, q_all, a.k, dst_rank=0, copy=False)

    for _ in range(a.warmup):
        step()
    ix.set_profiling(True)                 # HIP events around every distance-kernel launch, on its stream
    ix.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res_ids, res_d = step()
    barrier()
    dt = time.perf_counter() - t0
    if res_ids is not None:  # views of the exchange buffer: keep them past the next call
        res_ids, res_d = np.array(res_ids), np.array(res_d)
    st = ix.stats()
    ix.set_profiling(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(2):
        step_pcie()
    barrier()
    dt_pcie = (time.perf_counter() - t0) / 2
    small = None
    if world == 1 and 0 < a.small_batch < a.nq:  # the same step on a smaller batch: launch fill / tail effects
        ix.set_resident_queries(q_all[:a.small_batch])
        ix.knn_query_resident(a.k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ix.knn_query_resident(a.k)
        torch.cuda.synchronize()
        small = a.small_batch * 10 / (time.perf_counter() - t0)
    ix.set_resident_queries(q_all[lo:hi])
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---------------- rank 0: quality, roofline, CPU baseline ----------------
    nrec = min(a.recall_queries, nq_total)
    x_t = torch.from_numpy(x).cuda()
    gt = brute_force_topk(x_t, q_all[:nrec], a.k, a.metric)
    del x_t
    recall = float(np.mean([len(set(gt[i]) & set(res_ids[i])) / a.k for i in range(nrec)]))

    if a.traversal == "device":
        kname, t_evals, t_launches, k_ms = "graph_search_kernel", st["search_timed_evals"], st["search_timed_launches"], st["search_kernel_ms"]
    else:
        kname, t_evals, t_launches, k_ms = "slot_distance_kernel", st["timed_evals"], st["timed_launches"], st["kernel_ms"]
    kernel_s = k_ms / 1e3
    achieved = t_evals * st["row_bytes"] / kernel_s / 1e9 if kernel_s > 0 else 0.0
    traffic = None
    try:  # PMC traffic is collected in separate rocprofv3 passes (tools/run_profiles.sh); quote it only
        # for the very workload it was measured on
        pm = json.loads((ROOT / "profiles" / "r1_pmc_traffic.json").read_text())
        w = pm["workload"]
        if a.traversal == "device" and (w["n"], w["dim"], w["nq"], w["ef_search"], w["k"], w["max_edges"]) == \
                (a.n, a.dim, a.nq, a.ef_search, a.k, a.max_edges) and a.metric == "sq_euclid":
            traffic = round(pm["traffic_bytes_per_launch"])
    except Exception:
        pass
    roofline = {
        "bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
        "algorithmic_bytes_per_launch": round(t_evals / max(1, t_launches) * st["row_bytes"]),
        "bytes_per_eval": st["row_bytes"], "evals_per_launch": round(t_evals / max(1, t_launches), 1),
        "launches": t_launches, "avg_launch_us": round(1e3 * k_ms / max(1, t_launches), 2),
        "kernel_time_share_of_step": round(kernel_s / dt, 4),
    }

    cpu = None
    if not a.no_cpu_baseline and world == 1:
        import oracle
        cores = min(len(os.sched_getaffinity(0)), 16)
        ref = oracle.OracleIndex(a.dim, a.metric, max_edges=a.max_edges, min_nn=a.ef_search,
                                 max_candidates=a.ef_construction, collection_size=a.n + a.cpu_adds,
                                 allow_removals=False, use_avx=True)
        lv = ix.levels()
        layers = [ix.export_edges(L, 2 * a.max_edges + 2 if L == 0 else a.max_edges + 2) for L in range(int(lv.max()) + 1)]
        ref.import_graph(x, lv, ix.entry_point, layers)
        same_graph = ref.graph_hash() == ix.graph_hash()
        n1 = min(500, nq_total)
        t0 = time.perf_counter(); c_ids1, c_d1 = ref.knn_query(q_all[:n1], a.k, threads=1); t1 = time.perf_counter() - t0
        nm = min(a.cpu_queries, nq_total)
        ref.reset_n_eval()
        t0 = time.perf_counter(); c_ids, c_d = ref.knn_query(q_all[:nm], a.k, threads=cores); tm = time.perf_counter() - t0
        cpu_evals_per_query = ref.n_eval / nm
        parity_ids = bool((c_ids == res_ids[:nm]).all())
        parity_d = bool(c_d.tobytes() == np.ascontiguousarray(res_d[:nm]).tobytes())
        # Add baseline: sequential inserts of fresh vectors into the same 1M graph, one thread
        extra = make_data(a.cpu_adds, a.dim, 65539, a.metric, a.data)
        t0 = time.perf_counter(); ref.add(extra); ta = time.perf_counter() - t0
        cpu = {
            "value": round(nm / tm, 1), "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{nm} of the step's queries on the product-built {a.n}-node graph imported into the C restatement "
                      f"(oracle/, AVX2+FMA, {cores} threads = Parallel.For over queries); ids/distances compared bit for bit with the GPU run",
            "single_thread_queries_per_s": round(n1 / t1, 1),
            "single_thread_adds_per_s": round(a.cpu_adds / ta, 1),
            "add_sample": f"{a.cpu_adds} sequential HNSWIndex.Add into the {a.n}-node graph, 1 thread",
            "evals_per_query": round(cpu_evals_per_query, 1),
            "graph_hash_equal_after_import": same_graph,
            "gpu_ids_bit_exact_vs_cpu": parity_ids, "gpu_dists_bit_identical_vs_cpu": parity_d,
        }

    qps = nq_total * a.steps / dt
    shape = (a.dim, a.metric, a.max_edges, a.ef_construction)
    cfg_name = {(128, "sq_euclid", 16, 200): "C2" if a.n <= 1_000_000 else "C4-size", (768, "ucosine", 32, 400): "C3"}.get(shape, "custom")
    out = {
        "metric": "knn_queries_per_sec", "value": round(qps, 1), "unit": "queries/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if a.data == "uniform" else "synthetic (clustered)",
        "config": {
            "workload": f"{cfg_name}: {a.n}x{a.dim} f32 {a.metric}, M={a.max_edges} efConstruction={a.ef_construction} "
                        f"efSearch={a.ef_search} k={a.k}; step = batched knn_query of {a.nq} queries per GPU "
                        f"(query set sharded over ranks, one all-gather of top-k)",
            "n": a.n, "dim": a.dim, "queries_per_gpu_per_step": a.nq, "k": a.k, "max_edges": a.max_edges,
            "ef_construction": a.ef_construction, "ef_search": a.ef_search,
            "add_mode": f"snapshot-batched, cap {a.insert_batch}", "parallelism": f"query-shard x{world}, index replicated",
        },
        "recall_at_10": round(recall, 4),
        "recall_note": "exact brute-force ground truth; i.i.d. uniform data (the reference's test distribution) has no "
                       "neighbourhood structure at this size -- the CPU path returns the same ids (see cpu_baseline)",
        "pcie_inclusive_queries_per_sec": round(nq_total / dt_pcie, 1),
        "add_per_sec": round(a.n / build_s, 1), "build_seconds": round(build_s, 2),
        "build_evals": build_stats["evals"] + build_stats["search_evals"],
        "build_launches": build_stats["launches"] + build_stats["search_launches"],
        "evals_per_query": round((st["search_evals"] + st["evals"]) / (a.nq * a.steps), 1),
        "traversal": a.traversal, "search_overflows": st["search_overflows"], "search_repeats": st["search_repeats"],
        "small_batch": None if small is None else {"queries_per_batch": a.small_batch, "queries_per_sec": round(small, 1)},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
