#!/usr/bin/env python3
"""
bench.py -- BASELINE.json metric on MI355X: batched KnnQuery/sec and Add/sec on
1M x 128 float32, sq_euclid, M=16, efConstruction=200, efSearch=128, k=10, with recall@10.

A "step" is one pass of the hot path over one batch of synthetic queries THROUGH THE REFERENCE'S OWN EXPORT
`hnsw_knn_query` (host buffers in, host buffers out, a different query set every step -- the sets rotate):
every rank runs it on its shard of the batch, then one all-gather of the per-shard top-k (RCCL) when N > 1.
Vector matrix and graph are resident in HBM before the timed region; the queries are not (that is the boundary
a drop-in caller sees).  The same step with the query set already resident in HBM is reported beside it
(`resident_queries_per_sec`).  Timed: exactly K steps between (barrier + cuda.synchronize) pairs, max over ranks.
One JSON line on rank 0.  `--sharding native`: ONE process, the library itself shards the call over N GPUs.

The Add half of the metric (all on the same 1M index):
  add_per_sec            the build: hnsw_add of the whole set under the library's DEFAULT schedule -- snapshot batches capped at the
                         host's hardware threads T, an interleaving the reference's Parallel.For (HNSWIndex.cs:70-78) can produce on
                         a host with >= T threads
  add_modes.sequential   B = 1: the reference's HNSWIndex.Add(item), one call per item
  add_modes.exact_window the SAME graph as B = 1 (the only Add whose graph the reference defines) through
                         speculative windows with read-set validation, one call for the whole sample
  add_modes.bounded      the ladder B = 16 / 64 / 256 / 1024: each rung legal for a Parallel.For host with >= B threads
  add_modes.batched      one snapshot of 32 768 items on the built index (opt-in: this build's own schedule, rounds 1-4's default)
each beside the CPU restatement running the SAME schedule on the same vectors (graph hashes compared),
plus `roofline_add` for the two build kernels.

    python bench.py [--gpus N --steps K --warmup W] [--n 1000000 --nq 65536 ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PROFILE_ROUND = "r5"
PROFILE_FALLBACK = "r4"  # ceilings measured last round stay quoted until this round's passes are committed (traffic: only from a profile of THIS build, see below)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--n", "--index-size", dest="n", type=int, default=1_000_000, help="indexed vectors (BASELINE C2: 1M)")
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--metric", default="sq_euclid")
    p.add_argument("--nq", type=int, default=65_536, help="queries per step: per GPU (--scaling weak) or in total (--scaling strong)")
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                   help="weak: --nq queries per GPU per step; strong: --nq queries per step shared by all GPUs (C4: 100000 over 8)")
    p.add_argument("--small-batch", type=int, default=12_500, help="also time calls of this many queries (C4's per-GPU shard; N=1 only; 0 = skip)")
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--max-edges", type=int, default=16)
    p.add_argument("--ef-construction", type=int, default=200)
    p.add_argument("--ef-search", type=int, default=128)
    p.add_argument("--insert-batch", type=int, default=0, help="cap of Add's snapshot batches for the build (a batch is also <= linked/16): 0 = the library's default, "
                   "the host's hardware threads -- the items a Parallel.For on this host holds in flight; 65536 = the opt-in large snapshots of rounds 1-4")
    p.add_argument("--slots", type=int, default=0, help="lock-step search slots (0 = library default)")
    p.add_argument("--threads", type=int, default=0, help="host threads of the driver (0 = library default)")
    p.add_argument("--recall-queries", type=int, default=1000)
    p.add_argument("--cpu-queries", type=int, default=4000, help="bounded cpu_baseline sample (all-cores leg)")
    p.add_argument("--seq-adds", type=int, default=2000, help="sample of sequential (B=1) inserts into the built index, GPU and CPU")
    p.add_argument("--window-adds", type=int, default=8000, help="sample inserted through the exact window (same graph as B=1), GPU; the CPU adds them one at a time")
    p.add_argument("--window", type=int, default=256, help="cap of the speculative window of the exact-window Add (the live window is twice the recent prefix)")
    p.add_argument("--query-sets", type=int, default=4, help="distinct query sets the timed steps rotate through (fresh host buffers every step)")
    p.add_argument("--sharding", choices=["ranks", "native"], default="ranks",
                   help="N > 1: ranks = one process per GPU, torch.distributed over RCCL (what the driver launches); native = ONE process, "
                        "hnsw_knn_query shards over N device contexts inside the library (hnsw_mi355x_set_devices)")
    p.add_argument("--bounded-ladder", default="16,64,256,1024", help="caps B of the bounded-concurrency Add ladder (a snapshot batch of B items is legal for a "
                   "Parallel.For host with >= B threads); '' = skip")
    p.add_argument("--bounded-adds", type=int, default=2048, help="sample per rung of the ladder: max(this, 16 batches' worth), GPU and CPU")
    p.add_argument("--batched-adds", type=int, default=32768, help="sample inserted as one batch, GPU and CPU (all cores)")
    p.add_argument("--recall-study-n", type=int, default=32768, help="index size of the bounded-concurrency recall comparison (0 = skip)")
    p.add_argument("--data", choices=["uniform", "clustered"], default="uniform",
                   help="uniform: i.i.d. U[0,1) (the reference's test data, BASELINE.md); clustered: 1000-centre Gaussian "
                        "mixture, only to show recall on data that has neighbourhood structure")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-clustered-check", action="store_true", help="skip the recall check on data with neighbourhood structure")
    p.add_argument("--no-process-warmup", action="store_true", help="skip the throw-away index that warms the process up (profiler runs: "
                   "its launches would be averaged into the per-kernel statistics)")
    p.add_argument("--no-add-modes", action="store_true")
    p.add_argument("--build", choices=["replicate", "broadcast"], default="replicate",
                   help="N > 1: replicate = every rank builds the same deterministic graph itself, in parallel (wall time of one build, "
                        "no communication; default); broadcast = rank 0 builds, the graph is broadcast (RCCL) and imported by the others")
    p.add_argument("--dry-run", action="store_true", help="validate the launch plan of --gpus N (launcher, rendezvous, shard bounds, host and HBM memory per rank) and "
                   "print it as one JSON line WITHOUT touching a GPU or building anything")
    p.add_argument("--traversal", choices=["device", "host"], default="device",
                   help="device: graph-resident search kernel (default); host: the literal north-star split -- traversal on "
                        "host threads, distances batched step by step through the inner C ABI (hnswdev_step_submit / _wait)")
    return p.parse_args()


def make_data(n, dim, seed, metric, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "clustered":
        centres = np.random.default_rng(4242).random((1000, dim), dtype=np.float32)
        x = centres[rng.integers(0, 1000, n)] + (0.05 * rng.standard_normal((n, dim))).astype(np.float32)
    else:
        x = np.empty((n, dim), dtype=np.float32)  # Utils.cs:35-49: uniform [0,1); filled in chunks (10M x 128 = 5 GB)
        for i in range(0, n, 1_000_000):
            x[i:i + 1_000_000] = rng.random((min(1_000_000, n - i), dim), dtype=np.float32)
    if metric == "ucosine":
        for i in range(0, n, 250_000):
            c = x[i:i + 250_000]
            x[i:i + 250_000] = (c / np.sqrt((c * c).sum(axis=1, dtype=np.float32, keepdims=True))).astype(np.float32)
    return x


def brute_force_topk(x_t, q, k, metric):
    """Exact ground truth on the GPU in float32 (measurement harness, not the product path)."""
    import torch
    qt = torch.from_numpy(q).to(x_t.device)
    out = []
    for i in range(0, qt.shape[0], 256):
        qc = qt[i:i + 256]
        if metric in ("sq_euclid", "sq_euclid_i8"):  # int8: the ground truth is the float vectors' (what quantisation costs shows in recall)
            d = (x_t * x_t).sum(1, keepdim=True) - 2.0 * (x_t @ qc.T) + (qc * qc).sum(1)[None, :]
        else:
            d = 1.0 - (x_t @ qc.T) / (x_t.norm(dim=1, keepdim=True) * qc.norm(dim=1)[None, :])
        out.append(torch.topk(d, k, dim=0, largest=False).indices.T.cpu())
    return torch.cat(out).numpy()


def recall_of(x, q, k, metric, got_ids):
    import torch
    x_t = torch.from_numpy(x).cuda()
    gt = brute_force_topk(x_t, q, k, metric)
    del x_t
    return float(np.mean([len(set(gt[i]) & set(got_ids[i])) / k for i in range(q.shape[0])]))


def quote_pmc_traffic(path, workload, fetched_row_bytes, build_id):
    """roofline.traffic from the committed PMC profile (tools/install_profiles_r5.py): quoted only for the very workload
    (n, dim, queries per GPU per step, efSearch, k, MaxEdges) and row size it was measured on, and only when those passes ran on
    the library that is running now -- every configuration carries the build id of its passes (files from before that carry one
    id for the whole set).  -> (bytes per launch | None, {insert_search, link_half}, note)"""
    note = "no PMC profile of this workload under profiles/"
    try:
        pm = json.loads(Path(path).read_text())
    except Exception:
        return None, {}, note
    for c in pm.get("configs", {}).values():
        w = c.get("workload", {})
        if workload is None or tuple(w.get(k) for k in ("n", "dim", "queries_per_gpu_per_step", "ef_search", "k", "max_edges")) != tuple(workload) \
                or c.get("row_bytes_fetched") != fetched_row_bytes:
            continue
        measured_on = c.get("build_id", pm.get("build_id"))
        if measured_on != build_id:   # counters of another build say nothing about this one's kernels: not quoted
            note = (f"profiles/{Path(path).name} holds this workload measured on build {str(measured_on)[:16]}, this library is build {build_id[:16]}: "
                    "not quoted (tools/run_profiles_r5.sh re-measures)")
            continue
        return (round(c["graph_search_kernel"]["traffic_bytes_per_launch"]), {k: c[k] for k in ("insert_search", "link_half") if k in c},
                f"FETCH_SIZE x calibration + WRITE_SIZE, separate --pmc passes on this build (profiles/{Path(path).name})")
    return None, {}, note


def cgroup_quota():
    """CPUs this process's cgroup may use (cpu.max quota / period), or None: what .NET clamps Environment.ProcessorCount to."""
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        return None


def new_index(a, dev_index, capacity, insert_batch, devices=1):
    from hnswindex import Index
    ix = Index(a.dim, a.metric)
    ix.set_devices(devices)
    ix.set_collection_size(capacity)       # avoid the doubling resize (GraphData.cs:98-111)
    ix.set_max_edges(a.max_edges)
    ix.set_max_candidates(a.ef_construction)
    ix.set_min_nn(a.ef_search)             # ef = max(MinNN, k)  (HNSWIndex.cs:115)
    ix.set_allow_removals(False)           # build-rate runs drop in-edge upkeep
    ix.set_device(dev_index)
    ix.set_insert_batch(insert_batch)
    ix.set_device_traversal(a.traversal == "device")
    if a.slots:
        ix.set_search_slots(a.slots)
    if a.threads:
        ix.set_host_threads(a.threads)
    return ix


def relaunch_for_gpus(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes BEFORE anything
    touches the GPU (one process per GPU over RCCL), and exit with the launcher's code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    raise SystemExit(subprocess.run(cmd).returncode)


def dry_run(a):
    """`bench.py --gpus N --dry-run`: everything that can be checked about an N-rank run without a GPU.  No multi-GPU measurement exists yet
    (the pool gives one GPU per box); this makes sure the first real one does not die of a launcher, bounds or memory mistake."""
    import importlib.util
    native = a.sharding == "native" and a.gpus > 1
    world = 1 if native else a.gpus
    nshards = a.gpus
    nq_total = a.nq * nshards if a.scaling == "weak" else a.nq
    problems, notes = [], []
    # shard bounds from the very function the step uses (hnswindex.net_amd/distributed.py::shard_bounds, loaded from its source text:
    # importing the package would load the HIP library)
    import typing
    src = (ROOT / "hnswindex.net_amd" / "distributed.py").read_text()
    i0 = src.index("def shard_bounds")
    ns = {"Tuple": typing.Tuple}
    exec(compile(src[i0:src.index("\n\n\n", i0)], "distributed.py::shard_bounds", "exec"), ns)
    shards = [tuple(ns["shard_bounds"](nq_total, nshards, r)) for r in range(nshards)]
    if shards[0][0] != 0 or shards[-1][1] != nq_total or any(shards[i][1] != shards[i + 1][0] for i in range(nshards - 1)):
        problems.append("shards do not tile the query set")
    if a.scaling == "weak" and any(hi - lo != a.nq for lo, hi in shards):
        problems.append("weak scaling: a rank's shard differs from the N = 1 workload")
    if min(hi - lo for lo, hi in shards) <= 0:
        problems.append("a rank has no queries")
    # launcher and rendezvous
    if importlib.util.find_spec("torch.distributed.run") is None:
        problems.append("torch.distributed.run not importable")
    try:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    except OSError as e:
        problems.append(f"cannot bind a rendezvous port on 127.0.0.1: {e}"); port = None
    if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
        notes.append("HSA_ENABLE_IPC_MODE_LEGACY is not 0: RCCL needs dmabuf IPC on this pool")
    gpus_visible = None
    try:
        import torch
        gpus_visible = torch.cuda.device_count()   # counting devices does not initialise the GPU
        if gpus_visible < a.gpus:
            notes.append(f"{gpus_visible} GPUs visible here, the plan needs {a.gpus}: it can only be validated, not run, on this machine")
    except Exception as e:  # noqa: BLE001
        notes.append(f"torch not importable here ({type(e).__name__})")
    # memory per rank: every rank generates the base vectors and ALL query sets on the host (deterministic seeds: no broadcast needed)
    i8 = a.metric == "sq_euclid_i8"
    row_b = (128 if a.dim <= 120 else ((a.dim + 8 + 63) // 64) * 64) if i8 else a.dim * 4
    R = max(1, a.query_sets)
    host = {"base_vectors_f32": a.n * a.dim * 4, "query_sets": R * nq_total * a.dim * 4,
            "exchange_pinned": 2 * nq_total * a.k * 8, "result_arrays": 2 * nq_total * a.k * 8,
            "library_pinned_staging": 64 << 20, "host_graph_copy": a.n * (2 * a.max_edges + 2) * 4 + a.n * 24}
    per_gpu_q = max(hi - lo for lo, hi in shards)
    hbm = {"rows": a.n * row_b, "graph_mirror_adj0": a.n * (2 * a.max_edges + 2) * 4, "upper_layers_levels_prefix": a.n * 16 + (a.n // 16 + 1) * (a.max_edges + 2) * 4 * 2,
           "resident_queries": per_gpu_q * row_b, "results_jobs": per_gpu_q * (a.k * 8 + 64),
           "per_wave_scratch": 256 * 20 * (8192 * 8 + (64 << 10) + (0 if a.n > 4_000_000 else (a.n + 7) // 8)),
           "broadcast_staging": (a.n * (2 * a.max_edges + 2) * 4 if a.build == "broadcast" and world > 1 else 0)}
    host_rank = sum(host.values())
    procs = world
    mem_total = None
    try:
        mem_total = int([l for l in open("/proc/meminfo") if l.startswith("MemTotal")][0].split()[1]) * 1024
        cg = Path("/sys/fs/cgroup/memory.max").read_text().strip()
        if cg != "max":
            mem_total = min(mem_total, int(cg))
    except Exception:  # noqa: BLE001
        pass
    if mem_total and host_rank * procs > 0.8 * mem_total:
        problems.append(f"host memory: {procs} ranks x {host_rank / 2**30:.1f} GiB exceeds 80 % of the {mem_total / 2**30:.0f} GiB this process may use")
    if sum(hbm.values()) > 0.9 * 288e9:
        problems.append("HBM: a replica does not fit one MI355X (288 GB)")
    cmd = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
            "bench.py"] + [x for x in sys.argv[1:] if x != "--dry-run"]) if not native else [sys.executable, "bench.py"] + [x for x in sys.argv[1:] if x != "--dry-run"]
    print(json.dumps({
        "dry_run": True, "ok": not problems, "problems": problems, "notes": notes,
        "plan": {"gpus": a.gpus, "processes": procs, "sharding": "native (one process, n device contexts)" if native else "ranks (one process per GPU, torch.distributed nccl = RCCL)",
                 "scaling": a.scaling, "queries_per_step_total": nq_total, "shards": shards, "collective_per_step": None if native else
                 {"op": "all_gather_into_tensor", "bytes_per_rank": per_gpu_q * a.k * 8, "bytes_gathered": nq_total * a.k * 8},
                 "build": a.build if world > 1 else "one build", "replica_check": "all_gather of the graph hash (replicas_identical)",
                 "launch": " ".join(cmd), "children_started_before_any_gpu_call": True, "gpus_visible_here": gpus_visible},
        "memory": {"host_bytes_per_rank": host, "host_GiB_per_rank": round(host_rank / 2**30, 2), "host_GiB_all_ranks": round(host_rank * procs / 2**30, 2),
                   "host_GiB_available": round(mem_total / 2**30, 1) if mem_total else None,
                   "hbm_bytes_per_gpu": hbm, "hbm_GiB_per_gpu": round(sum(hbm.values()) / 2**30, 2)},
        "measured_on_more_than_one_gpu": False,
        "note": "no N > 1 measurement exists: every multi-GPU path of this repository has run as ranks / contexts sharing ONE GPU only"}))
    raise SystemExit(0 if not problems else 1)


def main():
    a = parse()
    if a.dry_run:
        dry_run(a)
    native = a.sharding == "native" and a.gpus > 1
    if "WORLD_SIZE" not in os.environ and a.gpus > 1 and not native:
        relaunch_for_gpus(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev_native = a.gpus if native else 1
    if native and world != 1:
        raise SystemExit("bench.py: --sharding native is one process (no launcher): python bench.py --gpus N --sharding native")
    if world != a.gpus and not native:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...)")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    # one process per GPU; BENCH_BACKEND=gloo + several ranks on one card is only a rehearsal of
    # the multi-rank code path on a single-GPU box
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if native and torch.cuda.device_count() < a.gpus and os.environ.get("BENCH_ALLOW_SHARED_GPUS") != "1":
        raise SystemExit(f"bench.py: --sharding native --gpus {a.gpus} but {torch.cuda.device_count()} GPUs visible "
                         "(BENCH_ALLOW_SHARED_GPUS=1 lets contexts share a GPU: a rehearsal, not a measurement)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": torch.device("cuda", dev_index)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)  # "nccl" is RCCL on ROCm
    import hnswindex
    dmod = hnswindex.net_amd.distributed

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- setup (untimed): data, index build, resident queries ----------------
    x = make_data(a.n, a.dim, 65537, a.metric, a.data)
    ladder = [int(b) for b in a.bounded_ladder.split(",") if b.strip()] if not a.no_add_modes else []
    ladder_items = [max(a.bounded_adds, 16 * b) for b in ladder]   # per rung: at least 16 batches
    extra_total = a.seq_adds + a.window_adds + sum(ladder_items) + a.batched_adds
    # process warm-up (untimed): a throw-away index loads the library's code objects and creates the HIP
    # context once -- 0.15 s on the first launch of every kernel family, which is not Add throughput
    if not a.no_process_warmup:
        warm = new_index(a, dev_index, 4096, a.insert_batch)
        warm.add(x[:4096])
        warm.knn_query(x[:64], a.k)
        del warm
    ix = new_index(a, dev_index, a.n + extra_total, a.insert_batch, ndev_native)
    ix.set_profiling(True)                 # HIP events around the build kernels too (roofline_add)
    barrier()
    t0 = time.perf_counter()
    if world > 1 and a.build == "broadcast":
        if rank == 0:
            ids = ix.add(x)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0      # rank 0's build; the others wait for the graph
        dmod.replicate_index(ix, x, a.max_edges, src=0)
        barrier()
        if rank != 0:
            ids = np.arange(a.n, dtype=np.int32)
    else:
        ids = ix.add(x)
        barrier()
        build_s = time.perf_counter() - t0
    assert ids.size == a.n
    build_stats = ix.stats()
    ix.set_profiling(False)
    my_hash = ix.graph_hash()
    replicas_identical = True
    if world > 1:  # every rank holds the same deterministic graph (replicas only): prove it
        h63 = torch.tensor([int(my_hash) & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        hs = torch.empty(world, dtype=torch.int64, device=h63.device)
        dist.all_gather_into_tensor(hs, h63)
        replicas_identical = bool((hs == hs[0]).all().item())

    ranks_seen = 1
    if world > 1:  # every rank must be reachable through the collective backend the step uses (RCCL when backend == "nccl")
        ones = torch.ones(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        assert ranks_seen == world, f"{ranks_seen} ranks answered the all-reduce, WORLD_SIZE is {world}"
    nshards = world * ndev_native
    nq_total = a.nq * nshards if a.scaling == "weak" else a.nq
    # R distinct query sets (all distinct from the base vectors): every timed step hands the export fresh host buffers
    R = max(1, a.query_sets)
    q_sets = [make_data(nq_total, a.dim, 65538 + 1000 * r, a.metric, a.data) for r in range(R)]
    q_all = q_sets[0]
    lo, hi = dmod.shard_bounds(nq_total, world, rank)
    per_gpu = (hi - lo) // ndev_native
    if a.scaling == "weak":  # weak scaling keeps the N = 1 (BENCH) workload on every GPU: same index, same queries per GPU and step
        assert per_gpu == a.nq, f"rank {rank}: {per_gpu} queries per GPU and step, the N = 1 configuration has {a.nq}" 

    # THE STEP: the reference's export hnsw_knn_query (HNSWIndexExports.cs:119-149) on host buffers; with several ranks one
    # all-gather of the packed top-k leaves the full result on every GPU and rank 0 copies it to its host -- SURVEY.md 8e
    def step(s):
        return dmod.knn_query_sharded(ix.knn_query, q_sets[s % R], a.k, dst_rank=0, copy=False)

    def step_resident():  # the same work with this rank's shard of set 0 uploaded once, before the timed region
        return dmod.knn_query_sharded(lambda qs, k: ix.knn_query_resident(k), q_all, a.k, dst_rank=0, copy=False)

    for s_ in range(a.warmup):
        step(s_)
    ix.set_profiling(True)                 # HIP events around every traversal launch, on its stream
    ix.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for s_ in range(a.steps):
        res_ids, res_d = step(a.warmup + s_)
    barrier()
    dt = time.perf_counter() - t0
    last_set = (a.warmup + a.steps - 1) % R
    if res_ids is not None:  # views of the exchange buffer: keep them past the next call
        res_ids, res_d = np.array(res_ids), np.array(res_d)
    st = ix.stats()
    st_all = [ix.stats_at(g) for g in range(ndev_native)] if native else [st]
    ix.set_profiling(False)
    ix.set_resident_queries(q_all[lo:hi])
    step_resident()
    barrier()
    t0 = time.perf_counter()
    nb_steps = max(2, min(5, a.steps))
    for _ in range(nb_steps):
        step_resident()
    barrier()
    dt_resident = (time.perf_counter() - t0) / nb_steps
    small = None
    if world == 1 and not native and 0 < a.small_batch < hi - lo:  # the same step on C4's per-GPU shard size: launch fill / tail effects
        ix.set_resident_queries(q_all[:a.small_batch])
        ix.knn_query_resident(a.k)
        ix.set_profiling(True)
        ix.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ix.knn_query_resident(a.k)
        torch.cuda.synchronize()
        dts = time.perf_counter() - t0
        ss = ix.stats()
        ix.set_profiling(False)
        ev_s, ms_s = (ss["search_timed_evals"], ss["search_kernel_ms"]) if a.traversal == "device" else (ss["timed_evals"], ss["kernel_ms"])
        t0 = time.perf_counter()
        for r_ in range(10):
            ix.knn_query(q_sets[r_ % R][:a.small_batch], a.k)
        dtb = time.perf_counter() - t0
        small = {"queries_per_call": a.small_batch, "queries_per_sec": round(a.small_batch * 10 / dtb, 1),
                 "resident_queries_per_sec": round(a.small_batch * 10 / dts, 1),
                 "roofline_frac": round(ev_s * ss["row_bytes"] / (ms_s / 1e3) / 1e9 / HBM_PEAK_GBPS, 4) if ms_s > 0 else None}
        # same-type calls on one handle overlap (README.md:64-65): T host threads, each issuing hnsw_knn_query calls of that
        # size back to back on its own query sets -- every call has a query lane of its own, the launches run side by side
        import threading
        for T in (2, 4):
            got = [None] * T

            def worker(t):
                for r_ in range(10):
                    got[t] = ix.knn_query(q_sets[(t + r_) % R][t * a.small_batch:(t + 1) * a.small_batch] if (t + 1) * a.small_batch <= nq_total
                                          else q_sets[(t + r_) % R][:a.small_batch], a.k)
            th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            dtt = time.perf_counter() - t0
            small[f"queries_per_sec_{T}_host_threads"] = round(T * a.small_batch * 10 / dtt, 1)
            small[f"speedup_{T}_host_threads_vs_one"] = round(T * a.small_batch * 10 / dtt / small["queries_per_sec"], 3)
    if world > 1:
        t = torch.tensor([dt, dt_resident], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_resident = float(t[0].item()), float(t[1].item())

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---------------- rank 0: quality, rooflines, CPU baseline, Add modes ----------------
    nrec = min(a.recall_queries, nq_total)
    q_last = q_sets[last_set]              # the set the last timed step answered
    recall = recall_of(x, q_last[:nrec], a.k, a.metric, res_ids)

    if a.traversal == "device":
        # native sharding: the contexts run side by side -- evaluations of all, launches and kernel time of the slowest one
        kname, t_evals, t_launches, k_ms = ("graph_search_kernel", sum(c["search_timed_evals"] for c in st_all) / len(st_all),
                                            st["search_timed_launches"], max(c["search_kernel_ms"] for c in st_all))
    else:
        kname, t_evals, t_launches, k_ms = "slot_distance_kernel", st["timed_evals"], st["timed_launches"], st["kernel_ms"]
    kernel_s = k_ms / 1e3
    achieved = t_evals * st["row_bytes"] / kernel_s / 1e9 if kernel_s > 0 else 0.0
    # PMC traffic and the measured random-gather ceilings are collected in separate passes (tools/run_profiles_r5.sh; --pmc runs
    # cannot be combined with tracing) and committed under profiles/: quoted only for the very workload they were measured on
    gather = None
    fetched_row_bytes = 128 if a.metric == "sq_euclid_i8" and a.dim <= 120 else st["row_bytes"]
    def profile_file(stem):
        for r in (PROFILE_ROUND, PROFILE_FALLBACK):
            f = ROOT / "profiles" / f"{r}_{stem}"
            if f.exists():
                return f
        return ROOT / "profiles" / f"{PROFILE_ROUND}_{stem}"
    build_id = hnswindex.net_amd.lib.hnsw_mi355x_build_id().decode()
    workload = (a.n, a.dim, per_gpu, a.ef_search, a.k, a.max_edges) if a.traversal == "device" and a.data == "uniform" else None
    traffic, traffic_add, traffic_note = quote_pmc_traffic(profile_file("pmc_traffic.json"), workload, fetched_row_bytes, build_id)
    gather_same_table = None
    try:
        gj = json.loads(profile_file("gather_ceilings.json").read_text())
        gather = gj["rows"].get(str(fetched_row_bytes))          # table >> Infinity Cache (4 GiB)
        table_bytes = a.n * fetched_row_bytes
        for e in gj.get("by_table", []):                          # uniform random rows of a table of THIS configuration's size
            if e["row_bytes"] == fetched_row_bytes and abs(e["table_bytes"] - table_bytes) <= 0.15 * table_bytes:
                gather_same_table = e
    except Exception:
        pass
    roofline = {
        "bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_note": traffic_note,
        "algorithmic_bytes_per_launch": round(t_evals / max(1, t_launches) * st["row_bytes"]),
        "bytes_per_eval": st["row_bytes"], "evals_per_launch": round(t_evals / max(1, t_launches), 1),
        "launches": t_launches, "avg_launch_us": round(1e3 * k_ms / max(1, t_launches), 2),
        "kernel_time_share_of_step": round(kernel_s / dt, 4),
    }
    # What RANDOM gathers of rows this size reach on this chip (tools/gather_bench): the honest ceiling is the one measured on a table of the
    # configuration's OWN size -- a 512-MB matrix half-lives in the 256-MiB Infinity Cache, and the counters and the kernel both see that.
    src_g = "profiles/" + profile_file("gather_ceilings.json").name
    rows_per_s = t_evals / kernel_s if kernel_s > 0 else 0.0
    gref = gather_same_table or gather
    if gref:
        roofline["measured_gather_ceiling"] = {"row_bytes_fetched": fetched_row_bytes, "GBps": gref["best_GBps"], "rows_per_s": round(gref["rows_per_s"]),
                                               "table": (f"{gref['table_bytes'] / 1e9:.2f} GB: this configuration's own matrix size" if gather_same_table else "4 GiB (16x the Infinity Cache)"),
                                               "source": src_g}
        roofline["frac_of_measured_gather"] = round(rows_per_s / gref["rows_per_s"], 4)
    if gather and gather_same_table:
        roofline["measured_gather_ceiling_beyond_cache"] = {"GBps": gather["best_GBps"], "rows_per_s": round(gather["rows_per_s"]), "table": "4 GiB (16x the Infinity Cache)"}
    roofline["peak_note"] = ("`frac` divides by the 8 TB/s HBM peak (the contract's denominator). " +
                             ("This matrix (%.2f GB) is small enough for the 256-MiB Infinity Cache to serve part of the rows, so `achieved` is cache-ASSISTED and can exceed what "
                              "HBM alone gives a gather; frac_of_measured_gather prices the kernel against uniform random gathers from a table of the same size. " % (a.n * fetched_row_bytes / 1e9)
                              if a.n * fetched_row_bytes < 2.0e9 else
                              "This matrix (%.2f GB) is far beyond the Infinity Cache: frac_of_measured_gather is against what random gathers of such rows reach on this chip. " % (a.n * fetched_row_bytes / 1e9)) +
                             "The figure that transfers to large indices is the C4-size one (profiles/).")
    # the Add half: graph_insert_search_kernel (search half + heuristic) and the link half, HIP events during the build
    bs = build_stats
    rb = bs["row_bytes"]
    ins_s, lnk_s = bs["insert_kernel_ms"] / 1e3, bs["link_kernel_ms"] / 1e3
    add_evals = bs["insert_timed_evals"] + bs["link_timed_evals"]
    roofline_add = None
    if a.traversal == "device" and ins_s > 0:
        in_kernel = add_evals * rb / (ins_s + lnk_s) / 1e9
        roofline_add = {
            "bound": "hbm", "kernels": "graph_insert_search_kernel + link half (link_plan/offsets/order + graph_link_kernel)",
            "schedule": f"the build under cap {ix.insert_batch_cap}" + ("" if ix.insert_batch_cap >= 4096 else ": launches of a few hundred traversals that do not fill the chip -- bound by the dependent chain "
                        "of ONE traversal (latency variants, DESIGN.md 3.6), not by bytes; the opt-in large snapshots' kernels are priced in add_modes.batched.roofline"),
            "achieved": round(in_kernel, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(in_kernel / HBM_PEAK_GBPS, 4),
            "traffic": (round(sum(v["traffic_bytes_total"] for v in traffic_add.values())) if len(traffic_add) == 2 else None),
            "traffic_over_algorithmic": ({k: round(v["traffic_over_algorithmic"], 3) for k, v in traffic_add.items()} or None),
            "bytes_per_eval": rb, "evals": add_evals,
            "insert_search": {"launches": bs["insert_timed_launches"], "seconds": round(ins_s, 4), "evals": bs["insert_timed_evals"],
                              "frac": round(bs["insert_timed_evals"] * rb / ins_s / 1e9 / HBM_PEAK_GBPS, 4)},
            "link_half": {"launches": bs["link_timed_launches"], "seconds": round(lnk_s, 4), "evals": bs["link_timed_evals"],
                          "frac": round(bs["link_timed_evals"] * rb / max(lnk_s, 1e-9) / 1e9 / HBM_PEAK_GBPS, 4)},
            "end_to_end_frac": round(add_evals * rb / build_s / 1e9 / HBM_PEAK_GBPS, 4),
        }

    # threads of the CPU legs: this process's CPU share -- the cgroup quota where there is one (the GPU box: 16 of the host's 256
    # hardware threads), else the affinity mask -- capped at 32; both host figures are printed beside it (cpu_baseline.host)
    import math
    quota = cgroup_quota()
    cores = max(1, min(len(os.sched_getaffinity(0)), math.ceil(quota) if quota else 1 << 30, 32))
    rng_x = np.random.default_rng(65539)
    extra = rng_x.random((max(extra_total, 1), a.dim), dtype=np.float32)
    if a.metric == "ucosine":
        extra = (extra / np.sqrt((extra * extra).sum(axis=1, dtype=np.float32, keepdims=True))).astype(np.float32)
    o1 = a.seq_adds; o2 = o1 + a.window_adds; o3 = o2 + sum(ladder_items)
    e_seq, e_win, e_bat = extra[:o1], extra[o1:o2], extra[o3:extra_total]
    e_rung, _o = {}, o2   # bounded concurrency: a snapshot batch of B items = what a Parallel.For over T >= B threads can hold in flight (HNSWIndex.cs:70-78)
    for b_, m_ in zip(ladder, ladder_items):
        e_rung[b_] = extra[_o:_o + m_]; _o += m_
    host_threads = hnswindex.net_amd.host_parallelism()

    cpu = None
    cpu_add = {}
    cpu_batched_evals = 0
    if not a.no_cpu_baseline and world == 1:
        import oracle
        ref = oracle.OracleIndex(a.dim, a.metric, max_edges=a.max_edges, min_nn=a.ef_search,
                                 max_candidates=a.ef_construction, collection_size=a.n + extra_total,
                                 allow_removals=False, use_avx=True)
        lv = ix.levels()
        layers = [ix.export_edges(L, 2 * a.max_edges + 2 if L == 0 else a.max_edges + 2) for L in range(int(lv.max()) + 1)]
        ref.import_graph(x, lv, ix.entry_point, layers)
        del layers
        ref.rng_skip(a.n)  # the product's level generator has drawn once per node
        same_graph = ref.graph_hash() == my_hash
        n1 = min(500, nq_total)
        t0 = time.perf_counter(); c_ids1, c_d1 = ref.knn_query(q_last[:n1], a.k, threads=1); t1 = time.perf_counter() - t0
        nm = min(a.cpu_queries, nq_total)
        ref.reset_n_eval()
        t0 = time.perf_counter(); c_ids, c_d = ref.knn_query(q_last[:nm], a.k, threads=cores); tm = time.perf_counter() - t0
        cpu_evals_per_query = ref.n_eval / nm
        parity_ids = bool((c_ids == res_ids[:nm]).all())
        parity_d = bool(c_d.tobytes() == np.ascontiguousarray(res_d[:nm]).tobytes())
        cpu = {
            "value": round(nm / tm, 1), "unit": "queries/s", "cores": cores, "kind": "port",
            "host": {"hardware_threads_visible": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": quota, "threads_used": cores},
            "sample": f"{nm} of the step's queries on the product-built {a.n}-node graph imported into the C restatement "
                      f"(oracle/, AVX2+FMA, {cores} threads = Parallel.For over queries); ids/distances compared bit for bit with the GPU run",
            "single_thread_queries_per_s": round(n1 / t1, 1),
            "evals_per_query": round(cpu_evals_per_query, 1),
            "graph_hash_equal_after_import": same_graph,
            "gpu_ids_bit_exact_vs_cpu": parity_ids, "gpu_dists_bit_identical_vs_cpu": parity_d,
        }
        if not a.no_add_modes:
            # the three Add schedules on the CPU, same vectors and same order as the GPU legs below
            t0 = time.perf_counter(); ref.add(e_seq); cpu_add["sequential"] = (a.seq_adds / (time.perf_counter() - t0), 1, ref.graph_hash())
            if a.window_adds:  # the exact window builds the SEQUENTIAL graph: the CPU's same schedule is one item after the other
                t0 = time.perf_counter(); ref.add(e_win); cpu_add["exact_window"] = (a.window_adds / (time.perf_counter() - t0), 1, ref.graph_hash())
            for b_ in ladder:  # one add_batched call = the library's own batch loop under cap B (batches of exactly B on a graph this size)
                t0 = time.perf_counter(); ref.add_batched(e_rung[b_], b_, threads=cores)
                cpu_add[f"B{b_}"] = (e_rung[b_].shape[0] / (time.perf_counter() - t0), cores, ref.graph_hash())
            ref.reset_n_eval()
            t0 = time.perf_counter(); ref.add_batched(e_bat, 1 << 20, threads=cores)
            cpu_add["batched"] = (a.batched_adds / (time.perf_counter() - t0), cores, ref.graph_hash())
            cpu_batched_evals = ref.n_eval      # distance evaluations the reference's algorithm performs for this batch
            cpu["single_thread_adds_per_s"] = round(cpu_add["sequential"][0], 1)
        del ref

    # The traversal kernels run without a visited set: they count every row they MEASURE, a few per cent more than
    # the reference's evaluations (it skips neighbours it has seen).  The roofline's numerator is the ALGORITHMIC work -- the
    # reference's evaluations, counted by the CPU restatement on the sample it answered -- whenever that count is at hand.
    # Two numerators, each under keys that mean the same thing in every run: *_rows_measured = what the device counted (always at hand),
    # *_algorithmic = the reference's evaluations (None without the CPU leg).  `achieved` / `frac` / `evals_per_launch` are the
    # ALGORITHMIC figures whenever they are known and say so in `achieved_basis`.
    roofline["rows_measured_per_launch"] = roofline["evals_per_launch"]
    roofline["achieved_rows_measured"], roofline["frac_rows_measured"] = roofline["achieved"], roofline["frac"]
    roofline["achieved_algorithmic"] = roofline["frac_algorithmic"] = roofline["algorithmic_evals_per_launch"] = None
    roofline["achieved_basis"] = "rows measured (device counter): no CPU leg in this run"
    if cpu is not None and kernel_s > 0 and a.traversal == "device":
        alg = min(t_evals, cpu["evals_per_query"] * per_gpu * ndev_native * max(1, t_launches) / max(1, ndev_native))
        achieved = alg * st["row_bytes"] / kernel_s / 1e9
        roofline.update({"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "achieved_algorithmic": round(achieved, 1), "frac_algorithmic": round(achieved / HBM_PEAK_GBPS, 4),
                         "algorithmic_bytes_per_launch": round(alg / max(1, t_launches) * st["row_bytes"]),
                         "evals_per_launch": round(alg / max(1, t_launches), 1), "algorithmic_evals_per_launch": round(alg / max(1, t_launches), 1),
                         "achieved_basis": "reference evaluations per query (CPU restatement, cpu_baseline.evals_per_query) x queries per launch"})
        if gref:
            roofline["frac_of_measured_gather"] = round(alg / kernel_s / gref["rows_per_s"], 4)

    add_modes = None
    if not a.no_add_modes and world == 1:
        def leg(name, vecs, call, note):
            ix.reset_stats()
            t0 = time.perf_counter()
            for i in range(0, vecs.shape[0], call):
                ix.add(vecs[i:i + call])
            dtl = time.perf_counter() - t0
            h = ix.graph_hash()
            d = {"schedule": note, "inserts": int(vecs.shape[0]), "adds_per_sec": round(vecs.shape[0] / dtl, 1),
                 "ms_per_call": round(1e3 * dtl / max(1, -(-vecs.shape[0] // call)), 3)}
            if name in cpu_add:
                r, c, hh = cpu_add[name]
                d.update({"cpu_adds_per_sec": round(r, 1), "cpu_threads": c, "graph_hash_equal_to_cpu_same_schedule": bool(hh == h)})
            if name == "batched" and cpu_batched_evals and a.traversal == "device":
                stl = ix.stats()
                fetched = stl["insert_evals"] + stl["link_evals"]
                d["rows_fetched"] = int(fetched)                    # what roofline_add counts: candidate rows the kernels read
                d["reference_evaluations"] = int(cpu_batched_evals)  # Distance() calls of the reference's loops for the same batch, same graph
                d["reference_evaluations_per_row_fetched"] = round(cpu_batched_evals / max(1, fetched), 3)
            return d
        def window_leg():
            ix.set_insert_batch_live(-a.window)
            w0 = ix.exact_window_stats()
            d = leg("exact_window", e_win, max(1, a.window_adds),
                    f"the graph of B=1 (HNSWIndex.Add(item) per item, HNSWIndex.cs:55-65) through speculative windows of {a.window} items: searches on "
                    "one snapshot with read-set validation, valid prefix linked in order, the rest searched again (DESIGN.md 4.2)")
            w1 = ix.exact_window_stats()
            ix.set_insert_batch_live(a.insert_batch)
            rounds, searches = w1["rounds"] - w0["rounds"], w1["searches"] - w0["searches"]
            d.update({"window": a.window, "rounds": rounds, "items_per_round": round(a.window_adds / max(1, rounds), 2),
                      "ms_per_round": round(1e3 * a.window_adds / d["adds_per_sec"] / max(1, rounds), 3),
                      "searches_per_item": round(searches / max(1, a.window_adds), 3),
                      "replay_rate": round(max(0, searches - a.window_adds) / max(1, a.window_adds), 3),
                      "items_inserted_alone": w1["alone"] - w0["alone"]})
            if "cpu_adds_per_sec" in d:
                d["speedup_vs_one_cpu_core_same_graph"] = round(d["adds_per_sec"] / d["cpu_adds_per_sec"], 2)
            return d
        def batched_leg():
            ix.set_insert_batch_live(65536)
            ix.set_profiling(True)          # HIP events around the two Add kernels of this one large snapshot: the opt-in schedule's roofline
            d = leg("batched", e_bat, max(1, a.batched_adds), f"one snapshot batch of {a.batched_adds} into the built index (opt-in cap 65536: this build's own schedule, "
                    "not an interleaving any real host's Parallel.For produces; checked against its CPU restatement only)")
            sb = ix.stats()
            ix.set_profiling(False)
            ix.set_insert_batch_live(a.insert_batch)
            i_s, l_s = sb["insert_kernel_ms"] / 1e3, sb["link_kernel_ms"] / 1e3
            if a.traversal == "device" and i_s > 0:
                d["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS, "bytes_per_eval": sb["row_bytes"],
                                 "insert_search": {"seconds": round(i_s, 5), "evals": sb["insert_timed_evals"], "frac": round(sb["insert_timed_evals"] * sb["row_bytes"] / i_s / 1e9 / HBM_PEAK_GBPS, 4)},
                                 "link_half": {"seconds": round(l_s, 5), "evals": sb["link_timed_evals"], "frac": round(sb["link_timed_evals"] * sb["row_bytes"] / max(l_s, 1e-9) / 1e9 / HBM_PEAK_GBPS, 4)},
                                 "frac": round((sb["insert_timed_evals"] + sb["link_timed_evals"]) * sb["row_bytes"] / (i_s + l_s) / 1e9 / HBM_PEAK_GBPS, 4),
                                 "note": "rows the two kernels read (device-counted) x row bytes / HIP-event kernel time, one 32 768-item snapshot on the built index"}
            return d
        def bounded_ladder():
            # one hnsw_add call per rung under cap B: consecutive snapshot batches of B items (the library's own loop), CPU restatement
            # of the same schedule beside it.  A rung is an interleaving HNSWIndex.Add(List)'s Parallel.For can produce iff B <= its threads.
            out_l = {"host_hardware_threads": host_threads, "cgroup_cpu_quota": cgroup_quota(), "default_cap": host_threads,
                     "note": "B items search one snapshot, then link in id order: legal for a Parallel.For host with >= B threads (HNSWIndex.cs:70-78). The library's "
                             "default cap is this host's hardware threads; .NET additionally clamps its thread count to a cgroup CPU quota where one is set"}
            for b_ in ladder:
                ix.set_insert_batch_live(b_)
                d = leg(f"B{b_}", e_rung[b_], e_rung[b_].shape[0], f"B={b_}: snapshot batches of {b_}; legal for a Parallel.For host with >= {b_} threads")
                d["ms_per_batch"] = round(1e3 * e_rung[b_].shape[0] / d["adds_per_sec"] / max(1, e_rung[b_].shape[0] // b_), 3)
                d["legal_on_this_host"] = bool(b_ <= host_threads)
                d.pop("ms_per_call", None)
                out_l[f"B{b_}"] = d
            ix.set_insert_batch_live(a.insert_batch)
            return out_l
        add_modes = {
            "sequential": leg("sequential", e_seq, 1, "B=1: HNSWIndex.Add(item) one at a time (HNSWIndex.cs:55-65), the reference-exact mode"),
            **({"exact_window": window_leg()} if a.window_adds and a.traversal == "device" else {}),
            "bounded": bounded_ladder(),
            "batched": batched_leg(),
        }
        if a.recall_study_n and a.traversal == "device":
            # does the cap change the graph's quality?  Full builds of a smaller index under every rung, beside the SEQUENTIAL graph
            # (the only Add whose graph the reference defines, built here through exact windows) and the opt-in large snapshots
            ns = min(a.recall_study_n, a.n)
            xs, qs = x[:ns], q_all[:min(1000, nq_total)]
            rec = {}
            for label, cap in [("sequential_graph", -256)] + [(f"B{b_}", b_) for b_ in ladder] + [("snapshot_65536", 65536)]:
                sub = new_index(a, dev_index, ns, cap)
                t0 = time.perf_counter()
                sub.add(xs)
                tb = time.perf_counter() - t0
                got, _ = sub.knn_query(qs, a.k)
                rec[label] = {"recall_at_10": round(recall_of(xs, qs, a.k, a.metric, got), 4), "adds_per_sec": round(ns / tb, 1)}
                del sub
            add_modes["bounded"]["recall_study"] = {"n": ns, "queries": int(qs.shape[0]), **rec}

    # The headline data (i.i.d. uniform, the reference's own test distribution) has no neighbourhood structure at
    # this dimension, so recall@10 is low on CPU and GPU alike.  The same build and query on data that has some:
    clustered = None
    if world == 1 and a.data == "uniform" and not a.no_clustered_check and a.n <= 2_000_000 and a.traversal == "device":
        xc = make_data(a.n, a.dim, 65537, a.metric, "clustered")
        qc = make_data(4096, a.dim, 65538, a.metric, "clustered")
        ic = new_index(a, dev_index, a.n, a.insert_batch)
        t0 = time.perf_counter(); ic.add(xc); tb = time.perf_counter() - t0
        ic.knn_query(qc, a.k)
        t0 = time.perf_counter(); got, _ = ic.knn_query(qc, a.k); tq = time.perf_counter() - t0
        clustered = {"data": "1000-centre Gaussian mixture, sigma 0.05", "recall_at_10": round(recall_of(xc, qc[:1000], a.k, a.metric, got[:1000]), 4),
                     "adds_per_sec": round(a.n / tb, 1), "queries_per_sec_4096_per_call_boundary": round(4096 / tq, 1)}
        del ic, xc

    if roofline_add and add_modes and add_modes["batched"].get("reference_evaluations_per_row_fetched"):
        # `frac` above prices the rows the kernels actually read.  The reference's Add evaluates more pairs than that for
        # the same graph: the grouped heuristic measures four candidates per fetched row, the link half skips pairs a list's
        # earlier greedy pass already tested.  Ratio measured on the batched sample (same batch on the CPU restatement):
        ratio = add_modes["batched"]["reference_evaluations_per_row_fetched"]
        roofline_add["reference_work"] = {
            "reference_evaluations_per_row_fetched": ratio,
            "equivalent_GBps_at_one_row_per_evaluation": round(roofline_add["achieved"] * ratio, 1),
            "note": "the reference reads one candidate row per Distance() call; this path does the same evaluations on fewer reads (rows reused from LDS); `frac` counts reads, not evaluations"}

    # How many queries a call needs before the GPU beats the host (a single traversal is a chain of dependent steps on one
    # wavefront: tiny calls are latency-bound and lose to a CPU core)
    crossover = None
    if world == 1 and not native and cpu is not None:
        sizes, rates = [1, 4, 16, 64, 256, 1024, 4096], []
        for b in sizes:
            reps = max(3, min(40, 2048 // b))
            ix.knn_query(q_sets[0][:b], a.k)
            t0 = time.perf_counter()
            for r_ in range(reps):
                off = (r_ * b) % max(1, nq_total - b)
                ix.knn_query(q_sets[r_ % R][off:off + b], a.k)
            rates.append(round(b * reps / (time.perf_counter() - t0), 1))

        def first_above(level):
            for i, (b, r_) in enumerate(zip(sizes, rates)):
                if r_ > level:
                    if i == 0:
                        return b
                    b0, r0 = sizes[i - 1], rates[i - 1]  # log-linear between the two measured call sizes
                    f = (np.log(level) - np.log(r0)) / max(1e-9, np.log(r_) - np.log(r0))
                    return int(np.ceil(float(np.exp(np.log(b0) + f * (np.log(b) - np.log(b0))) - 1e-9)))  # the smallest call that wins
            return None
        # where a ONE-query call's time goes: HIP events around its launch (a second pass, so that the events are not in the timed one)
        ix.set_profiling(True); ix.reset_stats()
        for r_ in range(40):
            ix.knn_query(q_sets[r_ % R][r_:r_ + 1], a.k)
        s1 = ix.stats(); ix.set_profiling(False)
        k1 = s1["search_kernel_ms"] / max(1, s1["search_timed_launches"])
        one_call = {"ms_per_call": round(1e3 / rates[0], 4), "kernel_ms": round(k1, 4), "fixed_ms": round(1e3 / rates[0] - k1, 4),
                    "cpu_one_thread_ms_per_query": round(1e3 / cpu["single_thread_queries_per_s"], 4),
                    "note": "fixed = job upload, launch, one result copy, the wait (five HIP calls); the kernel is ONE traversal on two waves (latency variant): "
                            "a chain of ~150 dependent expansions, which alone is longer than the CPU's query out of its caches"}
        crossover = {"queries_per_call": sizes, "gpu_queries_per_sec": rates, "one_query_call": one_call,
                     "cpu_one_thread_queries_per_sec": cpu["single_thread_queries_per_s"], f"cpu_{cores}_threads_queries_per_sec": cpu["value"],
                     "crossover_vs_one_cpu_thread": first_above(cpu["single_thread_queries_per_s"]),
                     f"crossover_vs_{cores}_cpu_threads": first_above(cpu["value"]),
                     "note": "hnsw_knn_query calls of that many queries, fresh host buffers each; below the crossover a call is faster on the host"}

    qps = nq_total * a.steps / dt
    shape = (a.dim, a.metric, a.max_edges, a.ef_construction)
    cfg_name = {(128, "sq_euclid", 16, 200): "C2" if a.n <= 1_000_000 else "C4-size", (768, "ucosine", 32, 400): "C3",
                (96, "sq_euclid_i8", 16, 200): "C5-size"}.get(shape, "custom")
    n_gpus = world * ndev_native
    out = {
        "metric": "knn_queries_per_sec", "value": round(qps, 1), "unit": "queries/s", "n_gpus": n_gpus,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
        "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "i8 (int32 dot, f64 epilogue)" if a.metric == "sq_euclid_i8" else "f32", "data": "synthetic" if a.data == "uniform" else "synthetic (clustered)",
        "config": {
            "workload": f"{cfg_name}: {a.n}x{a.dim} {'int8+scale' if a.metric == 'sq_euclid_i8' else 'f32'} {a.metric}, M={a.max_edges} efConstruction={a.ef_construction} "
                        f"efSearch={a.ef_search} k={a.k}; step = one hnsw_knn_query call (the reference's export: host buffers in and out, a different "
                        f"query set every step) of {per_gpu} queries per GPU, rows and graph resident in HBM "
                        + ("(one process, the library shards the call over its device contexts)" if native else "(query set sharded over ranks, one all-gather of top-k)"),
            "n": a.n, "dim": a.dim, "queries_per_gpu_per_step": per_gpu, "queries_per_step": nq_total, "k": a.k, "max_edges": a.max_edges,
            "ef_construction": a.ef_construction, "ef_search": a.ef_search, "query_sets": R,
            "add_mode": f"snapshot-batched, cap {ix.insert_batch_cap}" + (" (the library default: this host's hardware threads)" if a.insert_batch == 0 else " (set by --insert-batch)"),
            "parallelism": (f"native: one process, query-shard x{ndev_native} device contexts, replicas copied device to device" if native else
                            f"query-shard x{world}, index replicated ({a.build if world > 1 else 'one build'})"),
        },
        "entry_point": "hnsw_knn_query", "build_id": build_id,
        "collective_backend": (("rccl (torch.distributed nccl)" if backend == "nccl" else backend) if world > 1 else None), "rccl_ranks_seen": ranks_seen,
        "per_gpu_workload_equals_n1": bool(a.scaling == "weak"),
        "resident_queries_per_sec": round(nq_total / dt_resident, 1),
        "resident_note": "the same step with the query set uploaded once before the timed region (hnsw_mi355x_knn_query_resident); "
                         "`value` is the reference's export, PCIe-inclusive",
        "recall_at_10": round(recall, 4),
        "recall_note": "exact brute-force ground truth; i.i.d. uniform data (the reference's test distribution) has no "
                       "neighbourhood structure at this size -- the CPU path returns the same ids (see cpu_baseline); "
                       "recall_on_clustered_data is the same build and query on data that has structure",
        "recall_on_clustered_data": clustered,
        "add_per_sec": round(a.n / build_s, 1), "build_seconds": round(build_s, 3),
        "add_schedule": {"cap": ix.insert_batch_cap, "default": a.insert_batch == 0, "host_hardware_threads": hnswindex.net_amd.host_parallelism(), "cgroup_cpu_quota": cgroup_quota(),
                         "inside_reference_outcome_set_for_hosts_with_threads_at_least": ix.insert_batch_cap},
        "add_schedule_in_reference_outcome_set": True, "add_schedule_reference_defined": False,
        "add_per_sec_reference_graph": (add_modes or {}).get("exact_window", {}).get("adds_per_sec"),
        "add_note": "add_per_sec: hnsw_add of the whole set in one call under the cap named in add_schedule -- by default the host's hardware threads T: snapshot batches of "
                    "at most T items (T searches on one snapshot, then T links in id order) are an interleaving HNSWIndex.Add(List)'s Parallel.For (HNSWIndex.cs:70-78) can produce "
                    "on a host with >= T threads, so an unchanged caller gets a graph from the reference's outcome set; checked bit for bit against the CPU restatement of the same "
                    "schedule (add_modes.bounded, every rung). The reference DEFINES a graph only for HNSWIndex.Add(item) one item after the other (HNSWIndex.cs:55-65): "
                    "add_per_sec_reference_graph builds that one through exact windows (add_modes.exact_window), add_modes.sequential is the same graph one call per item. "
                    "add_modes.batched is the opt-in 65536-item snapshot of rounds 1-4 (hnsw_mi355x_set_insert_batch(65536)): ~10x the rate, outside any real host's outcome set",
        "tie_order_exposure": {
            "inserts_answered_by_the_exact_traversal": int(build_stats.get("insert_tie_reruns", 0)), "of_inserts": a.n,
            "share": round(build_stats.get("insert_tie_reruns", 0) / max(1, a.n), 6),
            "queries_answered_by_the_exact_traversal_per_step": round(sum(c["search_repeats"] for c in st_all) / max(1, a.steps), 2), "of_queries_per_step": per_gpu * ndev_native,
            "note": "searches whose outcome rests on the order of EQUAL keys -- Span.Sort's (Heuristic.cs:22), the heaps' layout -- BCL behaviour the oracle and the product "
                    "restate from memory: the part of 'parity unpinned' that real .NET fixtures would pin (tools/dotnet_fixture, case c2_shape_20k)"},
        "build_evals": build_stats["evals"] + build_stats["search_evals"],
        "build_launches": build_stats["launches"] + build_stats["search_launches"],
        "replicas_identical": replicas_identical,
        "evals_per_query": round(sum(c["search_evals"] + c["evals"] for c in st_all) / max(1, per_gpu * ndev_native * a.steps), 1),
        "traversal": a.traversal, "search_overflows": sum(c["search_overflows"] for c in st_all), "search_repeats": sum(c["search_repeats"] for c in st_all),
        "search_tie_windows": sum(c.get("tie_windows", 0) for c in st_all),
        "search_kernel_form": "lean (launches without visited sets: DESIGN.md 3.5)" if sum(c.get("lean_launches", 0) for c in st_all) > 0 else "plain",
        "small_batch": small, "crossover_batch_vs_cpu": crossover,
        "roofline": roofline, "roofline_add": roofline_add, "add_modes": add_modes, "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
