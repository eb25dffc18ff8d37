"""Removal throughput (hnsw_remove, ids in order) against the CPU restatement, same index and ids.
usage: python tools/removal_bench.py [n] [nremove] [batch]"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from hnswindex import Index
import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dim = 128
x = np.random.default_rng(65537).random((n, dim), dtype=np.float32)
ids = np.random.default_rng(1).permutation(n)[:m].astype(np.int32)
out = {"n": n, "removed": m}
hashes = {}
for mode in (("device", "host") if n <= 200_000 else ("device",)):   # the lock-step mode builds at 5 k adds/s: small indexes only
    ix = Index(dim); ix.set_collection_size(n); ix.set_max_candidates(200); ix.set_min_nn(128)
    ix.set_device_traversal(mode == "device")
    ix.set_insert_batch(65536)               # the large snapshots (opt-in since round 5): the oracle below builds under the same cap
    ix.add(x)
    ix.remove(ids[:50])                      # warm (graph fetch, kernels)
    mm = m if n <= 200_000 else min(m, 3000)  # the sequential legs on a bounded sample
    t0 = time.perf_counter(); ix.remove(ids[50:mm]); dt = time.perf_counter() - t0
    out[mode] = {"removals_per_sec": round((mm - 50) / dt, 1), "removed": mm}
    print(json.dumps(out), flush=True)
    hashes[mode] = ix.graph_hash()
    if mode == "device":
        q = np.random.default_rng(7).random((2000, dim), dtype=np.float32)
        got = ix.knn_query(q, 10)
# the batched schedule (hnsw_mi355x_set_remove_batch): removals with disjoint neighbourhoods together
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ix = Index(dim); ix.set_collection_size(n); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_remove_batch(B); ix.set_insert_batch(65536)
ix.add(x)
ix.remove(ids[:50])
ix.reset_stats()
t0 = time.perf_counter(); ix.remove(ids[50:]); dt = time.perf_counter() - t0
print(json.dumps({"device_batched_removals_per_sec": round((m - 50) / dt, 1)}), flush=True)
refb = oracle.OracleIndex(dim, "sq_euclid", max_candidates=200, min_nn=128, collection_size=n)
refb.add_batched(x, 65536, threads=16)
print("oracle index built", flush=True)
refb.remove_batched(ids[:50], B)
t0 = time.perf_counter(); refb.remove_batched(ids[50:], B); dtb = time.perf_counter() - t0
out["device_batched"] = {"batch": B, "removals_per_sec": round((m - 50) / dt, 1), "search_launches": ix.stats()["search_launches"],
                         "cpu_same_schedule_one_thread_removals_per_sec": round((m - 50) / dtb, 1), "graph_hash_equal": bool(ix.graph_hash() == refb.graph_hash())}
del ix, refb
ref = oracle.OracleIndex(dim, "sq_euclid", max_candidates=200, min_nn=128, collection_size=n)
ref.add_batched(x, 65536, threads=16)
mm = m if n <= 200_000 else min(m, 3000)
ref.remove(ids[:50])
t0 = time.perf_counter(); ref.remove(ids[50:mm]); dt = time.perf_counter() - t0
out["cpu_one_thread"] = {"removals_per_sec": round((mm - 50) / dt, 1)}
want = ref.knn_query(q, 10)
out["graph_hash_equal"] = bool(all(h == ref.graph_hash() for h in hashes.values()))
out["queries_after_removal_equal"] = bool((got[0] == want[0]).all() and got[1].tobytes() == want[1].tobytes())
print(json.dumps(out))
