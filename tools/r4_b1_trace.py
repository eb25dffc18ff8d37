"""B = 1 adds with the host phase timers on (HNSW_MI355X_DIAG=trace=1) and HIP events around the kernels: where a 1.1-ms add goes."""
import json, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import hnswindex
N = 1_000_000
x = np.random.default_rng(65537).random((N + 3000, 128), dtype=np.float32)
ix = hnswindex.Index(128); ix.set_collection_size(N + 3000); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_allow_removals(False)
ix.add(x[:N])
ix.set_insert_batch_live(1)
for i in range(100): ix.add(x[N + i:N + i + 1])
for prof in (False, True):
    ix.set_profiling(prof); ix.reset_stats()
    t0 = time.time()
    for i in range(1000): ix.add(x[N + 100 + 1000 * prof + i:N + 101 + 1000 * prof + i])
    dt = time.time() - t0
    st = ix.stats()
    print(json.dumps({"profiling": prof, "adds_per_s": round(1000 / dt, 1), "ms_per_add": round(dt, 4), "insert_kernel_ms_per_add": round(st["insert_kernel_ms"] / 1000, 4),
                      "link_kernel_ms_per_add": round(st["link_kernel_ms"] / 1000, 4), "insert_launches": st["insert_launches"], "link_launches": st["link_launches"]}))
q = np.random.default_rng(3).random((2000, 128), dtype=np.float32)
for prof in (False, True):
    ix.set_profiling(prof); ix.reset_stats()
    t0 = time.time()
    for i in range(1000): ix.knn_query(q[i:i + 1], 10)
    dt = time.time() - t0
    st = ix.stats()
    print(json.dumps({"profiling": prof, "single_queries_per_s": round(1000 / dt, 1), "ms_per_call": round(dt, 4), "search_kernel_ms_per_call": round(st["search_kernel_ms"] / 1000, 4)}))
