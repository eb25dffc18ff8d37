"""Query throughput against batch size (queries per launch) on the C2 index."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import hnswindex  # noqa: E402

n, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 128
x = np.random.default_rng(65537).random((n, dim), dtype=np.float32)
ix = hnswindex.Index(dim)
ix.set_collection_size(n); ix.set_max_edges(16); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_insert_batch(65536)
t = time.time(); ix.add(x); print(f"build {n / (time.time() - t):.0f} adds/s", flush=True)
ix.set_profiling(True)
for nq in [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (10_000, 12_288, 16_384, 24_576, 32_768, 65_536, 131_072):
    q = np.random.default_rng(65538).random((nq, dim), dtype=np.float32)
    ix.set_resident_queries(q)
    ix.knn_query_resident(10)
    ix.reset_stats()
    reps = max(3, 200_000 // nq)
    t = time.time()
    for _ in range(reps):
        ix.knn_query_resident(10)
    dt = (time.time() - t) / reps
    s = ix.stats()
    kms = s["search_kernel_ms"] / max(1, s["search_timed_launches"])
    gbs = s["search_timed_evals"] * dim * 4 / (s["search_kernel_ms"] * 1e-3) / 1e9
    print(f"nq={nq:7d}  {nq / dt:10.0f} q/s  kernel {kms:7.3f} ms  {gbs:7.1f} GB/s  frac {gbs / 8000:.3f}  repeats {s['search_repeats']}", flush=True)
