"""Build-rate probe on C3's shape (768-d unit vectors, ucosine, M=32, efC=400) at a reduced size."""
import os
import sys
import time

import numpy as np

os.environ.setdefault("HNSW_MI355X_DIAG", "trace=1")
sys.path.insert(0, ".")
import hnswindex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
x = np.random.default_rng(65537).random((n, 768), dtype=np.float32)
x /= np.sqrt((x * x).sum(axis=1, dtype=np.float32, keepdims=True))
ix = hnswindex.Index(768, "ucosine")
ix.set_collection_size(n); ix.set_max_edges(32); ix.set_max_candidates(400); ix.set_min_nn(128); ix.set_insert_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 65536)
ix.set_profiling(True)
t = time.time(); ix.add(x); dt = time.time() - t
s = ix.stats()
print(f"build {n / dt:.0f} adds/s ({dt:.2f} s); kernels {s['search_kernel_ms']:.0f} ms, {s['search_timed_evals'] / n:.0f} evals/insert, "
      f"{s['search_timed_evals'] * 3072 / (s['search_kernel_ms'] * 1e-3) / 1e9:.0f} GB/s, hash {ix.graph_hash():016x}", flush=True)
del ix
