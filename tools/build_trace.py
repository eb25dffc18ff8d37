"""Build the C2 index once with the host-side phase timers on (HNSW_MI355X_DIAG=trace=1)."""
import os
import sys
import time

import numpy as np

os.environ.setdefault("HNSW_MI355X_DIAG", "trace=1")
sys.path.insert(0, ".")
import hnswindex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
x = np.random.default_rng(65537).random((n, dim), dtype=np.float32)
ix = hnswindex.Index(dim)
ix.set_collection_size(n); ix.set_max_edges(16); ix.set_max_candidates(200); ix.set_min_nn(128)
ix.set_profiling(True)
t = time.time(); ix.add(x); dt = time.time() - t
s = ix.stats()
print(f"build {n / dt:.0f} adds/s ({dt:.2f} s); search+link kernels {s['search_kernel_ms']:.0f} ms over {s['search_timed_launches']} launches, "
      f"{s['search_timed_evals'] * dim * 4 / (s['search_kernel_ms'] * 1e-3) / 1e9:.0f} GB/s, repeats {s['search_repeats']}, hand-backs {s['search_overflows']}", flush=True)
del ix
