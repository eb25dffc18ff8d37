"""Phase clocks of latency-bound launches (B = 1 Add, single-query calls) on the 1M index.
Needs the -DEXP_PHASE_CLOCKS build: HNSW_MI355X_LIB=build_variants/phase.so python tools/r4_phase_b1.py"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import hnswindex

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x = np.random.default_rng(65537).random((N + 4000, 128), dtype=np.float32)
q = np.random.default_rng(65538).random((4096, 128), dtype=np.float32)
ix = hnswindex.Index(128)
ix.set_collection_size(N + 4000); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_allow_removals(False)
ix.add(x[:N])
ix.reset_stats()
print("== B=1 adds", file=sys.stderr, flush=True)
ix.set_insert_batch_live(1)
t0 = time.time()
for i in range(500):
    ix.add(x[N + i:N + i + 1])
dt = time.time() - t0
print(f"B=1: {500 / dt:.1f} adds/s", file=sys.stderr, flush=True)
ix.reset_stats()
print("== single-query calls", file=sys.stderr, flush=True)
t0 = time.time()
for i in range(300):
    ix.knn_query(q[i:i + 1], 10)
dt = time.time() - t0
print(f"1 query per call: {300 / dt:.1f} q/s", file=sys.stderr, flush=True)
ix.reset_stats()
print("== 64-query calls", file=sys.stderr, flush=True)
t0 = time.time()
for i in range(30):
    ix.knn_query(q[64 * i:64 * i + 64], 10)
dt = time.time() - t0
print(f"64 queries per call: {64 * 30 / dt:.1f} q/s", file=sys.stderr, flush=True)
ix.reset_stats()
