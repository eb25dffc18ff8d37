"""Development experiment: how long one traversal takes on an otherwise idle chip, sorted-list form against the exact
two-heap form (HNSW_MI355X_SORTED_TOP=0), per call size -- what an exact re-run costs a launch's tail."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import hnswindex

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(65537)
x = rng.random((n, 128), dtype=np.float32)
ix = hnswindex.Index(128)
ix.set_collection_size(n); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_allow_removals(False)
t = time.time(); ix.add(x); print("build", round(time.time() - t, 2), "s", flush=True)
for nq in (1, 16, 64, 256, 1024, 3072, 12500):
    q = np.random.default_rng(65538 + nq).random((nq, 128), dtype=np.float32)
    ix.knn_query(q, 10)
    reps = 20 if nq <= 1024 else 5
    t = time.time()
    for _ in range(reps): ix.knn_query(q, 10)
    dt = (time.time() - t) / reps
    print(f"nq={nq}: {dt * 1e3:.3f} ms per call, {nq / dt:.0f} q/s", flush=True)
