"""Does splitting ONE mid-size hnsw_knn_query call into concurrent part-calls (query lanes) shorten it?  1M x 128 index,
12 500 and 25 000 queries: one call, two / four concurrent calls on halves / quarters."""
import json, sys, threading, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import hnswindex
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x = np.random.default_rng(65537).random((N, 128), dtype=np.float32)
ix = hnswindex.Index(128); ix.set_collection_size(N); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_allow_removals(False)
ix.add(x)
out = {}
for nq in (6250, 12500, 25000):
    qs = [np.random.default_rng(100 + r).random((nq, 128), dtype=np.float32) for r in range(4)]
    for parts in (1, 2, 4):
        def run(r):
            q = qs[r % 4]
            if parts == 1:
                ix.knn_query(q, 10); return
            step = nq // parts
            th = [threading.Thread(target=lambda a=a: ix.knn_query(q[a * step:(a + 1) * step], 10)) for a in range(parts)]
            for t in th: t.start()
            for t in th: t.join()
        run(0)
        t0 = time.perf_counter()
        for r in range(12): run(r)
        dt = (time.perf_counter() - t0) / 12
        out[f"{nq}q_in_{parts}"] = {"ms": round(1e3 * dt, 3), "Mq_per_s": round(nq / dt / 1e6, 3)}
print(json.dumps(out))
