"""Add latency by call size on the 1M index (B = 1 / 4 / 16 / 64 items per call) and the exact window."""
import json, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import hnswindex
N = 1_000_000
x = np.random.default_rng(65537).random((N + 20000, 128), dtype=np.float32)
ix = hnswindex.Index(128)
ix.set_collection_size(N + 20000); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_allow_removals(False)
ix.add(x[:N])
out = {}
at = N
for b, calls in ((1, 300), (4, 100), (16, 60), (64, 30)):
    ix.set_insert_batch_live(b)
    ix.add(x[at:at + b]); at += b
    t0 = time.time()
    for _ in range(calls):
        ix.add(x[at:at + b]); at += b
    dt = time.time() - t0
    out[f"B{b}"] = {"adds_per_s": round(b * calls / dt, 1), "ms_per_call": round(1e3 * dt / calls, 3)}
ix.set_insert_batch_live(-64)
ix.add(x[at:at + 500]); at += 500
s0 = ix.exact_window_stats(); t0 = time.time(); ix.add(x[at:at + 3000]); dt = time.time() - t0; s1 = ix.exact_window_stats()
out["window64"] = {"adds_per_s": round(3000 / dt, 1), "ms_per_round": round(1e3 * dt / (s1["rounds"] - s0["rounds"]), 3)}
print(json.dumps(out))
