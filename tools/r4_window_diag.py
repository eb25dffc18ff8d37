"""Exact-window Add on the 1M index with HIP events around the launches: where a round's time goes."""
import json, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import hnswindex
N, T, W = 1_000_000, 6000, 64
x = np.random.default_rng(65537).random((N + T + 1000, 128), dtype=np.float32)
ix = hnswindex.Index(128)
ix.set_collection_size(N + T + 1000); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_allow_removals(False)
ix.add(x[:N])
ix.set_insert_batch_live(-W)
ix.add(x[N:N + 500])           # warm
for prof in (False, True):
    ix.set_profiling(prof)
    ix.reset_stats()
    s0 = ix.exact_window_stats()
    at = N + 500 + (3000 if prof else 0)
    t0 = time.time(); ix.add(x[at:at + 3000]); dt = time.time() - t0
    s1 = ix.exact_window_stats(); st = ix.stats()
    r = s1["rounds"] - s0["rounds"]
    print(json.dumps({"profiling": prof, "adds_per_s": round(3000 / dt, 1), "rounds": r, "ms_per_round": round(1e3 * dt / r, 3),
                      "searches": s1["searches"] - s0["searches"], "insert_launches": st["insert_launches"], "lat_launches": st.get("lat_launches"),
                      "insert_kernel_ms_per_launch": round(st["insert_kernel_ms"] / max(1, st["insert_timed_launches"]), 3),
                      "link_kernel_ms_per_launch": round(st["link_kernel_ms"] / max(1, st["link_timed_launches"]), 3),
                      "evals_per_search": round(st["insert_evals"] / max(1, s1["searches"] - s0["searches"]), 1)}), flush=True)
