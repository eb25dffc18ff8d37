"""Exact-window Add on a built index: adds/s, items per round, graph-hash equality with the CPU's sequential Add.

usage: python tools/exact_window_bench.py [N] [T] [W,W,...] [dim] [--no-cpu]
Builds N x dim with the default schedule, imports the graph into the CPU restatement, then for every W continues a
COPY of ... (one index: the windows run back to back on the same growing index, T items each; the CPU adds the same
items one at a time and the hashes are compared after every leg)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    N = int(args[0]) if len(args) > 0 else 1_000_000
    T = int(args[1]) if len(args) > 1 else 4000
    Ws = [int(w) for w in (args[2] if len(args) > 2 else "16,32,64,128").split(",")]
    dim = int(args[3]) if len(args) > 3 else 128
    cpu = "--no-cpu" not in sys.argv
    import hnswindex
    legs = len(Ws) + 1
    x = np.random.default_rng(65537).random((N + legs * T, dim), dtype=np.float32)
    ix = hnswindex.Index(dim)
    ix.set_collection_size(N + legs * T); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_allow_removals(False); ix.set_insert_batch(65536)
    t0 = time.time(); ix.add(x[:N]); t_build = time.time() - t0
    ref = None
    if cpu:
        import oracle
        ref = oracle.OracleIndex(dim, max_edges=16, max_candidates=200, min_nn=128, collection_size=N + legs * T, allow_removals=False)
        lv = ix.levels()
        ref.import_graph(x[:N], lv, ix.entry_point, [ix.export_edges(l, 34) for l in range(int(lv.max()) + 1)])
        ref.rng_skip(N)
        assert ref.graph_hash() == ix.graph_hash()
    out = {"n": N, "t": T, "dim": dim, "build_s": round(t_build, 2), "legs": []}
    at = N
    # leg 0: B = 1 (one item per call) for reference
    ix.set_insert_batch_live(1)
    t0 = time.time()
    for i in range(min(T, 1000)):
        ix.add(x[at + i:at + i + 1])
    dt = time.time() - t0
    leg = {"mode": "B=1", "adds_per_s": round(min(T, 1000) / dt, 1)}
    if ref is not None:
        t0 = time.time(); ref.add(x[at:at + min(T, 1000)]); leg["cpu_sequential_adds_per_s"] = round(min(T, 1000) / (time.time() - t0), 1)
        leg["hash_equal"] = ref.graph_hash() == ix.graph_hash()
    out["legs"].append(leg)
    at += min(T, 1000)
    for W in Ws:
        ix.set_insert_batch_live(-W)
        s0 = ix.exact_window_stats()
        t0 = time.time(); ix.add(x[at:at + T]); dt = time.time() - t0
        s1 = ix.exact_window_stats()
        d = {k: s1[k] - s0[k] for k in s1}
        leg = {"mode": f"window {W}", "adds_per_s": round(T / dt, 1), "items_per_round": round(T / max(1, d["rounds"]), 2),
               "searches_per_item": round(d["searches"] / T, 2), "alone": d["alone"], "ms_per_round": round(1e3 * dt / max(1, d["rounds"]), 3)}
        if ref is not None:
            t0 = time.time(); ref.add(x[at:at + T]); leg["cpu_sequential_adds_per_s"] = round(T / (time.time() - t0), 1)
            leg["hash_equal"] = ref.graph_hash() == ix.graph_hash()
        out["legs"].append(leg)
        at += T
        print(json.dumps(leg), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
