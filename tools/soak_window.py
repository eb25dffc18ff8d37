"""Exact-window soak: mid-size builds through the windows against the oracle's SEQUENTIAL Add (graph hash, levels) on data of every
kind and metric -- sizes where float-resolution ties, multi-layer items and hub conflicts all occur in numbers.
Usage: python tools/soak_window.py [n=40000]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import hnswindex  # noqa: E402
import oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000
fails = 0
cases = [("sq_euclid", 32, "uniform", 16, 200, 64), ("sq_euclid", 128, "clustered", 16, 200, 32), ("cosine", 64, "uniform", 8, 100, 128),
         ("ucosine", 96, "uniform", 24, 300, 48), ("sq_euclid_i8", 96, "uniform", 16, 200, 64), ("sq_euclid", 16, "grid", 12, 80, 24)]
for seed, (metric, dim, kind, M, efc, W) in enumerate(cases):
    rng = np.random.default_rng(4100 + seed)
    if kind == "clustered":
        centres = rng.random((200, dim), dtype=np.float32)
        x = (centres[rng.integers(0, 200, n)] + 0.05 * rng.standard_normal((n, dim), dtype=np.float32)).astype(np.float32)
    elif kind == "grid":  # few distinct coordinates: equal distances everywhere
        x = rng.integers(0, 4, size=(n // 4, dim)).astype(np.float32)
    else:
        x = rng.random((n, dim), dtype=np.float32)
    if metric == "ucosine":
        x = (x / np.sqrt((x * x).sum(axis=1, dtype=np.float32, keepdims=True))).astype(np.float32)
    m = x.shape[0]
    ix = hnswindex.Index(dim, metric)
    ix.set_collection_size(m); ix.set_max_edges(M); ix.set_max_candidates(efc); ix.set_insert_batch(-W); ix.set_allow_removals(seed % 2 == 0)
    t = time.time(); ix.add(x); tb = time.time() - t
    ref = oracle.OracleIndex(dim, metric, max_edges=M, max_candidates=efc, collection_size=m, allow_removals=seed % 2 == 0)
    t = time.time(); ref.add(x); tr = time.time() - t
    same = ix.graph_hash() == ref.graph_hash() and (ix.levels() == ref.levels()).all()
    st = ix.exact_window_stats()
    print(f"{metric} {dim}d {kind} n={m} M={M} efC={efc} W={W}: graph {'same' if same else 'DIFFERENT'} (gpu {tb:.1f}s = {m / tb:.0f} adds/s, "
          f"oracle sequential {tr:.1f}s = {m / tr:.0f} adds/s; {st['linked'] / max(1, st['rounds']):.1f} items per round, "
          f"{st['searches'] / max(1, st['linked']):.2f} searches per item, {st['alone']} alone)", flush=True)
    fails += not same
sys.exit(1 if fails else 0)
