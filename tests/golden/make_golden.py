"""Generates tests/golden/*.json from the ORACLE (the reference is C# and cannot run here:
no dotnet/mono).  The fixtures pin (seed, params) -> levels, graph hash,
knn ids and distance bit patterns, so that both the oracle (regression) and the HIP path
(parity, -m gpu) are held to the same committed numbers.

    python tests/golden/make_golden.py
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402

CASES = [
    # name, n, dim, metric, params, nq, k, batch (0 = sequential Add)
    ("c1_sq_euclid_seq", 1500, 64, "sq_euclid", dict(max_edges=16, max_candidates=100, min_nn=5), 64, 10, 0),
    ("cosine_seq", 800, 128, "cosine", dict(max_edges=16, max_candidates=100, min_nn=5), 32, 10, 0),
    ("ucosine_m8_seq", 800, 96, "ucosine", dict(max_edges=8, max_candidates=40, min_nn=16), 32, 5, 0),
    ("sq_euclid_dim127_seq", 500, 127, "sq_euclid", dict(max_edges=12, max_candidates=60, min_nn=1), 32, 3, 0),
    ("c1_sq_euclid_batched", 4000, 64, "sq_euclid", dict(max_edges=16, max_candidates=100, min_nn=5), 64, 10, 256),
]


def data(n, dim, metric, seed):
    x = np.random.default_rng(seed).random((n, dim), dtype=np.float32)
    if metric == "ucosine":  # Utils.Normalize (src/HNSWIndex.Tests/Utils.cs:23-30), in float32
        x = x / np.sqrt((x * x).sum(axis=1, dtype=np.float32, keepdims=True))
    return x.astype(np.float32)


def run_case(name, n, dim, metric, params, nq, k, batch):
    x = data(n, dim, metric, 65537)
    q = data(nq, dim, metric, 65538)
    ix = oracle.OracleIndex(dim, metric, collection_size=n, random_seed=31337, use_avx=False, **params)
    ids = ix.add_batched(x, batch) if batch else ix.add(x)
    kid, kd = ix.knn_query(q, k)
    return {
        "name": name, "n": n, "dim": dim, "metric": metric, "params": params, "nq": nq, "k": k, "batch": batch,
        "data_seed": 65537, "query_seed": 65538, "random_seed": 31337,
        "ids_are_sequential": bool((ids == np.arange(n)).all()),
        "levels_head": ix.levels()[:128].tolist(),
        "level_histogram": np.bincount(ix.levels()).tolist(),
        "entry_point": ix.entry_point,
        "graph_hash": f"{ix.graph_hash():016x}",
        "knn_ids": kid.tolist(),
        "knn_dist_bits": kd.view(np.uint32).tolist(),
    }


if __name__ == "__main__":
    out = Path(__file__).resolve().parent
    for c in CASES:
        r = run_case(*c)
        (out / f"{c[0]}.json").write_text(json.dumps(r, separators=(",", ":")))
        print(c[0], r["graph_hash"], r["level_histogram"])
