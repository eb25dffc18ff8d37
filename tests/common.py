"""Shared helpers for the test tiers."""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def uniform(n, dim, seed):
    """Reference tests use i.i.d. uniform [0,1) float32 (src/HNSWIndex.Tests/Utils.cs:35-49)."""
    return np.random.default_rng(seed).random((n, dim), dtype=np.float32)


def normalize_f32(x):
    """Utils.Normalize (Utils.cs:23-30) in float32."""
    return (x / np.sqrt((x * x).sum(axis=1, dtype=np.float32, keepdims=True))).astype(np.float32)


def golden_cases():
    return sorted(p.stem for p in GOLDEN.glob("*.json"))


def load_golden(name):
    g = json.loads((GOLDEN / f"{name}.json").read_text())
    x = uniform(g["n"], g["dim"], g["data_seed"])
    q = uniform(g["nq"], g["dim"], g["query_seed"])
    if g["metric"] == "ucosine":
        x, q = normalize_f32(x), normalize_f32(q)
    return g, x, q


def self_recall_at_1(index, x, ids, **kw):
    """Utils.Recall with k=1 on the inserted vectors (Utils.cs:54-70)."""
    res, _ = index.knn_query(x, 1, **kw)
    return float((res[:, 0] == ids).mean())


def diag_values():
    """The test hooks currently in force: HNSW_MI355X_DIAG = "name=value,..." (csrc/diag.h, read by the library on every use)."""
    import os
    return dict(p.strip().split("=", 1) for p in os.environ.get("HNSW_MI355X_DIAG", "").split(",") if "=" in p)


def set_diag(monkeypatch, **kw):
    """Adds test hooks to HNSW_MI355X_DIAG for the rest of the test (set_diag(monkeypatch, lat=2, novis=0))."""
    cur = diag_values()
    cur.update({k: str(v) for k, v in kw.items()})
    monkeypatch.setenv("HNSW_MI355X_DIAG", ",".join(f"{k}={v}" for k, v in cur.items()))


def novis_active(stats):
    """True when the search launches behind `stats` ran without a visited set (the default; diagnostic novis=0 keeps the
    sets, =1 drops them on hash-table graphs only): the kernel then counts every row it measures, which includes the
    neighbours the reference had already seen and skips."""
    mode = diag_values().get("novis", "2")
    return mode == "2" or (mode == "1" and stats.get("visited_hash_launches", 0) > 0)


def default_cap():
    """The cap of Add's snapshot batches when nothing is set: the host's hardware threads (include/hnsw_mi355x.h,
    hnsw_mi355x_set_insert_batch) -- the oracle restates the default schedule as add_batched(x, default_cap())."""
    import hnswindex
    return hnswindex.net_amd.host_parallelism()
