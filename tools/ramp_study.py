#!/usr/bin/env python3
"""Build time and recall@10 against the growth rule of the snapshot schedule (batch <= linked / div).
Run with HNSW_MI355X_BATCH_DIV=<div>; prints one JSON line."""
import json, os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench

def main():
    import torch
    from hnswindex import Index
    seed = int(os.environ.get("RAMP_SEED", "65537"))
    out = {"div": os.environ.get("HNSW_MI355X_BATCH_DIV", "16"), "late": os.environ.get("HNSW_MI355X_BATCH_DIV_LATE", ""), "seed": seed}
    for kind in ("uniform", "clustered"):
        x = bench.make_data(1_000_000, 128, seed, "sq_euclid", kind)
        q = bench.make_data(4000, 128, seed + 1, "sq_euclid", kind)
        ix = Index(128); ix.set_collection_size(1_000_000); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_allow_removals(False)
        torch.cuda.synchronize(); t0 = time.perf_counter(); ix.add(x); dt = time.perf_counter() - t0
        ids, _ = ix.knn_query(q, 10)
        out[kind] = {"build_s": round(dt, 3), "recall_at_10": round(bench.recall_of(x, q, 10, "sq_euclid", ids), 4)}
        del ix
    print(json.dumps(out))
main()
