"""Does the Add schedule cost graph quality against the reference's own sequential graph?
Builds the same n x dim set under the library's DEFAULT (snapshot batches capped at the host's hardware threads: inside the reference's
Parallel.For outcome set), under the opt-in 65 536-item snapshots of rounds 1-4, and as the SEQUENTIAL graph (HNSWIndex.Add(item) per item,
HNSWIndex.cs:55-65) through the exact window -- and compares recall@10 (exact brute-force ground truth), the mean layer-0 out-degree and
the build time.   usage: python tools/recall_study.py [n=1000000] [nq=2000] [window=256]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    dim = 128
    import torch
    import hnswindex
    out = {"n": n, "dim": dim, "queries": nq, "M": 16, "efConstruction": 200, "efSearch": 128, "k": 10, "data": {}}
    for kind in ("uniform", "clustered"):
        rng = np.random.default_rng(65537)
        if kind == "uniform":
            x = rng.random((n, dim), dtype=np.float32)
            q = np.random.default_rng(65538).random((nq, dim), dtype=np.float32)
        else:
            centres = np.random.default_rng(4242).random((1000, dim), dtype=np.float32)
            x = centres[rng.integers(0, 1000, n)] + (0.05 * rng.standard_normal((n, dim))).astype(np.float32)
            r2 = np.random.default_rng(65538)
            q = centres[r2.integers(0, 1000, nq)] + (0.05 * r2.standard_normal((nq, dim))).astype(np.float32)
        xt, qt = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
        gt = []
        for i in range(0, nq, 256):
            d = (xt * xt).sum(1, keepdim=True) - 2.0 * (xt @ qt[i:i + 256].T)
            gt.append(torch.topk(d, 10, dim=0, largest=False).indices.T.cpu())
        gt = torch.cat(gt).numpy()
        del xt, qt
        torch.cuda.empty_cache()
        res = {}
        cap = hnswindex.net_amd.host_parallelism()
        for label, batch in ((f"default_schedule_cap_{cap}", 0), ("opt_in_snapshots_65536", 65536), ("sequential_graph_exact_window", -W)):
            ix = hnswindex.Index(dim)
            ix.set_collection_size(n); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_allow_removals(False); ix.set_insert_batch(batch)
            t0 = time.time(); ix.add(x); tb = time.time() - t0
            ids, _ = ix.knn_query(q, 10)
            rec = float(np.mean([len(set(gt[i]) & set(ids[i])) / 10 for i in range(nq)]))
            cnt, _ = ix.export_edges(0, 34)
            res[label] = {"recall_at_10": round(rec, 4), "mean_out_degree_layer0": round(float(cnt.mean()), 3), "build_seconds": round(tb, 2),
                          "adds_per_sec": round(n / tb, 1), "graph_hash": f"{ix.graph_hash():016x}"}
            if batch < 0:
                st = ix.exact_window_stats()
                res[label]["items_per_round"] = round(st["linked"] / max(1, st["rounds"]), 2)
                res[label]["searches_per_item"] = round(st["searches"] / max(1, st["linked"]), 3)
            del ix
            print(kind, label, res[label], flush=True)
        out["data"][kind] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
