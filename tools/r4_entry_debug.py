import json, os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import hnswindex
for N in (3000, 50000):
    x = np.random.default_rng(1).random((N, 64), dtype=np.float32)
    q = np.random.default_rng(2).random((5000, 64), dtype=np.float32)
    ix = hnswindex.Index(64); ix.set_collection_size(N); ix.set_min_nn(32)
    ix.add(x)
    lv = ix.levels(); top = int(lv.max()); ep = ix.entry_point
    ix.reset_stats()
    ids, d = ix.knn_query(q, 10)
    st = ix.stats()
    print(json.dumps({"N": N, "entry": os.environ.get("HNSW_MI355X_MFMA_ENTRY", "1"), "launches": st["entry_block_launches"], "search_launches": st["search_launches"], "lat": st["lat_launches"],
                      "evals_per_query": st["search_evals"] / 5000, "top": top, "ep": ep, "ep_level": int(lv[ep]), "n_top_nodes": int((lv == top).sum()),
                      "ep_edges_top": ix.edges(ep, top).tolist()}))
