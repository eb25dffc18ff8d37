"""Does the MFMA entry block decide first passes, and do the answers stay the same?  (1M x 128; HNSW_MI355X_MFMA_ENTRY=0/1 per process)"""
import hashlib, json, os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import hnswindex
N = 200_000
x = np.random.default_rng(65537).random((N, 128), dtype=np.float32)
q = np.random.default_rng(65538).random((20000, 128), dtype=np.float32)
ix = hnswindex.Index(128); ix.set_collection_size(N); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_allow_removals(False)
ix.add(x)
ix.reset_stats()
ids, d = ix.knn_query(q, 10)
st = ix.stats()
print(json.dumps({"entry": os.environ.get("HNSW_MI355X_MFMA_ENTRY", "1"), "entry_block_launches": st["entry_block_launches"], "evals_per_query": st["search_evals"] / 20000,
                  "ids": hashlib.sha256(ids.tobytes()).hexdigest()[:16], "dist": hashlib.sha256(d.tobytes()).hexdigest()[:16], "top_layer": int(ix.levels().max())}))
