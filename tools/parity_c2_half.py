"""C2's parameters at a size the oracle can still build in minutes: the product's graph (default
schedule, device traversal, device-grouped link half) against the oracle's batched restatement,
hash for hash, plus a query batch.  Usage: python tools/parity_c2_half.py [n]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
import hnswindex  # noqa: E402
import oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
cap = 65536
x = np.random.default_rng(65537).random((n, 128), dtype=np.float32)
q = np.random.default_rng(65538).random((20_000, 128), dtype=np.float32)
ix = hnswindex.Index(128)
ix.set_collection_size(n); ix.set_max_edges(16); ix.set_max_candidates(200); ix.set_min_nn(128); ix.set_insert_batch(cap); ix.set_allow_removals(False)
t = time.time(); ix.add(x); tg = time.time() - t
print(f"gpu build {tg:.2f} s ({n / tg:.0f} adds/s)", flush=True)
ref = oracle.OracleIndex(128, max_edges=16, max_candidates=200, min_nn=128, collection_size=n, allow_removals=False)
done = threading.Event()


def heartbeat():  # the oracle call is silent for minutes; the GPU runner takes silence for a hang
    while not done.wait(60):
        print(f"  oracle building ... {time.time() - t:.0f} s", flush=True)


t = time.time()
threading.Thread(target=heartbeat, daemon=True).start()
ref.add_batched(x, cap, threads=16)  # the same schedule, searches and per-list link work on 16 host threads
done.set()
tr = time.time() - t
print(f"oracle build {tr:.1f} s ({n / tr:.0f} adds/s)", flush=True)
same = ix.graph_hash() == ref.graph_hash()
got, want = ix.knn_query(q, 10), ref.knn_query(q, 10, threads=16)
same_q = bool((got[0] == want[0]).all()) and got[1].tobytes() == want[1].tobytes()
print(f"n={n}: graph {'same' if same else 'DIFFERENT'}, 20000 queries {'same' if same_q else 'DIFFERENT'}, stats {ix.stats()['search_repeats']} repeats", flush=True)
sys.exit(0 if same and same_q else 1)
