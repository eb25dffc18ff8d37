"""Parity soak at sizes between the unit tests and the full-size test (optional 4th argument
`clustered`): float-resolution ties are
common enough there (a few per thousand traversals) to exercise the sorted-list traversal's tie
rules against the oracle.  Usage: python tools/soak.py [n] [dim] [nq] [uniform|clustered] [metric]
(metric: sq_euclid (default), cosine, ucosine, sq_euclid_i8; with the third parameter set's beam of 300 candidates and
dim >= 256 the cosine family and sq_euclid go through the MFMA-prefiltered heuristic)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import hnswindex  # noqa: E402
import oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
clustered = len(sys.argv) > 4 and sys.argv[4] == "clustered"  # Gaussian mixture: the heuristic rejects far more than on uniform data
metric = sys.argv[5] if len(sys.argv) > 5 else "sq_euclid"
CAP = int(__import__('os').environ.get('SOAK_CAP', 16384))  # Add's snapshot cap, product and oracle alike (256: what the library defaults to on a 256-thread host)
fails = 0
for seed, (M, efc, ef, k) in enumerate([(16, 200, 128, 10), (8, 100, 64, 5), (24, 300, 256, 20)]):
    rng = np.random.default_rng(900 + seed)
    if clustered:
        centres = rng.random((200, dim), dtype=np.float32)
        x = (centres[rng.integers(0, 200, n)] + 0.05 * rng.standard_normal((n, dim), dtype=np.float32)).astype(np.float32)
        q = (centres[rng.integers(0, 200, nq)] + 0.05 * rng.standard_normal((nq, dim), dtype=np.float32)).astype(np.float32)
    else:
        x = rng.random((n, dim), dtype=np.float32)
        q = rng.random((nq, dim), dtype=np.float32)
    if metric == "ucosine":
        x = (x / np.sqrt((x * x).sum(axis=1, dtype=np.float32, keepdims=True))).astype(np.float32)
        q = (q / np.sqrt((q * q).sum(axis=1, dtype=np.float32, keepdims=True))).astype(np.float32)
    ix = hnswindex.Index(dim, metric)
    ix.set_collection_size(n); ix.set_max_edges(M); ix.set_max_candidates(efc); ix.set_min_nn(ef)
    ix.set_insert_batch(CAP); ix.set_allow_removals(False)
    t = time.time(); ix.add(x); tb = time.time() - t
    ref = oracle.OracleIndex(dim, metric, max_edges=M, max_candidates=efc, min_nn=ef, collection_size=n, allow_removals=False)
    t = time.time(); ref.add_batched(x, CAP, threads=16); tr = time.time() - t
    same_graph = ix.graph_hash() == ref.graph_hash()
    s0 = ix.stats()
    got = ix.knn_query(q, k)
    s1 = ix.stats()
    want = ref.knn_query(q, k, threads=16)
    same_q = bool((got[0] == want[0]).all()) and got[1].tobytes() == want[1].tobytes()
    print(f"M={M} efC={efc} ef={ef} k={k}: graph {'same' if same_graph else 'DIFFERENT'} (gpu {tb:.1f}s, oracle {tr:.1f}s, "
          f"build repeats {s0['search_repeats']}), queries {'same' if same_q else 'DIFFERENT'} "
          f"(repeats {s1['search_repeats'] - s0['search_repeats']}, group windows closed {s1['tie_windows'] - s0['tie_windows']}, "
          f"hand-backs {s1['search_overflows']})", flush=True)
    fails += (not same_graph) + (not same_q)
sys.exit(1 if fails else 0)
