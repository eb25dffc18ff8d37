"""More of tests/test_gpu_fuzz.py's tie-heavy random cases (fresh seeds) against the oracle: python tools/fuzz_more.py [first_seed=5000] [count=80]"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import hnswindex  # noqa: E402
import oracle  # noqa: E402
from test_gpu_fuzz import _case  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 80
bad = 0
tot = {"search_repeats": 0, "tie_windows": 0}
for seed in range(first, first + count):
    c = _case(np.random.default_rng(seed))
    ref = oracle.OracleIndex(c["dim"], c["metric"], max_edges=c["M"], max_candidates=c["efc"], min_nn=c["ef"], collection_size=64, random_seed=c["seed"])
    if c["batch"] == 1:
        ref.add(c["x"])
    else:
        ref.add_batched(c["x"], c["batch"])
    q = np.concatenate([c["q"]] * 8)  # 960 queries: several per wave slot is not needed here, ties are what counts
    want = ref.knn_query(q, c["k"])
    ix = hnswindex.Index(c["dim"], c["metric"])
    ix.set_collection_size(64); ix.set_max_edges(c["M"]); ix.set_max_candidates(c["efc"]); ix.set_min_nn(c["ef"])
    ix.set_random_seed(c["seed"]); ix.set_insert_batch(c["batch"])
    ix.add(c["x"])
    ok_g = ix.graph_hash() == ref.graph_hash()
    got = ix.knn_query(q, c["k"])
    ok_q = bool((got[0] == want[0]).all()) and got[1].tobytes() == want[1].tobytes()
    st = ix.stats()
    for k in tot:
        tot[k] += st[k]
    if not (ok_g and ok_q):
        bad += 1
        print("MISMATCH seed", seed, {k: v for k, v in c.items() if k not in ("x", "q")}, "graph", ok_g, "queries", ok_q, flush=True)
print(f"{count} cases from seed {first}: {bad} mismatches; exact re-runs {tot['search_repeats']}, group windows closed {tot['tie_windows']}")
sys.exit(1 if bad else 0)
