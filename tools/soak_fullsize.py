"""Full-size answer soak: the product builds the index, the CPU restatement imports that graph (as bench.py's cpu_baseline leg
does) and answers the SAME large query sets on all host cores; every id and every distance bit must agree.  At these sizes a
65 536-query launch opens about a hundred group windows of equal-distance candidates, hands two dozen searches to the exact
traversal and ends on shadow traversals (device_kernels.h) -- this is the check of those rules where they matter.

usage: python tools/soak_fullsize.py [n=1000000] [dim=128] [metric=sq_euclid] [queries=262144] [per_call=65536]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import hnswindex  # noqa: E402
import oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
metric = sys.argv[3] if len(sys.argv) > 3 else "sq_euclid"
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 262_144
per_call = int(sys.argv[5]) if len(sys.argv) > 5 else 65_536
M, efc, ef, k = 16, 200, 128, 10
rng = np.random.default_rng(65537)
x = rng.random((n, dim), dtype=np.float32)
ix = hnswindex.Index(dim, metric)
ix.set_collection_size(n); ix.set_max_edges(M); ix.set_max_candidates(efc); ix.set_min_nn(ef); ix.set_allow_removals(False)
ix.set_insert_batch(65536)   # the opt-in large snapshots: a 10M build in seconds (the graph is imported into the oracle below whatever the schedule)
t = time.time(); ix.add(x); tb = time.time() - t
lv = ix.levels()
ref = oracle.OracleIndex(dim, metric, max_edges=M, max_candidates=efc, min_nn=ef, collection_size=n, allow_removals=False)
ref.import_graph(x, lv, ix.entry_point, [ix.export_edges(L, 2 * M + 2 if L == 0 else M + 2) for L in range(int(lv.max()) + 1)])
same_graph = ref.graph_hash() == ix.graph_hash()
print(f"{n} x {dim} {metric}: built in {tb:.1f} s, graph hash after import {'equal' if same_graph else 'DIFFERENT'}", flush=True)
fails = 0 if same_graph else 1
s0 = ix.stats()
done = 0
call = 0
while done < nq:
    c = min(per_call, nq - done)
    q = np.random.default_rng(70000 + call).random((c, dim), dtype=np.float32)
    t = time.time(); got = ix.knn_query(q, k); tg = time.time() - t
    t = time.time(); want = ref.knn_query(q, k, threads=16); tc = time.time() - t
    ok = bool((got[0] == want[0]).all()) and got[1].tobytes() == want[1].tobytes()
    s1 = ix.stats()
    print(f"call {call}: {c} queries {'same' if ok else 'DIFFERENT'} (gpu {c / tg / 1e3:.0f} k/s, cpu {c / tc / 1e3:.1f} k/s; exact re-runs "
          f"{s1['search_repeats'] - s0['search_repeats']}, group windows closed {s1['tie_windows'] - s0['tie_windows']}, hand-backs {s1['search_overflows'] - s0['search_overflows']})", flush=True)
    s0 = s1
    fails += not ok
    done += c
    call += 1
sys.exit(1 if fails else 0)
