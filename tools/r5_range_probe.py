"""Bring-up aid for RangeQuery's finishing kernels (csrc/dk_range_finish.h): a small index, one range call per diagnostic setting
(range_finish = 0 / 1 / 2), each checked against the oracle.  Prints as it goes (run each setting under `timeout`)."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
mode = sys.argv[1] if len(sys.argv) > 1 else "2"
os.environ["HNSW_MI355X_DIAG"] = f"range_finish={mode}"
import hnswindex, oracle
n, dim = 2000, 128
x = np.random.default_rng(111).random((n, dim), dtype=np.float32)
ix = hnswindex.Index(dim); ix.set_collection_size(n); ix.set_insert_batch(256)
ix.add(x)
ref = oracle.OracleIndex(dim, collection_size=n); ref.add_batched(x, 256)
print("built", ix.graph_hash() == ref.graph_hash(), flush=True)
for radius, nq in ((12.0, 8), (16.0, 8), (16.0, 300), (19.0, 64)):
    t = time.time()
    ids, d = ix.range_query(x[:nq], radius)
    print(f"mode {mode} radius {radius} nq {nq}: {sum(len(a) for a in ids)} results in {time.time() - t:.3f}s", ix.stats()["range_device_ordered"], ix.stats()["range_host_ordered"], flush=True)
    rids, rd = ref.range_query(x[:nq], radius, cap=n)
    ok = all(a.tolist() == c.tolist() and b.tobytes() == e.tobytes() for a, b, c, e in zip(ids, d, rids, rd))
    print("  equal to the oracle:", ok, "ties:", sum(int(len(b) > 1 and (np.diff(b) == 0).any()) for b in d), flush=True)
