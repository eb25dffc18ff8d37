"""RangeQuery throughput: graph_range_kernel against the host lock-step path, same index, same radius.
usage: python tools/range_bench.py [n] [nq] [radius]"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import hnswindex

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 16_384
radius = float(sys.argv[3]) if len(sys.argv) > 3 else 11.0
dim = 128
x = np.random.default_rng(65537).random((n, dim), dtype=np.float32)
q = np.random.default_rng(65538).random((nq, dim), dtype=np.float32)
out = {"n": n, "nq": nq, "radius": radius}
for mode in ("device", "host"):
    ix = hnswindex.Index(dim, "sq_euclid")
    ix.set_collection_size(n); ix.set_max_edges(16); ix.set_max_candidates(200); ix.set_min_nn(128)
    ix.set_device_traversal(mode == "device")   # the build is the same either way (graph hashes are compared in the tests)
    ix.add(x)
    ix.set_profiling(True)
    ids, d = ix.range_query(q[:256], radius)   # warm
    ix.reset_stats()
    # the export itself (what a C host sees: per-query arrays allocated by the callee, HNSWIndexExports.cs:172-173) ...
    import ctypes as ct
    lib = hnswindex.net_amd.lib
    ids_pp, dists_pp, counts = (ct.c_void_p * nq)(), (ct.c_void_p * nq)(), (ct.c_int * nq)()
    dts = []
    for rep in range(4 if mode == "device" else 1):   # the first call of a size also allocates the pinned result buffers: steady state = the later ones
        t = time.perf_counter()
        rc = lib.hnsw_range_query(ix._h, q.ctypes.data_as(ct.POINTER(ct.c_float)), nq, dim, radius, ids_pp, dists_pp, counts)
        dts.append(time.perf_counter() - t)
        assert rc == 0
        lib.hnsw_free_results(ids_pp, dists_pp, nq)
    dt_export = min(dts)
    # ... and through the Python wrapper (one numpy copy per query on top)
    ix.reset_stats()
    t = time.perf_counter(); ids, d = ix.range_query(q, radius); dt = time.perf_counter() - t
    st = ix.stats()
    out[mode] = {"queries_per_sec": round(nq / dt_export, 1), "ms_per_call_each": [round(1e3 * v, 2) for v in dts], "queries_per_sec_python": round(nq / dt, 1), "results_per_query": round(sum(len(a) for a in ids) / nq, 2),
                 "evals_per_query": round(st["search_evals" if mode == "device" else "evals"] / nq, 1),
                 "handbacks": st["range_handbacks"], "kernel_ms": round(st["range_kernel_ms"], 3)}
    if mode == "device":
        keep = (ids, d)
    else:
        out["identical"] = all(a.tolist() == b.tolist() and c.tobytes() == e.tobytes() for a, b, c, e in zip(ids, keep[0], d, keep[1]))
print(json.dumps(out))
