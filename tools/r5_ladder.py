#!/usr/bin/env python3
"""Round 5: Add under BOUNDED concurrency.  A snapshot batch of B items is a legal interleaving of the reference's
Parallel.For (HNSWIndex.cs:70-78) iff B <= the host's threads; this prints adds/s of one hnsw_add call under caps
B = 16 ... 4096 on a built index (the library's own batch loop), and of calls of B items each.  One JSON line."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--n", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--efc", type=int, default=200)
    p.add_argument("--metric", default="sq_euclid")
    p.add_argument("--caps", default="16,64,256,1024,4096")
    p.add_argument("--per-cap", type=int, default=0, help="items per cap (0: 48 batches' worth, at least 2048)")
    p.add_argument("--build-cap", type=int, default=65536)
    a = p.parse_args()
    from hnswindex import Index
    caps = [int(c) for c in a.caps.split(",")]
    rng = np.random.default_rng(65537)
    x = rng.random((a.n, a.dim), dtype=np.float32)
    per = [a.per_cap or max(2048, 48 * b) for b in caps]
    extra = np.random.default_rng(65539).random((2 * sum(per), a.dim), dtype=np.float32)
    ix = Index(a.dim, a.metric)
    ix.set_collection_size(a.n + extra.shape[0] + 16); ix.set_max_candidates(a.efc); ix.set_min_nn(128); ix.set_allow_removals(False)
    ix.set_insert_batch(a.build_cap)
    t0 = time.perf_counter(); ix.add(x); build = time.perf_counter() - t0
    out = {"n": a.n, "build_cap": a.build_cap, "build_s": round(build, 3), "build_adds_per_s": round(a.n / build, 1),
           "host_threads_affinity": len(os.sched_getaffinity(0)), "cpu_count": os.cpu_count()}
    pos = 0
    ix.set_profiling(True)
    for b, m in zip(caps, per):
        ix.set_insert_batch_live(b)
        ix.reset_stats()
        t0 = time.perf_counter(); ix.add(extra[pos:pos + m]); dt = time.perf_counter() - t0; pos += m
        st = ix.stats()
        d = {"one_call_adds_per_s": round(m / dt, 1), "ms_per_batch": round(1e3 * dt / (m / b), 3), "items": m,
             "insert_kernel_ms_per_batch": round(st.get("insert_kernel_ms", 0) / (m / b), 3), "link_kernel_ms_per_batch": round(st.get("link_kernel_ms", 0) / (m / b), 3)}
        t0 = time.perf_counter()
        per_call = []
        for i in range(0, m, b):
            k0 = ix.stats()["insert_kernel_ms"]
            ix.add(extra[pos + i:pos + i + b])
            per_call.append(ix.stats()["insert_kernel_ms"] - k0)
        dt = time.perf_counter() - t0; pos += m
        d["calls_of_B_adds_per_s"] = round(m / dt, 1)
        d["insert_kernel_ms_per_call_sorted"] = [round(v, 2) for v in sorted(per_call)]
        out[f"B{b}"] = d
    print(json.dumps(out))


if __name__ == "__main__":
    main()
