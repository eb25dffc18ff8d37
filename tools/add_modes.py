#!/usr/bin/env python3
"""Add latency by batch size on a built index: B = 1 (the reference's sequential Add), B = 16 / 64
(what a Parallel.For on a T-core host can hold in flight), one big batch.  Prints one JSON line."""
import argparse
import json
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
    a = p.parse_args()
    from hnswindex import Index
    rng = np.random.default_rng(65537)
    x = rng.random((a.n, a.dim), dtype=np.float32)
    extra = np.random.default_rng(65539).random((60000, a.dim), dtype=np.float32)
    ix = Index(a.dim)
    ix.set_collection_size(a.n + 70000); ix.set_max_candidates(a.efc); ix.set_min_nn(128); ix.set_allow_removals(False)
    t0 = time.perf_counter(); ix.add(x); build = time.perf_counter() - t0
    out = {"n": a.n, "build_s": round(build, 3)}
    pos = 0
    for b, calls in ((1, 1500), (16, 400), (64, 200), (1024, 16), (16384, 1)):
        ix.reset_stats()
        t0 = time.perf_counter()
        for _ in range(calls):
            ix.add(extra[pos:pos + b]); pos += b
        dt = time.perf_counter() - t0
        st = ix.stats()
        out[f"B{b}"] = {"adds_per_s": round(b * calls / dt, 1), "ms_per_call": round(1e3 * dt / calls, 3),
                        "evals_per_add": round(st["search_evals"] / (b * calls), 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
