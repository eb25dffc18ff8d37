"""Round 5, verdict item 1(c): is there a 2x left in the exact window's VALIDATION?  The window links, per round, the items up to the
first one whose speculative search (on an older snapshot) is not known to equal its search on the exact graph; the product decides
that with read sets, dry runs and distance bounds (rules 0-4) -- 31 items per round on the 1M x 128 graph.  This
replays the same schedule on the CPU restatement with a PERFECT validator: an item's kept result counts as valid exactly when it
EQUALS what a search on the current graph returns (every layer's selection, element for element).  No rule that re-validates
instead of re-searching can link more items per round than that.
usage: python tools/window_sim3.py [N] [T] [W]   ->  one JSON line"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import oracle  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    Ws = [int(w) for w in sys.argv[3].split(",")] if len(sys.argv) > 3 else [64, 256]
    dim = 128
    x = np.random.default_rng(65537).random((N + T, dim), dtype=np.float32)
    out = {"n": N, "t": T}
    for W in Ws:
        ix = oracle.OracleIndex(dim, "sq_euclid", max_edges=16, max_candidates=200, collection_size=N + T, allow_removals=False, use_avx=True)
        t0 = time.time()
        ix.add_batched(x[:N], max_batch=65536, threads=8)
        ids = ix.alloc_only(x[N:])
        top = ix.max_layer(ix.entry_point)
        same = lambda a, b: len(a) == len(b) and all(np.array_equal(u, v) for u, v in zip(a, b))
        spec = {}
        p = rounds = searches = 0
        prefix = []
        while p < T:
            if ix.max_layer(ids[p]) > top:            # moves the entry point: alone
                ix.connect_allocated(ids[p]); p += 1; top = ix.max_layer(ix.entry_point); spec.clear(); continue
            hi = min(T, p + W)
            # round start: every item of the window whose kept result no longer equals a search on the graph as it stands searches again
            for t in range(p, hi):
                now = ix.window_search(ids[t])
                if t not in spec or not same(spec[t], now):
                    spec[t] = now; searches += 1
            rounds += 1
            t = p
            while t < hi and ix.max_layer(ids[t]) <= top:
                now = ix.window_search(ids[t]) if t > p else spec[t]
                if t > p and not same(spec[t], now):
                    break                              # the first item a perfect validator turns away ends the round
                ix.window_link(ids[t], now)
                del spec[t]
                t += 1
            prefix.append(t - p)
            p = t
        out[f"W{W}"] = {"items_per_round_perfect_validation": round(T / max(1, rounds), 2), "searches_per_item": round(searches / T, 2),
                        "rounds": rounds, "seconds": round(time.time() - t0, 1)}
        del ix
    print(json.dumps(out))


if __name__ == "__main__":
    main()
