"""Conflict study for the reference-exact windowed Add (DESIGN.md "exact window").

Builds N nodes with the CPU restatement (threaded snapshot schedule: fast, same kind of graph), then adds T more
items ONE AT A TIME (HNSWIndex.Add, HNSWIndex.cs:55-65) while logging which adjacency lists each insert's searches
read and which lists its link step writes.  From the log it replays the window schedule: W consecutive items search
one snapshot, the valid prefix links in order, an item is valid iff none of the lists it read was written by an
item at or after its snapshot; invalid items search again at the next snapshot, valid ones keep their result.
Reports items linked per round (= per dependent search launch) for several W.

usage: python tools/window_sim.py [N] [T] [dim] [M] [efC]
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import oracle  # noqa: E402


def simulate(reads, writes, W, soft=None):
    """reads/writes: per item, arrays of list keys.  Returns (rounds, searches).  soft: per item, the lists whose
    prune left the same set: they count as writes inside the round that links them (outcome unknown then) but not
    afterwards."""
    n = len(reads)
    last_mod = {}
    snap = np.full(n, -1, dtype=np.int64)
    f = rounds = searches = 0
    while f < n:
        hi = min(n, f + W)
        for j in range(f, hi):
            if snap[j] < 0:
                snap[j] = f; searches += 1
            else:
                s = snap[j]
                if any(last_mod.get(k, -1) >= s for k in reads[j]):
                    snap[j] = f; searches += 1
        rounds += 1
        j = f
        while j < hi:
            s = snap[j]
            if j > f and any(last_mod.get(k, -1) >= s for k in reads[j]):
                break
            for k in writes[j]:
                last_mod[k] = j
            if soft is not None:
                for k in soft[j]:
                    last_mod[k] = j
            j += 1
        if soft is not None:  # the link results are in: lists left unchanged no longer count
            for t in range(f, j):
                for k in soft[t]:
                    if last_mod.get(k) == t:
                        del last_mod[k]
        f = j
    return rounds, searches


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    dim = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    M = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    efc = int(sys.argv[5]) if len(sys.argv) > 5 else 200
    rng = np.random.default_rng(65537)
    x = rng.random((N + T, dim), dtype=np.float32)
    ix = oracle.OracleIndex(dim, "sq_euclid", max_edges=M, max_candidates=efc, collection_size=N + T, allow_removals=False)
    t0 = time.time()
    ix.add_batched(x[:N], max_batch=4096, threads=8)
    t_build = time.time() - t0
    ix.access_log(T * 2000)
    t0 = time.time()
    ix.add(x[N:])
    t_seq = time.time() - t0
    kind, layer, node = ix.access_log_fetch()
    key = (layer.astype(np.int64) << 32) | node
    starts = np.flatnonzero(kind == 2)
    bounds = list(starts) + [kind.size]
    reads, writes, soft = [], [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        k, kk = kind[a + 1:b], key[a + 1:b]
        reads.append(np.unique(kk[k == 0]).tolist())
        writes.append(np.unique(kk[k == 1]).tolist())
        soft.append(np.unique(kk[k == 3]).tolist())
    out = {"n": N, "t": len(reads), "dim": dim, "M": M, "efC": efc, "build_s": round(t_build, 1),
           "cpu_sequential_adds_per_s": round(T / t_seq, 1),
           "reads_per_item": float(np.mean([len(r) for r in reads])), "writes_per_item": float(np.mean([len(w) for w in writes])),
           "unchanged_set_prunes_per_item": float(np.mean([len(w) for w in soft])),
           "windows": {}, "windows_all_writes": {}, "windows_true_changes_only": {}}
    allw = [sorted(set(a) | set(b)) for a, b in zip(writes, soft)]
    for W in (8, 16, 32, 64):
        r, s = simulate(reads, writes, W, soft)
        out["windows"][str(W)] = {"items_per_round": round(len(reads) / r, 2), "searches_per_item": round(s / len(reads), 2)}
        r, s = simulate(reads, writes, W)
        out["windows_true_changes_only"][str(W)] = {"items_per_round": round(len(reads) / r, 2), "searches_per_item": round(s / len(reads), 2)}
        r, s = simulate(reads, allw, W)
        out["windows_all_writes"][str(W)] = {"items_per_round": round(len(reads) / r, 2), "searches_per_item": round(s / len(reads), 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
