"""Second conflict study for the exact window: how many of the TRUE list changes would a reader not notice?
(Round 5 adds rule 5, simulated only: a gained id a that WOULD be pushed -- d(j, a) < far at that expansion -- still changes nothing if it
is farther than everything the search pops afterwards and than the list's farthest entry at the last expansion: it enters the list, is
never popped and is pushed out again; whatever it makes the push test turn away meanwhile was the list's worst entry, never popped either.)
A reader j that expanded node v when its result list was full with farthest distance f is unaffected by a change of v's list that
(i) adds ids a with d(j, a) >= f (they would not have been pushed) and (ii) drops ids x with d(j, x) >= f (they were not pushed
either).  Replays the window schedule with that rule on top of "true changes only".
usage: python tools/window_sim2.py [N] [T]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import oracle  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    dim = 128
    x = np.random.default_rng(65537).random((N + T, dim), dtype=np.float32)
    ix = oracle.OracleIndex(dim, "sq_euclid", max_edges=16, max_candidates=200, collection_size=N + T, allow_removals=False)
    ix.add_batched(x[:N], max_batch=4096, threads=8)
    ix.access_log(T * 3000)
    ix.add(x[N:])
    kind, layer, node = ix.access_log_fetch()
    starts = list(np.flatnonzero(kind == 2)) + [kind.size]
    items = []  # per item: id, reads {key: far}, changes [(key, added[], dropped[])], after {key: max(pop distance after that expansion, last far)}
    for a, b in zip(starts[:-1], starts[1:]):
        iid = int(node[a])
        reads, changes = {}, []
        seq0 = []   # layer-0 expansions in order: (key, popped node, far)
        e = a + 1
        while e < b:
            k = kind[e]
            key = (int(layer[e]) << 32) | int(node[e])
            if k == 0:
                e0 = e
                far = None
                if e + 1 < b and kind[e + 1] == 4:
                    bits = int(node[e + 1]) & 0xFFFFFFFF
                    far = np.inf if bits == 0xFFFFFFFF else float(np.array([bits], dtype=np.uint32).view(np.float32)[0])
                    if bits == 0xFFFFFFFF:
                        far = -1.0  # not full: everything is pushed
                    e += 1
                else:
                    far = -2.0      # a descent pass: no rule
                reads[key] = min(reads.get(key, np.inf), far) if key in reads else far
                if far > -2.0 and int(layer[e0]) == 0:
                    seq0.append((key, int(node[e0]), far))
            elif k == 1:
                added, dropped = [], []
                while e + 1 < b and kind[e + 1] in (5, 6):
                    (added if kind[e + 1] == 5 else dropped).append(int(node[e + 1]))
                    e += 1
                changes.append((key, added, dropped))
            e += 1
        after = {}
        if seq0:
            pd = [float(np.dot(x[iid] - x[v], x[iid] - x[v])) for _, v, _ in seq0]
            last_far = seq0[-1][2] if seq0[-1][2] >= 0 else np.inf
            run = last_far
            for i in range(len(seq0) - 1, -1, -1):
                after[seq0[i][0]] = max(after.get(seq0[i][0], 0.0), run)   # max pop distance AFTER expansion i, and the last far
                run = max(run, pd[i])
        items.append((iid, reads, changes, after))

    def dist(a, b):
        d = x[a] - x[b]
        return float(np.dot(d, d))

    def simulate(W, rule):
        n = len(items)
        log = {}  # key -> list of (item index, added, dropped)
        snap = np.full(n, -1, dtype=np.int64)
        f = rounds = searches = 0

        def valid(j):
            iid, reads, _, after = items[j]
            s = snap[j]
            for key, far in reads.items():
                for (t, added, dropped) in log.get(key, ()):
                    if t < s:
                        continue
                    if not rule or far < 0:
                        return False
                    if any(dist(iid, d) < far for d in dropped):
                        return False
                    for a in added:
                        da = dist(iid, a)
                        if da < far and not (rule == 2 and da > after.get(key, np.inf)):
                            return False
            return True
        while f < n:
            hi = min(n, f + W)
            for j in range(f, hi):
                if snap[j] < 0 or not valid(j):
                    snap[j] = f; searches += 1
            rounds += 1
            j = f
            while j < hi and (j == f or valid(j)):
                for key, added, dropped in items[j][2]:
                    log.setdefault(key, []).append((j, added, dropped))
                j += 1
            f = j
        return round(n / rounds, 2), round(searches / n, 2)
    out = {"n": N, "t": len(items)}
    for W in (64, 256):
        out[f"W{W}"] = {"true_changes_only": simulate(W, 0), "plus_distance_rule": simulate(W, 1), "plus_rule_5": simulate(W, 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
