"""CPU tier: the exact-window Add as a MODEL on the oracle.  hnsw_index.cpp::insert_exact_window builds the graph of strictly
sequential inserts (HNSWIndex.Add(item) per item, /root/reference/src/HNSWIndex/HNSWIndex.cs:55-65) from speculative searches;
what makes that sound are three rules about which earlier writes a search result survives.  Here the same schedule runs on the
CPU restatement, taken apart into its steps (oracle: orc_window_search / _dry / _link), with the rules restated in Python, and
the result must be the graph of the oracle's plain sequential Add -- over many small random cases, which the GPU tier
(tests/test_gpu_exact_window.py: the real implementation, fewer cases) cannot afford.

 rule 0  a result is good while no list it READ (descent passes, expansions: GraphNavigator.cs:65,152-156) has been written since
 rule 1  an append whose PruneOverflow leaves the list reading exactly as before is not a write (GraphConnector.cs:207-212)
 rule 2  a changed layer-0 list is harmless to a reader whose result list was full, with farthest distance f, when it expanded
         that node, if every id the list gained or lost is at distance >= f from the reader (GraphNavigator.cs:165)
"""
import numpy as np
import pytest

import oracle
from common import uniform


def _reads(ix):
    """Parse the access log of ONE window_search: ({layer-0 node: far}, set of upper (layer, node))."""
    kind, layer, node = ix.access_log_fetch()
    r0, ru = {}, set()
    e = 0
    while e < kind.size:
        if kind[e] == 0:
            far = None
            if e + 1 < kind.size and kind[e + 1] == 4:
                bits = int(node[e + 1]) & 0xFFFFFFFF
                far = None if bits == 0xFFFFFFFF else float(np.array([bits], dtype=np.uint32).view(np.float32)[0])
                e += 1
                if layer[e - 1] == 0:
                    v = int(node[e - 1])
                    r0[v] = far if v not in r0 else (None if (far is None or r0[v] is None) else min(far, r0[v]))
                else:
                    ru.add((int(layer[e - 1]), int(node[e - 1])))
            else:  # a descent pass
                ru.add((int(layer[e]), int(node[e])))
        e += 1
    return r0, ru


def window_build(ix, ids, W, rule1, rule2):
    """Links the allocated, unlinked nodes `ids` (in order) through windows of W; returns (rounds, searches)."""
    n = len(ids)
    spec = {}          # t -> dict(snap, sels, r0, ru, dry)
    mod = {}           # (layer, node) -> seq of the last write
    seq = 0
    p = rounds = searches = 0
    top_of = lambda: ix.max_layer(ix.entry_point)
    while p < n:
        if ix.entry_point < 0 or ix.max_layer(ids[p]) > top_of():   # alone: it moves the entry point (GraphConnector.cs:27-41)
            ix.connect_allocated(ids[p]); p += 1; seq += 1; spec.clear(); continue
        hi = min(n, p + W)
        for t in range(p + 1, hi):
            if ix.max_layer(ids[t]) > top_of():
                hi = t; break
        R = seq
        for t in range(p, hi):                                   # search what has no good result
            s = spec.get(t)
            ok = s is not None and all(mod.get((0, v), 0) <= s["snap"] for v in s["r0"]) and all(mod.get(k, 0) <= s["snap"] for k in s["ru"])
            if ok:
                continue
            ix.access_log(200000)
            sels = ix.window_search(ids[t])
            r0, ru = _reads(ix)
            ix.access_log(0)
            dry = [[ix.window_dry(nb, l, ids[t]) for nb in sel] for l, sel in enumerate(sels)] if rule1 else None
            spec[t] = dict(snap=R, sels=sels, r0=r0, ru=ru, dry=dry)
            searches += 1
        rounds += 1
        first = {}                                              # layer-0 list -> first change of this round (t, gained, lost or None)
        second = {}
        t = p
        while t < hi:
            s = spec[t]
            blocked = any(mod.get(k, 0) > s["snap"] for k in s["ru"])   # upper layers: rule 0 / 1 only
            for v, far in s["r0"].items():
                if v not in first:
                    continue
                ct, gained, lost = first[v]
                if v in second or not rule2 or far is None or lost is None:
                    blocked = True; break
                if any(ix.dist_ids(ids[t], a) < far for a in gained + lost):
                    blocked = True; break
            if blocked and t > p:
                break
            seq += 1
            for l, sel in enumerate(s["sels"]):
                for e, nb in enumerate(sel):
                    same = rule1 and s["dry"][l][e][0] == 0 and mod.get((l, int(nb)), 0) <= s["snap"]
                    if same:
                        continue
                    mod[(l, int(nb))] = seq
                    if l == 0:
                        if int(nb) in first:
                            second[int(nb)] = t
                        else:
                            code, lost = s["dry"][0][e] if rule1 else (1, None)
                            known = rule1 and lost is not None and code != 0
                            first[int(nb)] = (t, [ids[t]] if (code & 2) else [], lost if known else None)
            ix.window_link(ids[t], s["sels"])
            del spec[t]
            t += 1
        p = t
    return rounds, searches


CASES = [(600, 16, 8, 40, 8), (900, 24, 6, 30, 16), (1200, 12, 16, 60, 32), (700, 32, 5, 25, 5)]


@pytest.mark.parametrize("rules", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("case", CASES)
def test_window_schedule_builds_the_sequential_graph(case, rules):
    n, dim, M, efc, W = case
    for seed in range(2):
        x = uniform(n, dim, 1000 * seed + n)
        kw = dict(max_edges=M, max_candidates=efc, collection_size=n, random_seed=77 + seed, allow_removals=False)
        ref = oracle.OracleIndex(dim, **kw)
        ref.add(x)
        ix = oracle.OracleIndex(dim, **kw)
        n0 = 40
        ix.add(x[:n0])                                          # a small sequential start, then windows
        ids = ix.alloc_only(x[n0:])
        rounds, searches = window_build(ix, ids.tolist(), W, *rules)
        assert ix.graph_hash() == ref.graph_hash(), (case, rules, seed)
        assert rounds <= n - n0


def test_rules_buy_items_per_round():
    n, dim = 2500, 16
    x = uniform(n, dim, 5)
    out = []
    for rules in ((False, False), (True, False), (True, True)):
        ix = oracle.OracleIndex(dim, max_edges=8, max_candidates=40, collection_size=n, allow_removals=False)
        ix.add(x[:500])
        ids = ix.alloc_only(x[500:])
        rounds, _ = window_build(ix, ids.tolist(), 32, *rules)
        out.append((n - 500) / rounds)
    assert out[0] < out[1] < out[2], out                        # each rule links more items per round


def test_tie_heavy_data_with_the_sequence_rule_only():
    # equal distances everywhere: searches are order-sensitive there, so only rules 0 and 1 (the list reads EXACTLY as before) apply
    rng = np.random.default_rng(3)
    x = rng.integers(0, 3, size=(700, 10)).astype(np.float32)
    kw = dict(max_edges=6, max_candidates=30, collection_size=700, allow_removals=False)
    ref = oracle.OracleIndex(10, **kw); ref.add(x)
    ix = oracle.OracleIndex(10, **kw); ix.add(x[:30])
    window_build(ix, ix.alloc_only(x[30:]).tolist(), 12, True, False)
    assert ix.graph_hash() == ref.graph_hash()
