"""CPU tier: the tie rules of the latency variants' register pool (csrc/dk_pool_top.h) -- identity doubts (i),
group windows (ii) with (a)-(d), order ties (iii) -- restated in plain Python and held to the reference's SearchLayer on
data where equal distances are everywhere.

The pool keeps the beam in no particular order: a pop is "the closest open entry", an eviction "the farthest entry", and
with equal distances WHICH of several such entries is taken is not what the reference's two BinaryHeaps would take (their
array layout decides, GraphNavigator.cs:123-189 over BinaryHeap.cs).  The rules' claim: whenever the traversal ends without
its `tie` verdict, the result is the reference's whichever twin was taken at every such choice -- and whenever that cannot be
shown, the verdict is raised and the exact two-heap traversal answers instead.  So the model below takes its twins AT RANDOM,
the reference side is the oracle's own SearchLayer (orc_search_layer: the restated heaps, on the oracle-built graph, queries
that are stored vectors so that both sides read the same float32 distances), and every search that ends without the verdict
must equal it: as a set for an insert's candidate list, and entry by entry over the prefix a KnnQuery consumes in order.
(traverse_sorted -- the loaded launches' one sorted list, dk_sorted_top.h -- states the same rules on positions; its way of
taking a twin, by position, is one of the choices the random model makes.)

What the data shows of the rules themselves (each switched off in turn, same searches): without (i)'s doubts at an eviction,
without (ii)'s windows, or without the closing check on doubtful entries the comparison fails within a few hundred searches;
(b) and the "hard doubt inside a window" clause are decisive once in 18 000 searches of the harsher sets below; the pop-time
doubt check and (a), (c), (d) never were on these sets -- they stay as the device states them (cautious is allowed: the price
is an exact re-run)."""
import numpy as np
import pytest

import oracle

def pool_search(ref, q_item, entry, k, ordered_prefix, ids_matter_everywhere, rng):
    """traverse_pool, statement by statement (the servers' part -- lists, distances, first visits -- done inline).
    Returns (tie, order_tie, window_closed, [(key, id)] ascending)."""
    dist = lambda i: float(ref.dist_ids(q_item, i))  # noqa: E731  (float32 value: the order of keys is the order of these)
    key, open_, ident = [], [], []                   # the pool: one slot per entry, no order
    visited = {entry}
    key.append(dist(entry)); open_.append(True); ident.append(entry)
    far = key[0]
    tie = far_doubt = doubt_hard = False
    grp_key, grp_cnt = None, 0
    window = False
    while not tie:
        opens = [s for s in range(len(key)) if open_[s]]
        if not opens:
            break
        ck = min(key[s] for s in opens)
        pos = int(rng.choice([s for s in opens if key[s] == ck]))   # WHICH twin: immaterial, says the rule
        if far_doubt and ck == far:
            tie = True; break
        if grp_cnt > 0 and ck > grp_key:                             # the window closes: (c)
            if sum(1 for v in key if v == grp_key) != grp_cnt:
                tie = True; break
            grp_cnt = 0; window = True
        open_[pos] = False
        opens = [s for s in range(len(key)) if open_[s]]
        nxt_key = min((key[s] for s in opens), default=None)
        if nxt_key is not None and nxt_key == ck:                    # (ii): the popped candidate has an open twin
            if grp_cnt == 0:
                grp_key, grp_cnt = ck, sum(1 for v in key if v == ck)
            elif ck != grp_key:
                tie = True                                           # (d)
        listed = [int(v) for v in ref.edges(ident[pos], 0)]
        fresh = [v for v in listed if v not in visited]
        visited.update(fresh)
        if not fresh:
            continue
        fk = {v: dist(v) for v in fresh}
        if grp_cnt > 0:                                              # (a), (b)
            if any(fk[v] == grp_key for v in fresh) or (len(key) >= k and any(fk[v] == far for v in fresh)):
                tie = True; break
        for v in fresh:                                              # the push loop, adjacency order
            dk = fk[v]
            if len(key) < k:
                key.append(dk); open_.append(True); ident.append(v)
                if len(key) == k:
                    far = max(key)
            elif dk < far:
                hits = [s for s in range(len(key)) if key[s] == far]
                slot = int(rng.choice(hits))                         # WHICH of the farthest leaves: immaterial, says the rule
                if len(hits) != 1:                                   # (i)
                    hard = ids_matter_everywhere or open_[slot] or any(open_[s] for s in hits if s != slot)
                    far_doubt = True
                    doubt_hard = doubt_hard or hard
                    if grp_cnt > 0 and hard:
                        tie = True
                key[slot], open_[slot], ident[slot] = dk, True, v
                was, far = far, max(key)
                if far != was:
                    far_doubt = False
            elif grp_cnt > 0 and dk == far:
                tie = True                                           # (b): turned away by equality
    if grp_cnt > 0 and not tie:                                      # (c) at the end of the search
        if sum(1 for v in key if v == grp_key) != grp_cnt:
            tie = True
        else:
            window = True
    doubtful = [far_doubt and key[s] == far for s in range(len(key))]
    order = sorted(range(len(key)), key=lambda s: (key[s], 0 if doubtful[s] else 1, s))
    first_doubt = next((r for r, s in enumerate(order) if doubtful[s]), None)
    upto = min(len(key), ordered_prefix)
    if first_doubt is not None and (doubt_hard or first_doubt < upto):
        tie = True                                                   # (i) left unresolved
    out = [(key[s], ident[s]) for s in order]
    order_tie = any(out[p][0] == out[p - 1][0] for p in range(1, upto))   # (iii)
    return tie, order_tie, window, out


def grid_index(seed, n, dim, levels, m, efc):
    rng = np.random.default_rng(seed)
    x = (rng.integers(0, levels, (n, dim))).astype(np.float32)
    ref = oracle.OracleIndex(dim, "sq_euclid", max_edges=m, max_candidates=efc, min_nn=efc, collection_size=n, random_seed=seed)
    ref.add(x)
    return ref, x, rng


@pytest.mark.parametrize("seed,n,dim,levels,m", [(11, 500, 8, 3, 5), (14, 600, 16, 2, 8), (15, 500, 6, 8, 5), (17, 500, 10, 5, 6), (18, 800, 6, 12, 5)])
def test_pool_rules_on_integer_grids(seed, n, dim, levels, m):
    ref, x, rng = grid_index(seed, n, dim, levels, m, 40)
    clean = raised = windows = 0
    for t in range(1500):
        item = int(rng.integers(0, n))
        k = int(rng.choice([4, 10, 25, 40]))
        k_out = min(k, int(rng.choice([1, 3, 5])))
        q = x[item]  # a stored vector: both sides read orc_dist_ids' float32 values
        entry = ref.find_entry_point(0, q)
        want_ids, want_d = ref.search_layer(entry, 0, k, q)          # the reference's result heap, array order
        insert_like = bool(t % 2)
        tie, order_tie, window, got = pool_search(ref, item, entry, k, k if insert_like else k_out + 1, insert_like, rng)
        if tie or (order_tie and not insert_like):
            raised += 1
            continue                                                  # the exact traversal answers this one
        clean += 1
        windows += window
        assert sorted(i for _, i in got) == sorted(int(i) for i in want_ids), "result sets differ without the verdict"
        if insert_like:
            continue  # (the order among equal distances is the heuristic's business: order_tie is handed on, dk_insert_kernels.h)
        # KnnQuery's tail (HNSWIndex.cs:119-123): OrderBy(distance) -- stable over the heap array -- then Take(k_out)
        ref_order = sorted(range(len(want_ids)), key=lambda r: float(want_d[r]))
        assert [int(want_ids[r]) for r in ref_order[:k_out]] == [i for _, i in got[:k_out]], "the ordered prefix differs without the verdict"
    assert clean >= 20, "the rules gave up on (almost) every search: nothing was compared"
    assert raised >= 20, "hardly any equal distances met: this data does not exercise the rules"
    assert windows >= 5, "no group window closed cleanly: rule (ii) was not exercised"


def harsh_data(kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "duplicates":  # sixty distinct vectors, four hundred items
        x = rng.integers(0, 3, (60, 4)).astype(np.float32)[rng.integers(0, 60, 400)]
    elif kind == "line":
        x = rng.integers(0, 40, (400, 1)).astype(np.float32)
    elif kind == "plane":
        x = rng.integers(0, 12, (500, 2)).astype(np.float32)
    else:
        x = rng.integers(0, 5, (500, 6)).astype(np.float32)
    return x


@pytest.mark.parametrize("kind", ["duplicates", "line", "plane", "grid"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_pool_rules_on_harsher_sets(kind, seed):
    # short lists (M = 4), beams from 2 to 30, duplicates of whole vectors: every second distance has a twin
    x = harsh_data(kind, seed)
    n, dim = x.shape
    ref = oracle.OracleIndex(dim, "sq_euclid", max_edges=4, max_candidates=30, min_nn=30, collection_size=n, random_seed=seed)
    ref.add(x)
    rng = np.random.default_rng(100 + seed)
    clean = 0
    for t in range(1500):
        item = int(rng.integers(0, n))
        k = int(rng.choice([2, 3, 5, 8, 16, 30]))
        k_out = min(k, int(rng.choice([1, 2, 5])))
        entry = ref.find_entry_point(0, x[item])
        want_ids, want_d = ref.search_layer(entry, 0, k, x[item])
        insert_like = bool(t % 2)
        tie, order_tie, _, got = pool_search(ref, item, entry, k, k if insert_like else k_out + 1, insert_like, rng)
        if tie or (order_tie and not insert_like):
            continue
        clean += 1
        assert sorted(i for _, i in got) == sorted(int(i) for i in want_ids), "result sets differ without the verdict"
        if not insert_like:
            ref_order = sorted(range(len(want_ids)), key=lambda r: float(want_d[r]))
            assert [int(want_ids[r]) for r in ref_order[:k_out]] == [i for _, i in got[:k_out]], "the ordered prefix differs without the verdict"
    assert clean >= 100


def test_pool_rules_never_fire_without_equal_distances():
    rng = np.random.default_rng(5)
    n, dim = 600, 10
    x = rng.random((n, dim), dtype=np.float32)
    ref = oracle.OracleIndex(dim, "sq_euclid", max_edges=6, max_candidates=40, min_nn=40, collection_size=n, random_seed=5)
    ref.add(x)
    for t in range(80):
        item = int(rng.integers(0, n))
        k = int(rng.choice([5, 20, 40]))
        entry = ref.find_entry_point(0, x[item])
        want_ids, want_d = ref.search_layer(entry, 0, k, x[item])
        if len(set(float(v) for v in want_d)) != len(want_d):
            continue
        tie, order_tie, _, got = pool_search(ref, item, entry, k, k, True, rng)
        assert not tie and not order_tie
        ref_order = sorted(range(len(want_ids)), key=lambda r: float(want_d[r]))
        assert [int(want_ids[r]) for r in ref_order] == [i for _, i in got]
