"""CPU tier: the rule behind the search launches WITHOUT a visited set (traverse_sorted's oflags bit 3), held
to the reference's SearchLayer on a model where hundreds of searches are cheap.

SearchLayer (src/HNSWIndex/GraphNavigator.cs:123-256) skips a neighbour it has seen before (:158-161, :181).  The device
kernels do not keep that set: a neighbour x met again is either still in the result list -- found by its id -- or it was turned
away / pushed out at a farthest distance that has only shrunk since, and the strict push test (:165) turns it away again.  Both
searches below are written out plainly; they must pop the same candidates in the same order, evaluate to the same result list,
and the second must evaluate a superset of the first's rows -- on random data, on clustered data, and on integer-grid data where
distances collide all the time (both models break ties by id, so that the rule itself is what is compared: on the device the
order among equal distances is the tie rules' business)."""
import heapq

import numpy as np
import pytest


def build_graph(x, m, rng):
    """Any graph will do for the rule (it is about one search): each node linked to its m nearest among a random sample plus a few
    random long edges, symmetrised and capped like a layer-0 list (2 m)."""
    n = x.shape[0]
    nbrs = [set() for _ in range(n)]
    for i in range(n):
        cand = rng.choice(n, size=min(n, 6 * m), replace=False)
        d = ((x[cand] - x[i]) ** 2).sum(1)
        for j in cand[np.argsort(d, kind="stable")[:m + 1]]:
            if j != i:
                nbrs[i].add(int(j)); nbrs[int(j)].add(i)
        for j in rng.choice(n, size=2, replace=False):
            if j != i:
                nbrs[i].add(int(j))
    return [sorted(s, key=lambda j: (hash((i, j)) & 0xffff, j))[:2 * m] for i, s in enumerate(nbrs)]


def dist(x, i, q):
    return float(np.float32(((x[i] - q) ** 2).sum(dtype=np.float32)))


def search_reference(x, g, q, entry, k):
    """SearchLayerQuery with its visited set; returns (popped ids in order, evaluated ids in order, result [(d, id)] ascending)."""
    visited = {entry}
    d0 = dist(x, entry, q)
    top = [(-d0, entry)]            # farthest first
    cand = [(d0, entry)]            # closest first
    far = d0
    popped, evaluated = [], [entry]
    while cand:
        d, c = heapq.heappop(cand)
        if d > far and len(top) >= k:
            break
        popped.append(c)
        for nb in g[c]:
            if nb in visited:
                continue
            visited.add(nb)
            dn = dist(x, nb, q)
            evaluated.append(nb)
            if len(top) < k or dn < far:
                heapq.heappush(cand, (dn, nb))
                heapq.heappush(top, (-dn, nb))
                if len(top) > k:
                    heapq.heappop(top)
                far = -top[0][0]
    return popped, evaluated, sorted((-d, i) for d, i in top)


def search_without_visited_set(x, g, q, entry, k):
    """The same two heaps with the visited set taken out: every listed neighbour is measured; one that is among the k results is
    skipped by id; everything else is left to the push test."""
    d0 = dist(x, entry, q)
    top = [(-d0, entry)]
    cand = [(d0, entry)]
    far = d0
    popped, evaluated = [], [entry]
    while cand:
        d, c = heapq.heappop(cand)
        if d > far and len(top) >= k:
            break
        popped.append(c)
        for nb in g[c]:
            dn = dist(x, nb, q)               # measured whether seen before or not
            evaluated.append(nb)
            if any(i == nb for _, i in top):
                continue                      # still a result: the reference has it visited
            if len(top) < k or dn < far:
                heapq.heappush(cand, (dn, nb))
                heapq.heappush(top, (-dn, nb))
                if len(top) > k:
                    heapq.heappop(top)
                far = -top[0][0]
    return popped, evaluated, sorted((-d, i) for d, i in top)


def search_on_one_list(x, g, q, entry, k):
    """... and the device's data structure for it: ONE list of at most k entries (distance, id, expanded).  Equal to the two
    heaps whenever no two distances in play are equal (with equal distances the heaps may keep expanding an entry that has left
    the results -- the tie rules' business on the device)."""
    d0 = dist(x, entry, q)
    lst = [[d0, entry, False]]
    popped = []
    while True:
        open_ = [e for e in lst if not e[2]]
        if not open_:
            break
        c = min(open_, key=lambda e: (e[0], e[1]))
        c[2] = True
        popped.append(c[1])
        far = max(e[0] for e in lst)
        for nb in g[c[1]]:
            dn = dist(x, nb, q)
            if any(e[1] == nb for e in lst):
                continue
            if len(lst) < k or dn < far:
                lst.append([dn, nb, False])
                if len(lst) > k:
                    lst.remove(max(lst, key=lambda e: (e[0], -e[1])))
                far = max(e[0] for e in lst)
    return popped, sorted((e[0], e[1]) for e in lst)


def distinct(values):
    return len(set(values)) == len(values)


@pytest.mark.parametrize("kind", ["uniform", "clustered", "grid"])
@pytest.mark.parametrize("k", [1, 8, 40])
def test_no_visited_set_is_the_same_search(kind, k):
    rng = np.random.default_rng(hash((kind, k)) & 0xffff)
    n, dim, m = 500, 6, 5
    if kind == "uniform":
        x = rng.random((n, dim), dtype=np.float32)
    elif kind == "clustered":
        x = (rng.random((12, dim), dtype=np.float32)[rng.integers(0, 12, n)] + 0.03 * rng.standard_normal((n, dim))).astype(np.float32)
    else:
        x = (rng.integers(0, 4, (n, dim)) / 4).astype(np.float32)
    g = build_graph(x, m, rng)
    exact_cases = 0
    for t in range(60):
        q = x[rng.integers(0, n)] + (0.0 if kind == "grid" else 0.01) * rng.standard_normal(dim).astype(np.float32) if t % 2 else rng.random(dim, dtype=np.float32)
        q = q.astype(np.float32)
        entry = int(rng.integers(0, n))
        p1, e1, r1 = search_reference(x, g, q, entry, k)
        p2, e2, r2 = search_without_visited_set(x, g, q, entry, k)
        # both models break ties the same way (by id), so the rule must hold to the letter, equal distances or not
        assert p1 == p2, "the searches expand different candidates"
        assert r1 == r2, "the result lists differ"
        assert set(e1) <= set(e2), "a row the reference evaluates was not measured"
        if distinct([dist(x, i, q) for i in set(e2)]):
            exact_cases += 1
            p3, r3 = search_on_one_list(x, g, q, entry, k)
            assert p3 == p1 and r3 == r1, "the one-list form differs from the two heaps without any equal distances"

    if kind != "grid":
        assert exact_cases >= 50
