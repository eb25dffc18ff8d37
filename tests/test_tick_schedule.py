"""CPU tier: the tick schedule of the oracle (orc_add_ticks) -- the CPU model behind DESIGN.md 4.3's analysis, not a product path.
One more deterministic member of the outcome set of HNSWIndex.Add(List)'s Parallel.For (HNSWIndex.cs:70-78): items start in id
order, at most `slots` of them are in flight, a multi-layer item searches its top layer one tick ahead of the layers below and
every item links all its layers in its last tick, in id order.  Held here: with one slot it IS the reference's sequential Add; it
never holds more items than slots; its graphs keep the reference's structural invariants; and they answer like the sequential
graph (the two variants that took every layer apart did not, or were not interleavings: see the header of orc_add_ticks)."""
import numpy as np
import pytest

import oracle
from common import normalize_f32, self_recall_at_1, uniform


def _seq(dim, x, **kw):
    ix = oracle.OracleIndex(dim, collection_size=len(x), **kw)
    ix.add(x)
    return ix


@pytest.mark.parametrize("metric,kw", [("sq_euclid", {}), ("ucosine", {"max_edges": 8}), ("cosine", {"allow_removals": False, "max_candidates": 40}),
                                       ("sq_euclid_i8", {"distribution_rate": 0.8})])
def test_one_slot_is_the_sequential_add(metric, kw):
    x = uniform(1500, 24, 7)
    if metric == "ucosine":
        x = normalize_f32(x)
    a = _seq(24, x, metric=metric, **kw)
    b = oracle.OracleIndex(24, metric, collection_size=len(x), **kw)
    ids, st = b.add_ticks(x, slots=1)
    assert (ids == np.arange(len(x))).all() and st["max_in_flight"] == 1
    assert b.graph_hash() == a.graph_hash() and b.entry_point == a.entry_point
    assert st["steps"] == st["ticks"]                      # one item per tick


@pytest.mark.parametrize("slots", [16, 256])
def test_in_flight_bound_steps_and_invariants(slots):
    n, dim, M = 5000, 32, 16
    x = uniform(n, dim, 11)
    ix = oracle.OracleIndex(dim, collection_size=n)
    ids, st = ix.add_ticks(x, slots=slots)
    assert (ids == np.arange(n)).all()
    assert st["max_in_flight"] <= slots
    lv = ix.levels()
    # a one-layer item holds its slot for one tick, a multi-layer item for two; the items that raised the top layer (and the
    # first one) went alone
    top_so_far, steps, alone = lv[0], 0, 0
    for i in range(1, n):
        if lv[i] > top_so_far:
            alone, top_so_far = alone + 1, lv[i]
        else:
            steps += 2 if lv[i] > 0 else 1
    assert st["alone"] == alone and st["steps"] == steps
    assert st["ticks"] >= steps / slots and st["long_ticks"] <= int((lv >= 2).sum())
    # structural invariants of the reference's graph (GraphConnector.cs:187-262): lists within MaxEdges(layer), no self-loop,
    # no duplicate, every neighbour alive on that layer
    for i in range(n):
        for layer in range(lv[i] + 1):
            e = ix.edges(i, layer)
            assert len(e) <= (2 * M if layer == 0 else M) and i not in e and len(set(e)) == len(e)
            assert all(lv[j] >= layer for j in e)
    # the same call again gives the same graph
    jx = oracle.OracleIndex(dim, collection_size=n)
    jx.add_ticks(x, slots=slots)
    assert jx.graph_hash() == ix.graph_hash()


def test_tick_graph_answers_like_the_sequential_graph():
    n, dim = 6000, 32
    x, q = uniform(n, dim, 3), uniform(400, dim, 4)
    d = ((q[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    truth = np.argsort(d, axis=1)[:, :10]

    def recall(ix):
        got, _ = ix.knn_query(q, 10)
        return np.mean([len(set(g) & set(t)) / 10 for g, t in zip(got.tolist(), truth.tolist())])
    seq = _seq(dim, x)
    r_seq = recall(seq)
    for slots in (64, 256):
        ix = oracle.OracleIndex(dim, collection_size=n)
        ids, _ = ix.add_ticks(x, slots=slots)
        assert self_recall_at_1(ix, x, ids) > 0.85       # the reference's own bar (bindings/__tests__/recall_test.py:7-15)
        assert recall(ix) > r_seq - 0.02, (slots, recall(ix), r_seq)


def test_ticks_continue_an_index_built_otherwise_and_removals_still_work():
    x = uniform(3000, 16, 5)
    ix = oracle.OracleIndex(16, collection_size=4096)
    ix.add(x[:1000])
    ix.add_batched(x[1000:2000], 64)
    ids, st = ix.add_ticks(x[2000:], slots=128)
    assert ids.tolist() == list(range(2000, 3000)) and st["max_in_flight"] <= 128
    assert self_recall_at_1(ix, x, np.arange(3000)) > 0.85
    ix.remove(np.arange(0, 3000, 7, dtype=np.int32))      # in-edge lists were kept current by the tick links
    left = np.setdiff1d(np.arange(3000), np.arange(0, 3000, 7))
    got, _ = ix.knn_query(x[left], 1)
    assert (got[:, 0] == left).mean() > 0.85
