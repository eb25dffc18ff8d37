"""Oracle at index level: the reference's own assertions restated on
seeded data, plus the committed golden fixtures."""
import numpy as np
import pytest

import oracle
from common import golden_cases, load_golden, normalize_f32, self_recall_at_1, uniform


@pytest.mark.parametrize("name", golden_cases())
def test_golden_fixture(name):
    g, x, q = load_golden(name)
    ix = oracle.OracleIndex(g["dim"], g["metric"], collection_size=g["n"], random_seed=g["random_seed"], **g["params"])
    ids = ix.add_batched(x, g["batch"]) if g["batch"] else ix.add(x)
    assert (ids == np.arange(g["n"])).all()
    assert ix.levels()[:128].tolist() == g["levels_head"]
    assert np.bincount(ix.levels()).tolist() == g["level_histogram"]
    assert ix.entry_point == g["entry_point"]
    assert f"{ix.graph_hash():016x}" == g["graph_hash"]
    kid, kd = ix.knn_query(q, g["k"])
    assert kid.tolist() == g["knn_ids"]
    assert kd.view(np.uint32).tolist() == g["knn_dist_bits"]


def test_default_recall_sq_euclid():
    # bindings/__tests__/recall_test.py:7-15
    x = uniform(2000, 128, 1)
    ix = oracle.OracleIndex(128)
    ids = ix.add(x)
    assert self_recall_at_1(ix, x, ids) > 0.85


def test_build_graph_single_thread_ucosine_and_edge_balance():
    # src/HNSWIndex.Tests/GraphTests.cs:16-37
    x = normalize_f32(uniform(2000, 128, 65537))
    ix = oracle.OracleIndex(128, "ucosine")
    ids = ix.add(x)
    assert self_recall_at_1(ix, x, ids) > 0.85
    lv = ix.levels()
    for layer in range(lv.max() + 1):
        nodes = np.nonzero(lv >= layer)[0]
        out_total = sum(ix.edges(i, layer).size for i in nodes)
        in_total = sum(ix.edges(i, layer, incoming=True).size for i in nodes)
        assert out_total == in_total  # AvgOutEdges == AvgInEdges
    # in-edges are exactly the transpose of out-edges
    for i in range(0, 2000, 97):
        for j in ix.edges(i, 0):
            assert i in ix.edges(int(j), 0, incoming=True)


def test_parameter_min_nn_window():
    # ParametersTests.cs:14-30: MinNN=1 => 0.70 < recall < 0.90 (cosine on normalised data)
    x = normalize_f32(uniform(1000, 128, 65537))
    ix = oracle.OracleIndex(128, "cosine", min_nn=1)
    ids = ix.add(x)
    r = self_recall_at_1(ix, x, ids)
    assert 0.70 < r < 0.90, r


def test_parameter_max_candidates_32():
    # ParametersTests.cs:32-48
    x = normalize_f32(uniform(1000, 128, 65537))
    ix = oracle.OracleIndex(128, "cosine", max_candidates=32)
    ids = ix.add(x)
    assert self_recall_at_1(ix, x, ids) > 0.90


def test_parameter_low_recall():
    # ParametersTests.cs:50-66
    x = normalize_f32(uniform(1000, 128, 65537))
    ix = oracle.OracleIndex(128, "cosine", max_edges=8, min_nn=1, max_candidates=16)
    ids = ix.add(x)
    assert self_recall_at_1(ix, x, ids) < 0.50


def test_parameter_allow_removals_false():
    # ParametersTests.cs:68-88: recall > 0.9, no in-edges kept
    x = uniform(1000, 128, 65537)
    ix = oracle.OracleIndex(128, "sq_euclid", allow_removals=False)
    ids = ix.add(x)
    assert self_recall_at_1(ix, x, ids) > 0.9
    assert all(ix.edges(i, 0, incoming=True).size == 0 for i in range(0, 1000, 50))
    # AllowRemovals only switches in-edge upkeep: the out-graph is the same
    iy = oracle.OracleIndex(128, "sq_euclid", allow_removals=True)
    iy.add(x)
    assert ix.graph_hash() == iy.graph_hash()


def test_python_parameter_thresholds():
    # bindings/__tests__/parameters_test.py:24-45
    x = uniform(2000, 128, 3)
    ix = oracle.OracleIndex(128, max_edges=1)
    ids = ix.add(x)
    assert self_recall_at_1(ix, x, ids) < 0.1
    iy = oracle.OracleIndex(128, max_candidates=1)
    ids = iy.add(x)
    assert self_recall_at_1(iy, x, ids) < 0.6


def test_resize_from_small_collection_size():
    # recall_test.py:37-46 / GraphResizeTests.cs:16-125
    x = uniform(2000, 128, 4)
    a = oracle.OracleIndex(128, collection_size=100)
    ids = a.add(x)
    assert self_recall_at_1(a, x, ids) > 0.85
    b = oracle.OracleIndex(128, collection_size=4096)
    b.add(x)
    assert a.graph_hash() == b.graph_hash()  # capacity never changes results


def test_determinism_and_threaded_queries():
    # GraphTests.cs:82-120 (multi-thread ids == single-thread ids); parameters_test.py:60-81
    x = normalize_f32(uniform(2000, 128, 65537))
    a = oracle.OracleIndex(128, "ucosine", random_seed=1337)
    b = oracle.OracleIndex(128, "ucosine", random_seed=1337)
    a.add(x)
    for row in x:  # one-at-a-time adds, as the reference's determinism test does
        b.add(row)
    assert a.graph_hash() == b.graph_hash()
    s_ids, s_d = a.knn_query(x, 10, threads=1)
    m_ids, m_d = a.knn_query(x, 10, threads=4)
    assert (s_ids == m_ids).all() and s_d.tobytes() == m_d.tobytes()


def test_metric_via_api_atol_1e5():
    # bindings/__tests__/metric_test.py:34-96: distance to the 2nd neighbour vs float64
    for metric in ("sq_euclid", "cosine", "ucosine"):
        x = uniform(100, 128, 11)
        if metric == "ucosine":
            x = normalize_f32(x)
        ix = oracle.OracleIndex(128, metric)
        ix.add(x)
        ids, d = ix.knn_query(x, 2)
        a, b = x.astype(np.float64), x[ids[:, 1]].astype(np.float64)
        if metric == "sq_euclid":
            want = ((a - b) ** 2).sum(1)
        else:
            want = 1 - (a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)
        assert np.allclose(d[:, 1], want, rtol=0, atol=1e-5)


def test_batched_schedule_b1_is_sequential_and_quality_holds():
    x = uniform(3000, 64, 5)
    a = oracle.OracleIndex(64); a.add(x)
    b = oracle.OracleIndex(64); b.add_batched(x, 1)
    assert a.graph_hash() == b.graph_hash()
    c = oracle.OracleIndex(64); ids = c.add_batched(x, 4096)
    assert self_recall_at_1(c, x, ids) > 0.85


def test_empty_and_small_edge_cases():
    ix = oracle.OracleIndex(16)
    ids, d = ix.knn_query(uniform(3, 16, 1), 4)
    assert (ids == -1).all() and np.isnan(d).all()  # Exports.cs:144 padding
    ix.add(uniform(2, 16, 2))
    ids, d = ix.knn_query(uniform(3, 16, 1), 4)
    assert ((ids[:, :2] >= 0).all()) and (ids[:, 2:] == -1).all() and np.isnan(d[:, 2:]).all()
    assert (np.diff(d[:, :2], axis=1) >= 0).all()


def test_range_query_results_within_radius_and_sorted():
    # bindings/__tests__/recall_test.py:49-58; src/HNSWIndex.Tests/GraphTests.cs:227-244
    x = uniform(2000, 128, 6)
    ix = oracle.OracleIndex(128, collection_size=100)
    ix.add(x)
    ids, d = ix.range_query(x[:200], 16.0)
    assert sum(len(a) for a in ids) > 200
    for a, b in zip(ids, d):
        assert (b <= 16.0).all() and (np.diff(b) >= 0).all() and len(set(a.tolist())) == len(a)


def test_removal_reference_thresholds_and_structure():
    # src/HNSWIndex.Tests/GraphTests.cs:122-171; GraphResizeTests.cs:60-109; recall_test.py:18-34
    x = normalize_f32(uniform(2000, 128, 65537))
    ix = oracle.OracleIndex(128, "ucosine")
    ids = ix.add(x)
    before = self_recall_at_1(ix, x, ids)
    ix.remove(ids[1::2])
    assert ix.count == 1000 and set(ix.active_ids().tolist()) == set(ids[0::2].tolist())
    res, _ = ix.knn_query(x[0::2], 1)
    assert before * 0.98 < float((res[:, 0] == ids[0::2]).mean())
    act, lv = set(ix.active_ids().tolist()), ix.levels()
    out_total = in_total = 0
    for i in act:                                   # in/out balance and no edge to a removed node
        for layer in range(lv[i] + 1):
            o = ix.edges(i, layer)
            out_total += o.size
            in_total += ix.edges(i, layer, incoming=True).size
            assert all(int(j) in act for j in o)
    assert out_total == in_total
    assert ix.add(uniform(3, 128, 9)).tolist() == [int(ids[1::2][-1]), int(ids[1::2][-2]), int(ids[1::2][-3])]  # LIFO slot reuse


def test_remove_all_then_rebuild_and_disabled_removals():
    x = uniform(300, 16, 1)
    ix = oracle.OracleIndex(16, collection_size=10)
    ids = ix.add(x)
    for k, i in enumerate(ids):
        ix.remove([i])
        assert ix.count == 300 - k - 1
    assert ix.entry_point == -1
    assert ix.add(x[:2]).tolist() == [299, 298] and ix.entry_point == 299
    iy = oracle.OracleIndex(16, allow_removals=False)
    iy.add(x)
    with pytest.raises(RuntimeError):
        iy.remove([0])


def test_threaded_batched_add_builds_the_same_graph():
    # bench.py's like-for-like CPU baseline: the batched schedule with searches and per-list link work spread
    # over host threads must end in the very graph the single-thread restatement builds
    x = uniform(6000, 24, 77)
    hashes = set()
    for threads, removals in ((1, False), (4, False), (7, False), (4, True)):
        ix = oracle.OracleIndex(24, max_candidates=60, collection_size=6000, allow_removals=removals)
        ix.add_batched(x, 65536, threads=threads)
        hashes.add(ix.graph_hash())
    assert len(hashes) == 1
    # bounded batches: B = 16 per call, threaded == single-thread == one call with max_batch 16 (no new entry point inside)
    a = oracle.OracleIndex(24, max_candidates=60, collection_size=6000, allow_removals=False)
    b = oracle.OracleIndex(24, max_candidates=60, collection_size=6000, allow_removals=False)
    for i in range(0, 3000, 16):
        a.add_batched(x[i:i + 16], 16, threads=4)
        b.add_batched(x[i:i + 16], 16, threads=1)
    assert a.graph_hash() == b.graph_hash()


def test_rng_skip_continues_an_imported_graph_exactly():
    x, more = uniform(3000, 16, 5), uniform(200, 16, 6)
    a = oracle.OracleIndex(16, collection_size=3200, allow_removals=False)
    a.add(x)
    lv = a.levels()
    layers = []
    for L in range(int(lv.max()) + 1):
        cnt = np.full(lv.size, -1, np.int32)
        ed = np.zeros((lv.size, 34), np.int32)
        for i in np.nonzero(lv >= L)[0]:
            e = a.edges(int(i), L)
            cnt[i] = e.size
            ed[i, :e.size] = e
        layers.append((cnt, ed))
    b = oracle.OracleIndex(16, collection_size=3200, allow_removals=False)
    b.import_graph(x, lv, a.entry_point, layers)
    assert b.graph_hash() == a.graph_hash()
    b.rng_skip(3000)
    a.add(more); b.add(more)
    assert b.graph_hash() == a.graph_hash()



def test_batched_removal_schedule_of_the_oracle():
    # orc_remove_batched (the builder's schedule, hnsw_mi355x_set_remove_batch): with B = 1 it IS the sequential
    # Remove; with B > 1 the ids leave in another order and on a snapshot, but the same nodes are gone, nothing
    # points at them any more, and every list stays within MaxEdges without duplicates
    import numpy as np
    rng = np.random.default_rng(5)
    n, dim, M = 3000, 16, 8
    x = rng.random((n, dim), dtype=np.float32)
    victims = rng.permutation(n)[:900].astype(np.int32)

    def build():
        ix = oracle.OracleIndex(dim, "sq_euclid", max_edges=M, max_candidates=60, collection_size=n)
        ix.add_batched(x, 256, threads=4)
        return ix
    seq, one, many = build(), build(), build()
    victims[3] = seq.entry_point if seq.entry_point not in victims[:3] else victims[3]
    victims = np.array(list(dict.fromkeys(victims.tolist())), dtype=np.int32)
    seq.remove(victims); one.remove_batched(victims, 1); many.remove_batched(victims, 64)
    assert one.graph_hash() == seq.graph_hash() and one.active_ids().tolist() == seq.active_ids().tolist()
    assert many.count == seq.count == n - victims.size
    assert sorted(many.active_ids().tolist()) == sorted(seq.active_ids().tolist())
    gone = set(victims.tolist())
    for i in many.active_ids().tolist():
        for layer in range(many.max_layer(i) + 1):
            e = many.edges(i, layer).tolist()
            assert len(e) == len(set(e)) <= (2 * M if layer == 0 else M) and not (set(e) & gone) and i not in e
    q = rng.random((100, dim), dtype=np.float32)
    ids, _ = many.knn_query(q, 5)
    assert not np.isin(ids, victims).any() and (ids >= 0).all()
    with pytest.raises(RuntimeError):
        many.remove_batched(victims[:1], 8)          # already removed
