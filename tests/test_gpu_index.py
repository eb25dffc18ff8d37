"""GPU tier: Add / KnnQuery through the reference-shaped C ABI and Python `Index`, against
the oracle and the committed golden fixtures.  Integer results (ids, levels, adjacency,
graph hash) bit-exact; distances bit-identical (north_star tolerance: 1e-5)."""
import os

import numpy as np
import pytest

import oracle
from common import default_cap, golden_cases, load_golden, normalize_f32, novis_active, self_recall_at_1, set_diag, uniform

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Index():
    import hnswindex
    assert hnswindex.net_amd.lib.hnswdev_device_count() > 0, "GPU tier needs a HIP device"
    return hnswindex.Index


TRAVERSALS = ["device", "host"]


def _build(Index, g, x, traversal="device"):
    ix = Index(g["dim"], g["metric"])
    ix.set_device_traversal(traversal == "device")
    ix.set_collection_size(g["n"])
    ix.set_random_seed(g["random_seed"])
    ix.set_max_edges(g["params"]["max_edges"])
    ix.set_max_candidates(g["params"]["max_candidates"])
    ix.set_min_nn(g["params"]["min_nn"])
    ix.set_insert_batch(g["batch"] if g["batch"] else 1)
    ids = ix.add(x)
    return ix, ids


@pytest.mark.parametrize("traversal", TRAVERSALS)
@pytest.mark.parametrize("name", golden_cases())
def test_golden_fixture(Index, name, traversal):
    g, x, q = load_golden(name)
    ix, ids = _build(Index, g, x, traversal)
    assert (ids == np.arange(g["n"])).all()
    assert ix.levels()[:128].tolist() == g["levels_head"]
    assert ix.entry_point == g["entry_point"]
    assert f"{ix.graph_hash():016x}" == g["graph_hash"]
    kid, kd = ix.knn_query(q, g["k"])
    assert kid.tolist() == g["knn_ids"]
    assert kd.view(np.uint32).tolist() == g["knn_dist_bits"]


@pytest.mark.parametrize("traversal", TRAVERSALS)
@pytest.mark.parametrize("metric", ["sq_euclid", "cosine", "ucosine"])
def test_sequential_add_and_query_match_oracle(Index, metric, traversal):
    n, dim = 1200, 128
    x, q = uniform(n, dim, 21), uniform(200, dim, 22)
    if metric == "ucosine":
        x, q = normalize_f32(x), normalize_f32(q)
    ix = Index(dim, metric)
    ix.set_device_traversal(traversal == "device")
    ix.set_collection_size(256)                 # forces two doubling resizes
    ix.set_insert_batch(1)
    ids = ix.add(x)
    ref = oracle.OracleIndex(dim, metric, collection_size=256)
    assert (ids == ref.add(x)).all()
    assert ix.graph_hash() == ref.graph_hash()
    assert ix.entry_point == ref.entry_point
    for i in (0, 1, n // 2, n - 1):
        for layer in range(ref.max_layer(i) + 1):
            assert ix.edges(i, layer).tolist() == ref.edges(i, layer).tolist()
    for k in (1, 10, 50):
        a_ids, a_d = ix.knn_query(q, k)
        b_ids, b_d = ref.knn_query(q, k)
        assert (a_ids == b_ids).all() and a_d.tobytes() == b_d.tobytes()


def test_one_at_a_time_adds_equal_one_sequential_call(Index):
    # bindings/__tests__/parameters_test.py:60-81 (the reference's determinism recipe)
    x = uniform(400, 64, 31)
    a = Index(64); a.set_random_seed(1337); a.set_collection_size(400)
    for row in x:
        a.add([row])
    b = Index(64); b.set_random_seed(1337); b.set_collection_size(400); b.set_insert_batch(1)
    b.add(x)
    assert a.graph_hash() == b.graph_hash()
    ref = oracle.OracleIndex(64, random_seed=1337, collection_size=400)
    ref.add(x)
    assert a.graph_hash() == ref.graph_hash()


def test_batched_add_matches_oracle_schedule_and_recall(Index):
    n, dim = 6000, 64
    x = uniform(n, dim, 41)
    ix = Index(dim)
    ix.set_collection_size(n)
    ix.set_insert_batch(512)
    ids = ix.add(x)
    ref = oracle.OracleIndex(dim, collection_size=n)
    ref.add_batched(x, 512)
    assert ix.graph_hash() == ref.graph_hash()
    assert self_recall_at_1(ix, x, ids) > 0.85      # recall_test.py:15
    # and adding in two calls continues the same deterministic schedule
    iy = Index(dim); iy.set_collection_size(n); iy.set_insert_batch(512)
    iy.add(x[:2500]); iy.add(x[2500:])
    rz = oracle.OracleIndex(dim, collection_size=n)
    rz.add_batched(x[:2500], 512); rz.add_batched(x[2500:], 512)
    assert iy.graph_hash() == rz.graph_hash()


def test_reference_python_thresholds_default_batched_add(Index):
    # bindings/__tests__/recall_test.py:7-15, parameters_test.py:7-57 on seeded data
    x = uniform(2000, 128, 51)
    ix = Index(128)
    ids = ix.add(x)
    default_recall = self_recall_at_1(ix, x, ids)
    assert default_recall > 0.85
    iy = Index(128); iy.set_min_nn(1)
    ids = iy.add(x)
    assert self_recall_at_1(iy, x, ids) < default_recall
    iz = Index(128); iz.set_max_edges(1)
    ids = iz.add(x)
    assert self_recall_at_1(iz, x, ids) < 0.1
    iw = Index(128); iw.set_allow_removals(False)
    ids = iw.add(x)
    assert self_recall_at_1(iw, x, ids) > 0.85


def test_metric_via_api_atol_1e5(Index):
    # bindings/__tests__/metric_test.py:34-96
    for metric in ("sq_euclid", "cosine", "ucosine"):
        x = uniform(100, 128, 61)
        if metric == "ucosine":
            x = normalize_f32(x)
        ix = Index(128, metric)
        ix.add(x)
        ids, d = ix.knn_query(x, 2)
        a, b = x.astype(np.float64), x[ids[:, 1]].astype(np.float64)
        want = ((a - b) ** 2).sum(1) if metric == "sq_euclid" else \
            1 - (a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)
        assert np.allclose(d[:, 1], want, rtol=0, atol=1e-5)


def test_edge_cases_and_errors(Index):
    ix = Index(16)
    ix.set_collection_size(4)
    assert ix.add(np.empty((0, 16), np.float32)).size == 0          # count <= 0 => 0 ids (Exports.cs:81)
    ids = ix.add(uniform(2, 16, 1))
    assert ids.tolist() == [0, 1]
    kid, kd = ix.knn_query(uniform(3, 16, 2), 4)                     # fewer than k results: padded (:144)
    assert (kid[:, :2] >= 0).all() and (kid[:, 2:] == -1).all() and np.isnan(kd[:, 2:]).all()
    ref = oracle.OracleIndex(16, collection_size=4); ref.add(uniform(2, 16, 1))
    rid, rd = ref.knn_query(uniform(3, 16, 2), 4)
    assert (kid == rid).all() and kd[:, :2].tobytes() == rd[:, :2].tobytes()
    with pytest.raises(ValueError):
        ix.add(uniform(2, 8, 1))                                      # python-side dim check (bindings.py:137)
    import hnswindex
    lib, F, I = hnswindex.net_amd.lib, hnswindex.net_amd.bindings._F, hnswindex.net_amd.bindings._I
    v = uniform(1, 8, 3); out = np.zeros(1, np.int32)
    assert lib.hnsw_add(ix._h, v.ctypes.data_as(F), 1, 8, out.ctypes.data_as(I)) == -1
    assert "dimension mismatch" in hnswindex.net_amd.last_error()


def test_stats_count_every_evaluation_and_profiling_times_kernels(Index):
    x, q = uniform(3000, 64, 71), uniform(500, 64, 72)
    ix = Index(64); ix.set_collection_size(3000); ix.set_device_traversal(False)
    ix.set_profiling(True)
    ix.add(x)
    ix.reset_stats()
    ix.knn_query(q, 10)
    s = ix.stats()
    assert s["evals"] > 500 * 10 and s["launches"] > 0
    assert s["timed_launches"] == s["launches"] and s["timed_evals"] == s["evals"]
    assert s["kernel_ms"] > 0 and s["row_bytes"] == 64 * 4


@pytest.mark.parametrize("dim,metric,M", [(127, "sq_euclid", 12), (768, "ucosine", 32), (96, "cosine", 5), (8, "sq_euclid", 40)])
def test_device_and_host_traversal_agree_with_oracle_odd_shapes(Index, dim, metric, M):
    n = 3000
    x, q = uniform(n, dim, 81), uniform(300, dim, 82)
    if metric == "ucosine":
        x, q = normalize_f32(x), normalize_f32(q)
    ref = oracle.OracleIndex(dim, metric, max_edges=M, max_candidates=60, min_nn=40, collection_size=n)
    ref.add_batched(x, 256)
    want_ids, want_d = ref.knn_query(q, 7)
    for traversal in TRAVERSALS:
        ix = Index(dim, metric)
        ix.set_collection_size(n); ix.set_max_edges(M); ix.set_max_candidates(60); ix.set_min_nn(40)
        ix.set_insert_batch(256); ix.set_device_traversal(traversal == "device")
        ix.add(x)
        assert ix.graph_hash() == ref.graph_hash()
        ids, d = ix.knn_query(q, 7)
        assert (ids == want_ids).all() and d.tobytes() == want_d.tobytes(), traversal


@pytest.mark.parametrize("spill_cap,expect_handback", [("8", True), ("8192", False)])
def test_candidate_heap_spill_and_handback_are_exact(Index, monkeypatch, spill_cap, expect_handback):
    # a tiny LDS candidate heap forces (a) the HBM spill path, (b) with a tiny spill area too,
    # the hand-back to the lock-step path; results must not change, for queries and for inserts
    x, q = uniform(4000, 64, 91), uniform(256, 64, 92)
    ref = oracle.OracleIndex(64, collection_size=4000, min_nn=64)
    ref.add_batched(x, 512)
    want_ids, want_d = ref.knn_query(q, 10)
    set_diag(monkeypatch, sorted_top="0")   # the two-heap traversal is the one with a candidate heap
    set_diag(monkeypatch, cand_cap="24")
    set_diag(monkeypatch, spill_cap=spill_cap)
    ix = Index(64); ix.set_collection_size(4000); ix.set_min_nn(64); ix.set_insert_batch(512)
    ix.add(x)
    assert ix.graph_hash() == ref.graph_hash()
    ix.reset_stats()
    ids, d = ix.knn_query(q, 10)
    assert (ids == want_ids).all() and d.tobytes() == want_d.tobytes()
    assert (ix.stats()["search_overflows"] > 0) == expect_handback


@pytest.mark.parametrize("novis", ["2", "0"])
def test_search_stats_count_device_evaluations(Index, monkeypatch, novis):
    # novis=2 is the default (search launches keep no visited set and count every row they MEASURE); novis=0 keeps the sets:
    # the device then evaluates exactly the reference's pairs
    set_diag(monkeypatch, novis=novis)
    x, q = uniform(3000, 64, 71), uniform(500, 64, 72)
    ix = Index(64); ix.set_collection_size(3000)
    ix.add(x)
    ix.set_profiling(True)
    ix.reset_stats()
    ix.knn_query(q, 10)
    s = ix.stats()
    assert s["search_launches"] == 1 and s["search_timed_launches"] == 1
    assert s["search_evals"] > 500 * 10 and s["search_kernel_ms"] > 0
    ref = oracle.OracleIndex(64, collection_size=3000); ref.add_batched(x, default_cap()); ref.reset_n_eval(); ref.knn_query(q, 10)
    # same traversal => same evaluations, except that the oracle re-measures the layer-0 entry
    # point once per query (GraphNavigator.cs:200) and re-measures the start node on each upper layer
    slack = 500 * (1 + ref.levels().max())
    from common import novis_active
    assert novis_active(s) == (novis == "2")
    if novis_active(s):
        # without a visited set the kernel also measures neighbours the reference had already seen and skips: a few per cent more
        # rows, never fewer -- and never many more (a broken listed-id lookup would re-measure every neighbour)
        assert ref.n_eval - slack <= s["search_evals"] <= 1.15 * ref.n_eval + slack
    else:
        assert abs(s["search_evals"] - ref.n_eval) <= slack


def test_config_c1_full_size_sequential(Index):
    # BASELINE.json configs[0]: 10k x 64 f32, sq_euclid, M=16, efConstruction=100, strictly
    # sequential Add then KnnQuery k=10 (MinNN default 5 => ef = 10), against the oracle
    n = 10000
    x, q = uniform(n, 64, 65537), uniform(1000, 64, 65538)
    ix = Index(64)
    ix.set_collection_size(n); ix.set_max_edges(16); ix.set_max_candidates(100); ix.set_insert_batch(1)
    ids = ix.add(x)
    ref = oracle.OracleIndex(64, max_edges=16, max_candidates=100, collection_size=n)
    assert (ids == ref.add(x)).all()
    assert ix.graph_hash() == ref.graph_hash()
    a_ids, a_d = ix.knn_query(q, 10)
    b_ids, b_d = ref.knn_query(q, 10)
    assert (a_ids == b_ids).all() and a_d.tobytes() == b_d.tobytes()


def test_config_c3_shape_ucosine_m32_efc400(Index):
    # BASELINE.json configs[2] at test size: dim 768, unit-normalised rows, ucosine, M=32,
    # efConstruction=400 (DistributionRate left at 1/ln16), batched Add, ef=128
    n = 3000
    x, q = normalize_f32(uniform(n, 768, 65537)), normalize_f32(uniform(200, 768, 65538))
    ix = Index(768, "ucosine")
    ix.set_collection_size(n); ix.set_max_edges(32); ix.set_max_candidates(400); ix.set_min_nn(128); ix.set_insert_batch(512)
    ix.add(x)
    ref = oracle.OracleIndex(768, "ucosine", max_edges=32, max_candidates=400, min_nn=128, collection_size=n)
    ref.add_batched(x, 512)
    assert ix.graph_hash() == ref.graph_hash()
    a_ids, a_d = ix.knn_query(q, 10)
    b_ids, b_d = ref.knn_query(q, 10)
    assert (a_ids == b_ids).all() and a_d.tobytes() == b_d.tobytes()


@pytest.mark.parametrize("mfma", ["1", "0"])
@pytest.mark.parametrize("metric,dim,kind", [("ucosine", 768, "unit"), ("ucosine", 256, "long"), ("cosine", 264, "raw"), ("sq_euclid", 256, "raw"),
                                             ("sq_euclid", 512, "centred")])
def test_mfma_prefiltered_heuristic_decides_like_the_exact_one(Index, monkeypatch, metric, dim, kind, mfma):
    # RelativeNeighborPruning with the MFMA Gram-block prefilter (beams above 256 candidates, rows of >= 256
    # floats): the approximate dot products only ever settle a comparison whose margin exceeds the rounding
    # bound; everything else is measured exactly -- so the graph is the oracle's, with the prefilter on or off.
    # "long": rows of length 3 under ucosine (the bound assumes unit rows: those blocks must take the exact path).
    set_diag(monkeypatch, mfma=mfma)
    n, M, efc = 2500, 24, 300
    x = uniform(n, dim, 71)
    if kind == "unit":
        x = normalize_f32(x)
    elif kind == "long":
        x = normalize_f32(x) * np.float32(3.0)
    elif kind == "centred":
        x = x - np.float32(0.5)
    q = x[:100] + np.float32(0.01)
    ix = Index(dim, metric)
    ix.set_collection_size(n); ix.set_max_edges(M); ix.set_max_candidates(efc); ix.set_min_nn(64); ix.set_insert_batch(700)
    ix.add(x)
    ref = oracle.OracleIndex(dim, metric, max_edges=M, max_candidates=efc, min_nn=64, collection_size=n)
    ref.add_batched(x, 700)
    assert ix.graph_hash() == ref.graph_hash()
    a_ids, a_d = ix.knn_query(q, 10)
    b_ids, b_d = ref.knn_query(q, 10)
    assert (a_ids == b_ids).all() and a_d.tobytes() == b_d.tobytes()


@pytest.mark.parametrize("metric,radius", [("sq_euclid", 16.0), ("cosine", 0.2), ("ucosine", 0.2)])
def test_range_query_matches_oracle(Index, metric, radius):
    # bindings/__tests__/recall_test.py:49-58, GraphTests.cs:227-244: every result within the radius;
    # and, against the oracle's SearchLayerRange restatement, the same ids / distance bits
    n, dim = 2000, 128
    x = uniform(n, dim, 111)
    if metric == "ucosine":
        x = normalize_f32(x)
    ix = Index(dim, metric); ix.set_collection_size(100); ix.set_insert_batch(256)
    ix.add(x)
    ref = oracle.OracleIndex(dim, metric, collection_size=100); ref.add_batched(x, 256)
    assert ix.graph_hash() == ref.graph_hash()
    ids, dists = ix.range_query(x[:300], radius)
    rids, rdists = ref.range_query(x[:300], radius)
    assert sum(len(a) for a in ids) > 300
    for a, b, c, d in zip(ids, dists, rids, rdists):
        assert (b <= radius).all()
        assert a.tolist() == c.tolist() and b.tobytes() == d.tobytes()


@pytest.mark.parametrize("metric", ["sq_euclid", "sq_euclid_i8"])
def test_range_query_runs_on_the_device_and_hands_back_what_it_must(Index, metric, monkeypatch):
    # graph_range_kernel answers RangeQuery.  Results of equal distance (their order is the reference's heap layout)
    # are put in order by replaying the heaps on the host; a visited table that fills up goes to the lock-step path
    n, dim = 12000, 16
    x = uniform(n, dim, 171)
    x[500:560] = x[100:160]                      # exact duplicates: equal distances in the same result
    q = np.concatenate([x[100:130], uniform(170, dim, 172)])
    ix = Index(dim, metric); ix.set_collection_size(n); ix.set_insert_batch(1024)
    ix.add(x)
    ref = oracle.OracleIndex(dim, metric, collection_size=n); ref.add_batched(x, 1024)
    assert ix.graph_hash() == ref.graph_hash()

    def same(radius, qq):
        a_ids, a_d = ix.range_query(qq, radius)
        for lo in range(0, len(qq), 1000):       # the oracle's result buffers are dense: keep them small
            b_ids, b_d = ref.range_query(qq[lo:lo + 1000], radius, cap=n if len(qq) < 100 else 6000)
            for a, b, c, d in zip(a_ids[lo:lo + 1000], a_d[lo:lo + 1000], b_ids, b_d):
                assert a.tolist() == c.tolist() and b.tobytes() == d.tobytes()
        same.tied = sum(int(len(b) > 1 and (np.diff(b) == 0).any()) for b in a_d)
        return sum(len(a) for a in a_ids)

    ix.reset_stats()
    assert same(0.9, q) > 200
    st = ix.stats()
    assert st["range_launches"] >= 1 and st["range_handbacks"] == 0
    assert same.tied >= 10                       # a duplicated row and its twin, both within reach of the first 30 queries
    same(0.0, q[:40])                            # at most the query's own row and its twin
    assert same(-1.0, q[:40]) == 0               # nothing within reach, the entry point included
    # a launch-wide arena too small for everything: the unfinished queries run once more in one of the right size
    qs = uniform(8000, dim, 173)
    ix.reset_stats()
    total = same(1.2 if metric == "sq_euclid" else 1.25, qs)
    assert total > (1 << 20) and ix.stats()["range_launches"] == 2 and ix.stats()["range_handbacks"] == 0
    ix.reset_stats()
    assert same(1e30, q[:3]) == 3 * n            # the whole graph: beyond a wave's list, run again with lists as long as the graph
    assert ix.stats()["range_launches"] >= 2 and ix.stats()["range_handbacks"] == 0
    # per-wave visited hash tables (graphs above 4M nodes), forced here
    set_diag(monkeypatch, vis_hash="1")
    ix.reset_stats()
    assert same(0.9, q) > 200
    assert ix.stats()["visited_hash_launches"] >= 1 and ix.stats()["range_handbacks"] == 0
    set_diag(monkeypatch, vis_hash_cap="64")   # (raised to 4096:) 3072 visited ids, then the traversal is handed back
    ix.reset_stats()
    assert same(1.3, q) > 10000
    assert 0 < ix.stats()["range_handbacks"] < len(q)


def test_shapes_beyond_the_device_kernels_fall_back_to_host_traversal(Index):
    # MaxEdges > 63 and beam widths beyond the LDS budget are served by the lock-step path --
    # same results, no error
    x, q = uniform(1500, 32, 131), uniform(64, 32, 132)
    ref = oracle.OracleIndex(32, max_edges=70, max_candidates=90, collection_size=1500); ref.add_batched(x, 128)
    ix = Index(32); ix.set_collection_size(1500); ix.set_max_edges(70); ix.set_max_candidates(90); ix.set_insert_batch(128)
    ix.add(x)
    assert ix.graph_hash() == ref.graph_hash()
    a, b = ix.knn_query(q, 5), ref.knn_query(q, 5)
    assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()
    ref2 = oracle.OracleIndex(32, collection_size=1500); ref2.add_batched(x, 128)
    iy = Index(32); iy.set_collection_size(1500); iy.set_insert_batch(128)
    iy.add(x)
    a, b = iy.knn_query(q[:8], 1400), ref2.knn_query(q[:8], 1400)   # k = 1400 of 1500: huge beam
    assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()


def test_removal_matches_oracle_and_reference_thresholds(Index):
    # src/HNSWIndex.Tests/GraphTests.cs:122-171 (recall after removing every other node >= 0.98 x
    # before), GraphResizeTests.cs:60-93 (Count, Ids), bindings/__tests__/recall_test.py:18-34
    n, dim = 2000, 128
    x = normalize_f32(uniform(n, dim, 65537))
    ix = Index(dim, "ucosine"); ix.set_collection_size(64); ix.set_insert_batch(1)
    ids = ix.add(x)
    ref = oracle.OracleIndex(dim, "ucosine", collection_size=64)
    ref.add(x)
    before = self_recall_at_1(ix, x, ids)
    ix.reset_stats()
    ix.remove(ids[1::2])
    ref.remove(ids[1::2])
    st = ix.stats()
    # graph-resident removal: one traversal per removed node and layer (graph_search_kernel) + one re-link launch; the
    # lock-step distance launches appear only for steps handed back (results depending on the heap-array order)
    assert st["search_launches"] >= n // 2 and st["launches"] < st["search_launches"]
    assert ix.count == n // 2 == ref.count
    assert sorted(ix.ids().tolist()) == ids[0::2].tolist()
    assert ix.ids().tolist() == ref.active_ids().tolist()          # ActiveSet order too
    assert ix.graph_hash() == ref.graph_hash() and ix.entry_point == ref.entry_point
    res, d = ix.knn_query(x[0::2], 10)
    rres, rd = ref.knn_query(x[0::2], 10)
    assert (res == rres).all() and d.tobytes() == rd.tobytes()
    assert np.isin(res, ids[0::2]).all()                            # removed ids never come back
    after = float((res[:, 0] == ids[0::2]).mean())
    assert before * 0.98 < after
    # vacated slots are reused LIFO by later adds (GraphData.cs:85-91), results stay in step
    more = normalize_f32(uniform(300, dim, 777))
    a, b = ix.add(more), ref.add(more)                              # both strictly sequential
    assert (a == b).all() and a[0] == ids[1::2][-1]
    assert ix.graph_hash() == ref.graph_hash()
    res, d = ix.knn_query(x[:200], 5)
    rres, rd = ref.knn_query(x[:200], 5)
    assert (res == rres).all() and d.tobytes() == rd.tobytes()


def test_remove_all_one_by_one_and_disabled_removals(Index):
    # GraphResizeTests.cs:95-109 (Count after every Remove), ParametersTests.cs:87 (Remove throws)
    x = uniform(200, 16, 5)
    ix = Index(16); ix.set_collection_size(10)
    ids = ix.add(x)
    ref = oracle.OracleIndex(16, collection_size=10); ref.add_batched(x, default_cap())
    for k, i in enumerate(ids):
        ix.remove([i]); ref.remove([i])
        assert ix.count == 200 - k - 1
        if k % 37 == 0:
            assert ix.graph_hash() == ref.graph_hash() and ix.entry_point == ref.entry_point
    assert ix.entry_point == -1
    kid, kd = ix.knn_query(x[:2], 3)
    assert (kid == -1).all() and np.isnan(kd).all()
    assert ix.add(x[:3]).tolist() == ref.add_batched(x[:3], default_cap()).tolist()   # rebuilt from empty, reusing slots
    iy = Index(16); iy.set_allow_removals(False)
    iy.add(x)
    with pytest.raises(RuntimeError, match="InvalidOperationException"):
        iy.remove([0])
    with pytest.raises(RuntimeError, match="not in the index"):
        ix.remove([100000])


def test_resident_query_set_gives_the_same_answers(Index):
    x, q = uniform(3000, 64, 141), uniform(400, 64, 142)
    ix = Index(64); ix.set_collection_size(3000)
    ix.add(x)
    a_ids, a_d = ix.knn_query(q, 10)
    ix.set_resident_queries(q)
    for _ in range(2):
        b_ids, b_d = ix.knn_query_resident(10)
        assert (a_ids == b_ids).all() and a_d.tobytes() == b_d.tobytes()


@pytest.mark.parametrize("metric", ["sq_euclid", "cosine"])
def test_import_graph_makes_an_identical_replica(Index, metric):
    # hnsw_mi355x_import_nodes / _edges (build once, broadcast, import): the replica has the source's graph hash and
    # answers, and grows identically afterwards (the level generator is advanced past the imported items)
    n, dim, M = 6000, 48, 12
    x, more, q = uniform(n, dim, 301), uniform(500, dim, 302), uniform(300, dim, 303)
    def make():
        ix = Index(dim, metric)
        ix.set_collection_size(4096); ix.set_max_edges(M); ix.set_max_candidates(80); ix.set_min_nn(40)
        return ix
    src = make()
    src.add(x)
    lv = src.levels()
    layers = [src.export_edges(L, 2 * M + 2 if L == 0 else M + 2) for L in range(int(lv.max()) + 1)]
    rep = make()
    rep.import_graph(x, lv, src.entry_point, layers)
    assert rep.graph_hash() == src.graph_hash() and rep.count == n and (rep.levels() == lv).all()
    a, b = src.knn_query(q, 10), rep.knn_query(q, 10)
    assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()
    assert (src.add(more) == rep.add(more)).all() and rep.graph_hash() == src.graph_hash()
    rep.remove([5, 17]); src.remove([5, 17])
    assert rep.graph_hash() == src.graph_hash()
    # refused: a second import, and lists that would send a traversal out of bounds
    with pytest.raises(RuntimeError, match="already holds items"):
        rep.import_graph(x, lv, src.entry_point, layers)
    flat = int(np.nonzero(lv == 0)[0][0])
    tall = int(np.nonzero(lv >= 1)[0][0])
    bad = [(c.copy(), e.copy()) for c, e in layers]
    bad[1][1][tall, 0] = flat                                  # layer-1 edge to a node without layer 1
    fresh = make()
    with pytest.raises(RuntimeError, match="outside the layer"):
        fresh.import_graph(x, lv, src.entry_point, bad)
    bad = [(c.copy(), e.copy()) for c, e in layers]
    bad[0][1][flat, 1] = bad[0][1][flat, 0]                    # duplicate id
    fresh2 = make()
    with pytest.raises(RuntimeError, match="duplicate id"):
        fresh2.import_graph(x, lv, src.entry_point, bad)



@pytest.mark.parametrize("metric,batch", [("sq_euclid", 64), ("ucosine", 16), ("sq_euclid_i8", 256)])
def test_batched_removal_matches_its_cpu_restatement(Index, metric, batch):
    # hnsw_mi355x_set_remove_batch(B): removals with disjoint neighbourhoods taken together (the deterministic
    # counterpart of Remove(List) = Parallel.For under region locks, HNSWIndex.cs:95-101): graph, Ids() order, entry
    # point, later queries and later adds equal the oracle's restatement of the same schedule
    n, dim = 6000, 32
    x = uniform(n, dim, 4242)
    if metric == "ucosine":
        x = normalize_f32(x)
    ix = Index(dim, metric); ix.set_collection_size(n); ix.set_insert_batch(1024); ix.set_remove_batch(batch)
    ids = ix.add(x)
    ref = oracle.OracleIndex(dim, metric, collection_size=n); ref.add_batched(x, 1024)
    assert ix.graph_hash() == ref.graph_hash()
    rng = np.random.default_rng(99)
    victims = rng.permutation(n)[:2500].astype(np.int32)
    victims[7] = ref.entry_point if ref.entry_point not in victims[:7] else victims[7]   # the entry point goes alone
    victims = np.array(list(dict.fromkeys(victims.tolist())), dtype=np.int32)
    ix.reset_stats()
    ix.remove(victims); ref.remove_batched(victims, batch)
    assert ix.count == ref.count == n - victims.size
    assert ix.ids().tolist() == ref.active_ids().tolist() and ix.entry_point == ref.entry_point
    assert ix.graph_hash() == ref.graph_hash()
    st = ix.stats()
    assert st["search_launches"] < victims.size          # far fewer launches than removals: they went in batches
    q = uniform(400, dim, 4243)
    if metric == "ucosine":
        q = normalize_f32(q)
    a, b = ix.knn_query(q, 10), ref.knn_query(q, 10)
    assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()
    assert not np.isin(a[0], victims).any()
    more = uniform(500, dim, 4244)
    if metric == "ucosine":
        more = normalize_f32(more)
    c, d = ix.add(more), ref.add_batched(more, 1024)      # vacated slots reused (LIFO), same Add schedule on both
    assert (c == d).all() and ix.graph_hash() == ref.graph_hash()


@pytest.mark.parametrize("metric", ["sq_euclid", "ucosine", "sq_euclid_i8"])
@pytest.mark.parametrize("hooks", [dict(lat=0), dict(lat=2), dict(novis_insert=0), dict(lat=2, novis_insert=0), dict(novis=0, lat=2), dict(lean=0, lat=0)])
def test_forced_traversal_forms_build_and_answer_like_the_oracle(Index, monkeypatch, metric, hooks):
    """The switches that pick a traversal form -- the latency variants never / whenever possible, Add's searches with their visited
    sets kept, the search launches with theirs, the flags-9 launches on the plain kernel forms instead of the lean ones -- set IN the suite (they are read on every call): graph hash and answers must be
    the oracle's under each of them, for batches small enough for the latency variants and for a batch that fills the chip."""
    set_diag(monkeypatch, **hooks)
    n, dim = 5000, 48
    x, q = uniform(n, dim, 321), uniform(700, dim, 322)
    if metric == "ucosine":
        x, q = normalize_f32(x), normalize_f32(q)
    ix = Index(dim, metric); ix.set_collection_size(n); ix.set_max_candidates(70); ix.set_min_nn(48); ix.set_insert_batch(4096)
    ref = oracle.OracleIndex(dim, metric, max_candidates=70, min_nn=48, collection_size=n)
    ix.add(x[:4000]); ref.add_batched(x[:4000], 4096)
    for i in range(4000, n, 40):                                  # small calls: the latency variants' launches (when allowed)
        ix.add(x[i:i + 40]); ref.add_batched(x[i:i + 40], 4096)
    assert ix.graph_hash() == ref.graph_hash()
    for qs in (q, q[:9]):
        a, b = ix.knn_query(qs, 10), ref.knn_query(qs, 10)
        assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()
    st = ix.stats()
    if hooks.get("lat") == 0:
        assert st["lat_launches"] == 0
    if hooks.get("lat") == 2:
        assert st["lat_launches"] > 0
    if hooks.get("lean") == 0:
        assert st["lean_launches"] == 0
    elif hooks.get("lat") == 0 and novis_active(st):
        assert st["lean_launches"] > 0                             # the default for launches without visited sets (kFormLean)


@pytest.mark.parametrize("metric", ["sq_euclid", "ucosine", "sq_euclid_i8"])
def test_range_order_is_completed_on_the_device(Index, metric):
    """RangeQuery's ORDER on the device (csrc/dk_range_finish.h): lists of distinct distances are ranked by counting, lists that hold
    equal distances are replayed -- the reference's two heaps on the distances already found -- and ranked in heap-array order.
    Coordinates on a coarse grid: nearly every result list holds ties, several hundred entries long; every id and distance bit
    must be the oracle's, and the host must have been left only what the kernels hand back by design (lists beyond 2 048 entries)."""
    n, dim = 9000, 8
    rng = np.random.default_rng(77)
    x = (rng.integers(0, 6, (n, dim)) / 4).astype(np.float32)
    x += (rng.random((n, dim), dtype=np.float32) * np.float32(1e-3)) * (rng.random((n, 1)) < 0.5)   # half the rows off the grid
    q = (rng.integers(0, 6, (600, dim)) / 4).astype(np.float32)
    if metric == "ucosine":
        x, q = normalize_f32(x + 0.25), normalize_f32(q + 0.25)
    ix = Index(dim, metric); ix.set_collection_size(n); ix.set_max_edges(12); ix.set_insert_batch(2048)
    ix.add(x)
    ref = oracle.OracleIndex(dim, metric, max_edges=12, collection_size=n); ref.add_batched(x, 2048)
    assert ix.graph_hash() == ref.graph_hash()
    radii = {"sq_euclid": (0.8, 1.6, 3.0), "sq_euclid_i8": (0.8, 1.6, 3.0), "ucosine": (0.02, 0.05, 0.12)}[metric]
    for radius in radii:
        ix.reset_stats()
        a_ids, a_d = ix.range_query(q, radius)
        tied = longest = 0
        for lo in range(0, len(q), 200):
            b_ids, b_d = ref.range_query(q[lo:lo + 200], radius, cap=n)
            for a, b, c, d in zip(a_ids[lo:lo + 200], a_d[lo:lo + 200], b_ids, b_d):
                assert a.tolist() == c.tolist() and b.tobytes() == d.tobytes(), (metric, radius)
                tied += int(len(b) > 1 and (np.diff(b) == 0).any())
                longest = max(longest, len(b))
        st = ix.stats()
        assert st["range_handbacks"] == 0
        lists = sum(1 for b in a_d if len(b) >= 2)
        big = sum(1 for b in a_d if len(b) > 2048)
        assert st["range_device_ordered"] + st["range_host_ordered"] == lists
        assert st["range_host_ordered"] <= big + sum(1 for b in a_d if (b == 0).any()), (st["range_host_ordered"], big)   # (a -0 distance sends a list to the host)
    assert tied > 100 and longest > 300, (tied, longest)

