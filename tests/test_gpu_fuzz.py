"""GPU tier: randomized shapes and TIE-HEAVY data against the oracle.  Small-integer coordinates
produce many exactly equal distances and duplicate vectors, which is where heap tie order,
the unstable introsort and the device's integer-key heaps could diverge from the host path."""
import os

import numpy as np
import pytest

import oracle
from common import normalize_f32, set_diag

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Index():
    import hnswindex
    assert hnswindex.net_amd.lib.hnswdev_device_count() > 0
    return hnswindex.Index


def _case(rng):
    dim = int(rng.choice([3, 8, 12, 16, 31, 64]))
    metric = str(rng.choice(["sq_euclid", "cosine", "ucosine"]))
    M = int(rng.integers(2, 24))
    efc = int(rng.integers(4, 90))
    ef = int(rng.integers(1, 70))
    k = int(rng.integers(1, 12))
    n = int(rng.integers(300, 2500))
    levels = int(rng.integers(2, 5))               # coordinate alphabet size: fewer => more ties
    x = rng.integers(0, levels, (n, dim)).astype(np.float32)
    x[rng.integers(0, n, n // 10)] = x[rng.integers(0, n, n // 10)]   # exact duplicates
    if metric != "sq_euclid":
        x += 1.0                                   # keep norms away from zero
    if metric == "ucosine":
        x = normalize_f32(x)
    q = x[rng.integers(0, n, 120)].copy()
    q[::3] += (rng.integers(0, 2, (q[::3].shape)) * 0.5).astype(np.float32)
    if metric == "ucosine":
        q = normalize_f32(q)
    batch = int(rng.choice([1, 7, 64, 16384]))
    return dict(dim=dim, metric=metric, M=M, efc=efc, ef=ef, k=k, n=n, x=x, q=q, batch=batch, seed=int(rng.integers(0, 1 << 30)))


@pytest.mark.parametrize("case_seed", range(12))
def test_tie_heavy_random_case(Index, case_seed):
    c = _case(np.random.default_rng(1000 + case_seed))
    ref = oracle.OracleIndex(c["dim"], c["metric"], max_edges=c["M"], max_candidates=c["efc"], min_nn=c["ef"],
                             collection_size=64, random_seed=c["seed"])
    if c["batch"] == 1:
        ref.add(c["x"])
    else:
        ref.add_batched(c["x"], c["batch"])
    want_ids, want_d = ref.knn_query(c["q"], c["k"])
    for traversal in ("device", "host"):
        ix = Index(c["dim"], c["metric"])
        ix.set_collection_size(64); ix.set_max_edges(c["M"]); ix.set_max_candidates(c["efc"]); ix.set_min_nn(c["ef"])
        ix.set_random_seed(c["seed"]); ix.set_insert_batch(c["batch"]); ix.set_device_traversal(traversal == "device")
        ix.add(c["x"])
        assert ix.graph_hash() == ref.graph_hash(), (traversal, {k: v for k, v in c.items() if k not in ("x", "q")})
        ids, d = ix.knn_query(c["q"], c["k"])
        assert (ids == want_ids).all() and d.tobytes() == want_d.tobytes(), traversal
    # removal of a third of the items, then queries and a re-insert, still in step
    rm = np.random.default_rng(case_seed).permutation(c["n"])[: c["n"] // 3].astype(np.int32)
    ix.remove(rm); ref.remove(rm)
    assert ix.graph_hash() == ref.graph_hash() and ix.ids().tolist() == ref.active_ids().tolist()
    ids, d = ix.knn_query(c["q"], c["k"])
    want_ids, want_d = ref.knn_query(c["q"], c["k"])
    assert (ids == want_ids).all() and d.tobytes() == want_d.tobytes()


@pytest.mark.parametrize("M,efc,ef,k,batch", [(1, 1, 1, 1, 16384), (1, 1, 1, 1, 1), (2, 1, 3, 2, 64), (63, 5, 2, 1, 16384), (3, 300, 200, 50, 256)])
def test_extreme_parameters_match_oracle(Index, M, efc, ef, k, batch):
    # bindings/__tests__/parameters_test.py:24-45 uses max_edges=1 and max_candidates=1
    from common import uniform
    x, q = uniform(1200, 24, 171), uniform(100, 24, 172)
    ref = oracle.OracleIndex(24, max_edges=M, max_candidates=efc, min_nn=ef, collection_size=1200)
    ref.add(x) if batch == 1 else ref.add_batched(x, batch)
    want = ref.knn_query(q, k)
    for traversal in ("device", "host"):
        ix = Index(24); ix.set_collection_size(1200); ix.set_max_edges(M); ix.set_max_candidates(efc); ix.set_min_nn(ef)
        ix.set_insert_batch(batch); ix.set_device_traversal(traversal == "device")
        ix.add(x)
        assert ix.graph_hash() == ref.graph_hash(), traversal
        got = ix.knn_query(q, k)
        assert (got[0] == want[0]).all() and got[1].tobytes() == want[1].tobytes(), traversal


def _build(Index, x, M, efc, ef, batch):
    ix = Index(x.shape[1])
    ix.set_collection_size(x.shape[0]); ix.set_max_edges(M); ix.set_max_candidates(efc); ix.set_min_nn(ef)
    ix.set_insert_batch(batch)
    ix.add(x)
    return ix


@pytest.mark.parametrize("sorted_top", ["1", "0"])
def test_sorted_list_and_two_heap_traversals_agree_with_the_oracle(Index, monkeypatch, sorted_top):
    # the device traversal has a fast variant (one sorted list in registers, exact when no two
    # coexisting candidates are equidistant) and the exact two-heap variant it falls back to
    from common import uniform
    set_diag(monkeypatch, sorted_top=sorted_top)
    for (M, efc, ef, k) in [(8, 60, 40, 10), (16, 200, 128, 10), (12, 400, 300, 20), (6, 30, 600, 5)]:
        x, q = uniform(3000, 32, 77), uniform(300, 32, 78)
        ref = oracle.OracleIndex(32, max_edges=M, max_candidates=efc, min_nn=ef, collection_size=3000)
        ref.add_batched(x, 16384)
        ix = _build(Index, x, M, efc, ef, 16384)
        assert ix.graph_hash() == ref.graph_hash()
        want, got = ref.knn_query(q, k), ix.knn_query(q, k)
        assert (got[0] == want[0]).all() and got[1].tobytes() == want[1].tobytes()
        assert ix.stats()["search_overflows"] == 0
        if sorted_top == "0":
            assert ix.stats()["search_repeats"] == 0


def test_equal_distances_are_repeated_with_the_exact_traversal(Index):
    # integer grid: most candidates tie; the sorted-list variant must notice every time
    rng = np.random.default_rng(5)
    x = rng.integers(0, 3, (4000, 10)).astype(np.float32)
    q = rng.integers(0, 3, (500, 10)).astype(np.float32)
    ref = oracle.OracleIndex(10, max_edges=10, max_candidates=50, min_nn=40, collection_size=4000)
    ref.add_batched(x, 16384)
    ix = _build(Index, x, 10, 50, 40, 16384)
    assert ix.graph_hash() == ref.graph_hash()
    want, got = ref.knn_query(q, 8), ix.knn_query(q, 8)
    assert (got[0] == want[0]).all() and got[1].tobytes() == want[1].tobytes()
    assert ix.stats()["search_repeats"] > 400


@pytest.mark.parametrize("shadow", ["1", "0"])
def test_draining_launch_with_ties_shadow_traversals(Index, monkeypatch, shadow):
    # Far more queries than resident waves, on data where one query in three meets a tie: when the queue runs dry the
    # idle waves start exact "shadow" traversals of the jobs still running (graph_search_kernel), and whoever finishes
    # first answers.  Every answer must still be the oracle's, with the shadows on and off, and on the hashed visited
    # set as well.
    set_diag(monkeypatch, shadow=shadow)
    rng = np.random.default_rng(11)
    x = rng.integers(0, 4, (6000, 12)).astype(np.float32)
    x += rng.random((6000, 12), dtype=np.float32) * np.float32(1e-3) * (rng.random((6000, 1)) < 0.7)  # ties in a third of the rows
    q = x[rng.integers(0, 6000, 14000)] + (rng.integers(0, 2, (14000, 12)) * 0.5).astype(np.float32)
    ref = oracle.OracleIndex(12, max_edges=12, max_candidates=60, min_nn=48, collection_size=6000)
    ref.add_batched(x, 16384)
    want = ref.knn_query(q, 10)
    for vis_hash in ("0", "1"):
        set_diag(monkeypatch, vis_hash=vis_hash)
        ix = _build(Index, x, 12, 60, 48, 16384)
        assert ix.graph_hash() == ref.graph_hash()
        for _ in range(2):
            got = ix.knn_query(q, 10)
            assert (got[0] == want[0]).all() and got[1].tobytes() == want[1].tobytes()
        st = ix.stats()
        assert st["search_overflows"] == 0 and st["search_repeats"] > 100
        del ix


def test_group_windows_of_equal_distances_close_without_a_rerun(Index):
    # Coordinates on a 1/32 grid: squared distances are multiples of 1/1024, so two OPEN candidates of equal distance
    # come up in most searches -- tie (ii).  The sorted traversal opens a group window and goes on (device_kernels.h);
    # the answers, ids and distance bits, must be the oracle's two-heap answers whether the window closed cleanly
    # (stats: tie_windows) or the search was handed to the exact traversal (search_repeats).
    rng = np.random.default_rng(23)
    for dim, n, M, efc, ef, k in ((24, 20000, 12, 80, 64, 10), (16, 8000, 8, 60, 100, 20), (48, 12000, 16, 100, 40, 5)):
        x = (rng.integers(0, 32, (n, dim)) / np.float32(32)).astype(np.float32)
        q = (rng.integers(0, 64, (6000, dim)) / np.float32(64)).astype(np.float32)
        ref = oracle.OracleIndex(dim, max_edges=M, max_candidates=efc, min_nn=ef, collection_size=n)
        ref.add_batched(x, 16384)
        ix = _build(Index, x, M, efc, ef, 16384)
        assert ix.graph_hash() == ref.graph_hash()
        want, got = ref.knn_query(q, k), ix.knn_query(q, k)
        assert (got[0] == want[0]).all() and got[1].tobytes() == want[1].tobytes(), (dim, n)
        st = ix.stats()
        assert st["tie_windows"] > 200 and st["search_overflows"] == 0, st


def test_many_jobs_per_resident_wave(Index):
    # persistent launches: far more traversals than resident waves, so every wave reuses its visited
    # bitset (cleared in the kernel) many times; also Add in one call with a batch cap above the slots
    from common import uniform
    x, q = uniform(6000, 16, 301), uniform(120_000, 16, 302)
    ref = oracle.OracleIndex(16, max_edges=8, max_candidates=40, min_nn=24, collection_size=6000)
    ref.add_batched(x, 16384)
    ix = _build(Index, x, 8, 40, 24, 16384)
    assert ix.graph_hash() == ref.graph_hash()
    got = ix.knn_query(q, 6)
    want = ref.knn_query(q, 6, threads=8)
    assert (got[0] == want[0]).all() and got[1].tobytes() == want[1].tobytes()
    again = ix.knn_query(q[:5000], 6)   # the scratch was left clean
    assert (again[0] == want[0][:5000]).all()


@pytest.mark.parametrize("overlap", ["2", "0"])
def test_row_loads_overlapped_with_visited_atomics(Index, monkeypatch, overlap):
    # launches that do not fill the chip fetch the rows of all listed neighbours together with the
    # visited atomics (2 forces that for every launch, 0 forbids it): same results either way
    from common import uniform
    set_diag(monkeypatch, overlap=overlap)
    x, q = uniform(8000, 20, 601), uniform(20_000, 20, 602)
    ref = oracle.OracleIndex(20, max_edges=10, max_candidates=60, min_nn=32, collection_size=8000)
    ref.add_batched(x, 16384)
    ix = _build(Index, x, 10, 60, 32, 16384)
    assert ix.graph_hash() == ref.graph_hash()
    got, want = ix.knn_query(q, 7), ref.knn_query(q, 7, threads=8)
    assert (got[0] == want[0]).all() and got[1].tobytes() == want[1].tobytes()
    ix.set_profiling(True); ix.reset_stats()
    ref.reset_n_eval()
    ix.knn_query(q[:2000], 7); ref.knn_query(q[:2000], 7)
    # evaluations are counted for the unvisited neighbours only, overlapped or not -- unless the launch ran without a visited set
    # (hash-table graphs, forced from outside by diagnostic vis_hash=1): then every row measured counts
    st = ix.stats()
    from common import novis_active
    if novis_active(st):
        assert ref.n_eval - 2000 * (1 + ref.levels().max()) <= st["search_evals"] <= 1.15 * ref.n_eval + 2000 * (1 + ref.levels().max())
    else:
        assert abs(st["search_evals"] - ref.n_eval) <= 2000 * (1 + ref.levels().max())


@pytest.mark.parametrize("cap,expect_handback", [("16384", False), ("512", True)])
@pytest.mark.parametrize("sorted_top,novis", [("1", "0"), ("0", "0"), ("1", "2")])
def test_visited_id_hash_table(Index, monkeypatch, sorted_top, novis, cap, expect_handback):
    # large graphs keep the visited ids of a traversal in a per-wave hash table instead of a bitset
    # (forced here on a small graph); a table that fills up hands the job to the host traversal.
    # novis = "2" is the default of the search launches: the sorted traversal keeps no set at all, only its exact re-runs and
    # the shadow traversals of a draining launch do -- whether one of those fills a 512-entry table is a matter of timing,
    # so that leg checks the answers only
    from common import uniform
    set_diag(monkeypatch, vis_hash="1")
    set_diag(monkeypatch, vis_hash_cap=cap)
    set_diag(monkeypatch, sorted_top=sorted_top)
    set_diag(monkeypatch, novis=novis)
    x, q = uniform(6000, 24, 701), uniform(6000, 24, 702)
    ref = oracle.OracleIndex(24, max_edges=8, max_candidates=50, min_nn=96, collection_size=6000)
    ref.add_batched(x, 16384)
    ix = _build(Index, x, 8, 50, 96, 16384)
    assert ix.graph_hash() == ref.graph_hash()
    ix.reset_stats()
    got, want = ix.knn_query(q, 5), ref.knn_query(q, 5, threads=8)
    assert (got[0] == want[0]).all() and got[1].tobytes() == want[1].tobytes()
    if novis == "0" or not expect_handback:
        assert (ix.stats()["search_overflows"] > 0) == expect_handback


@pytest.mark.parametrize("case_seed", range(4))
def test_tie_heavy_batched_removal(Index, case_seed):
    # the batched removal schedule on integer-grid data (equal distances and duplicate vectors everywhere): the exact
    # two-heap search hands the re-link kernel the reference's candidate arrays, so Span.Sort's tie order follows
    rng = np.random.default_rng(5000 + case_seed)
    dim = int(rng.choice([4, 8, 16]))
    metric = str(rng.choice(["sq_euclid", "cosine"]))
    M = int(rng.integers(3, 14))
    n = int(rng.integers(800, 2500))
    x = rng.integers(0, 3, (n, dim)).astype(np.float32) + (1.0 if metric == "cosine" else 0.0)
    B = int(rng.choice([2, 16, 128]))
    seed = int(rng.integers(0, 1 << 30))
    ref = oracle.OracleIndex(dim, metric, max_edges=M, max_candidates=40, collection_size=64, random_seed=seed)
    ref.add_batched(x, 512)
    ix = Index(dim, metric)
    ix.set_collection_size(64); ix.set_max_edges(M); ix.set_max_candidates(40); ix.set_random_seed(seed); ix.set_insert_batch(512); ix.set_remove_batch(B)
    ix.add(x)
    assert ix.graph_hash() == ref.graph_hash()
    victims = rng.permutation(n)[: n // 3].astype(np.int32)
    ix.remove(victims); ref.remove_batched(victims, B)
    assert ix.graph_hash() == ref.graph_hash() and ix.ids().tolist() == ref.active_ids().tolist() and ix.entry_point == ref.entry_point
    q = x[rng.integers(0, n, 100)]
    a, b = ix.knn_query(q, 5), ref.knn_query(q, 5)
    assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()
