"""GPU tier: the exact-window Add (hnsw_mi355x_set_insert_batch(-W)) builds the graph of strictly sequential
inserts -- HNSWIndex.Add(item) per item, /root/reference/src/HNSWIndex/HNSWIndex.cs:55-65, the reference's own
determinism recipe (bindings/__tests__/parameters_test.py:65-68).  Every case compares the graph hash, the levels
and later query answers with the oracle's orc_add (sequential) on the same vectors."""
import numpy as np
import pytest

import oracle
from common import normalize_f32, uniform

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Index():
    import hnswindex
    assert hnswindex.net_amd.lib.hnswdev_device_count() > 0, "GPU tier needs a HIP device"
    return hnswindex.Index


def _pair(Index, n, dim, metric, W, *, M=16, efc=100, seed=31337, collection=None, allow_removals=True, x=None, calls=1):
    x = uniform(n, dim, 4242 + n + dim) if x is None else x
    if metric == "ucosine":
        x = normalize_f32(x)
    ix = Index(dim, metric)
    ix.set_collection_size(collection or n); ix.set_max_edges(M); ix.set_max_candidates(efc); ix.set_random_seed(seed)
    ix.set_allow_removals(allow_removals); ix.set_insert_batch(-W)
    ids = np.concatenate([ix.add(part) for part in np.array_split(x, calls)])
    ref = oracle.OracleIndex(dim, metric, max_edges=M, max_candidates=efc, collection_size=collection or n, random_seed=seed,
                             allow_removals=allow_removals)
    rids = ref.add(x)
    assert (ids == rids).all()
    return ix, ref, x


@pytest.mark.parametrize("W", [2, 7, 32, 256])
def test_window_sizes_give_the_sequential_graph(Index, W):
    ix, ref, x = _pair(Index, 3000, 32, "sq_euclid", W)
    assert ix.graph_hash() == ref.graph_hash()
    assert (ix.levels() == ref.levels()).all() and ix.entry_point == ref.entry_point
    st = ix.exact_window_stats()
    assert st["linked"] + st["alone"] == 3000 - 1 and st["rounds"] >= 1 and st["searches"] >= st["linked"]
    q = uniform(200, 32, 9)
    a, b = ix.knn_query(q, 10), ref.knn_query(q, 10)
    assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()


@pytest.mark.parametrize("metric", ["sq_euclid", "cosine", "ucosine", "sq_euclid_i8"])
def test_every_metric(Index, metric):
    ix, ref, _ = _pair(Index, 2500, 96, metric, 48, efc=120)
    assert ix.graph_hash() == ref.graph_hash()


def test_c1_full_size(Index):
    # BASELINE configs[0]: 10k x 64, M=16, efConstruction=100, one item after the other
    ix, ref, x = _pair(Index, 10000, 64, "sq_euclid", 64)
    assert ix.graph_hash() == ref.graph_hash()
    st = ix.exact_window_stats()
    assert st["rounds"] < 10000  # windows do link more than one item per round


def test_resizes_several_calls_and_removals_allowed(Index):
    # CollectionSize 256 forces doubling resizes; four Add calls; in-edge upkeep on (AllowRemovals default)
    ix, ref, x = _pair(Index, 4000, 48, "sq_euclid", 40, collection=256, calls=4)
    assert ix.graph_hash() == ref.graph_hash()
    rm = np.arange(0, 4000, 7, dtype=np.int32)
    ix.remove(rm); ref.remove(rm)
    assert ix.graph_hash() == ref.graph_hash()
    y = uniform(500, 48, 77)
    assert (ix.add(y) == ref.add(y)).all()      # slot reuse: the window is not used, the sequential path is
    assert ix.graph_hash() == ref.graph_hash()


def test_tie_heavy_integer_grid(Index):
    # equal distances everywhere: the sorted-list traversal hands most layers to the exact two-heap traversal,
    # whose expansions are logged the same way
    rng = np.random.default_rng(5)
    x = rng.integers(0, 3, size=(1500, 16)).astype(np.float32)
    ix, ref, _ = _pair(Index, 1500, 16, "sq_euclid", 24, M=8, efc=40, x=x)
    assert ix.graph_hash() == ref.graph_hash()


def test_odd_shapes(Index):
    for dim, M, efc in ((127, 5, 33), (8, 40, 90), (20, 12, 300)):
        ix, ref, _ = _pair(Index, 1200, dim, "sq_euclid", 16, M=M, efc=efc)
        assert ix.graph_hash() == ref.graph_hash(), (dim, M, efc)


def test_into_a_large_graph(Index):
    # 100k x 128 built with the default schedule, imported into the oracle; then 1 500 items through windows on the
    # GPU and one after the other on the CPU: equal hashes (the windows' conflict checks see real hub traffic here)
    n, dim, extra = 100_000, 128, 1500
    x = uniform(n + extra, dim, 65537)
    ix = Index(dim); ix.set_collection_size(n + extra); ix.set_max_candidates(200); ix.set_allow_removals(False)
    ix.add(x[:n])
    ref = oracle.OracleIndex(dim, max_edges=16, max_candidates=200, collection_size=n + extra, allow_removals=False)
    ref.import_graph(x[:n], ix.levels(), ix.entry_point, [ix.export_edges(l, 34) for l in range(int(ix.levels().max()) + 1)])
    ref.rng_skip(n)
    assert ref.graph_hash() == ix.graph_hash()
    ix.set_insert_batch_live(-64)
    a = ix.add(x[n:])
    b = ref.add(x[n:])
    assert (a == b).all() and ix.graph_hash() == ref.graph_hash()
    st = ix.exact_window_stats()
    assert st["linked"] / st["rounds"] > 1.5
