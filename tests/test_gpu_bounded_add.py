"""GPU tier: Add under BOUNDED concurrency.  B consecutive items searching one snapshot and linking in id order is an
interleaving the reference's HNSWIndex.Add(List) -- a Parallel.For over the items, src/HNSWIndex/HNSWIndex.cs:70-78 -- can
produce iff B <= the threads it runs on.  The default cap is therefore the host's hardware threads
(hnsw_mi355x_host_parallelism), larger snapshots are opt-in, and every rung of the ladder bench.py reports must equal the
CPU restatement of the same schedule (oracle add_batched(x, B)) bit for bit: levels, adjacency lists, graph hash, answers."""
import numpy as np
import pytest

import oracle
from common import default_cap, normalize_f32, uniform

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Index():
    import hnswindex
    assert hnswindex.net_amd.lib.hnswdev_device_count() > 0, "GPU tier needs a HIP device"
    return hnswindex.Index


def _same_graph(ix, ref):
    assert (ix.levels() == ref.levels()).all()
    assert ix.entry_point == ref.entry_point
    assert ix.graph_hash() == ref.graph_hash()


@pytest.mark.parametrize("B", [16, 64, 256, 1024])
def test_one_call_under_cap_B_equals_the_cpu_restatement(Index, B):
    n, dim = 14000, 32                                      # the batches reach the cap once linked / 4 >= B
    x, q = uniform(n, dim, 500 + B), uniform(300, dim, 77)
    ix = Index(dim); ix.set_collection_size(n); ix.set_max_candidates(60); ix.set_min_nn(40); ix.set_insert_batch(B)
    ids = ix.add(x)
    assert ix.insert_batch_cap == B
    ref = oracle.OracleIndex(dim, max_candidates=60, min_nn=40, collection_size=n)
    assert (ids == ref.add_batched(x, B)).all()
    _same_graph(ix, ref)
    a, b = ix.knn_query(q, 10), ref.knn_query(q, 10)
    assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()


@pytest.mark.parametrize("metric,B", [("cosine", 64), ("ucosine", 256), ("sq_euclid_i8", 256)])
def test_calls_of_B_items_equal_the_cpu_restatement(Index, metric, B):
    # what a host does that hands the export T items at a time: every call is one snapshot batch (smaller while the graph is small)
    n, dim = 6000, 48
    x = uniform(n, dim, 91)
    if metric == "ucosine":
        x = normalize_f32(x)
    ix = Index(dim, metric); ix.set_collection_size(n); ix.set_max_candidates(50); ix.set_insert_batch(B)
    ref = oracle.OracleIndex(dim, metric, max_candidates=50, collection_size=n)
    for i in range(0, n, B):
        assert (ix.add(x[i:i + B]) == ref.add_batched(x[i:i + B], B)).all()
    _same_graph(ix, ref)


def test_the_default_cap_is_the_hosts_hardware_threads(Index):
    import os
    import hnswindex
    T = hnswindex.net_amd.host_parallelism()
    assert T == default_cap() and 1 <= T <= max(1, len(os.sched_getaffinity(0)))
    n, dim = 5000 + 20 * T, 24
    x = uniform(n, dim, 17)
    ix = Index(dim); ix.set_collection_size(n)
    assert ix.insert_batch_cap == T                          # nothing set: pending default
    ix.add(x[:n // 2])
    assert ix.insert_batch_cap == T
    ref = oracle.OracleIndex(dim, collection_size=n)
    ref.add_batched(x[:n // 2], T)
    _same_graph(ix, ref)
    # opt-in: the large snapshots of rounds 1-4, then back to the default (0) on the same index
    ix.set_insert_batch_live(65536); assert ix.insert_batch_cap == 65536
    ix.add(x[n // 2:n // 2 + 1500]); ref.add_batched(x[n // 2:n // 2 + 1500], 65536)
    _same_graph(ix, ref)
    ix.set_insert_batch_live(0); assert ix.insert_batch_cap == T
    ix.add(x[n // 2 + 1500:]); ref.add_batched(x[n // 2 + 1500:], T)
    _same_graph(ix, ref)


def test_bounded_batches_with_removals_allowed_and_resizes(Index):
    # default CollectionSize growth (doubling) and in-edge upkeep under a small cap
    n, dim, B = 3000, 16, 16
    x = uniform(n, dim, 3)
    ix = Index(dim); ix.set_collection_size(64); ix.set_insert_batch(B)
    ref = oracle.OracleIndex(dim, collection_size=64)
    ix.add(x[:2000]); ref.add_batched(x[:2000], B)
    rm = list(range(100, 160))
    ix.remove(rm); ref.remove(rm)
    ix.add(x[2000:]); ref.add_batched(x[2000:], B)
    assert ix.graph_hash() == ref.graph_hash() and ix.entry_point == ref.entry_point
    assert np.array_equal(np.sort(ix.ids()), np.sort(ref.active_ids()))
