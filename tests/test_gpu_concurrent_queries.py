"""GPU tier: hnsw_knn_query calls on ONE handle from several host threads overlap (the reference's contract: operations
of one type may run concurrently, /root/reference/README.md:64-65; GraphTests.QueryGraphMultiThread,
/root/reference/src/HNSWIndex.Tests/GraphTests.cs:82-120) -- each call on a query lane of its own -- and return exactly
what one thread gets."""
import threading

import numpy as np
import pytest

from common import normalize_f32, uniform

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Index():
    import hnswindex
    assert hnswindex.net_amd.lib.hnswdev_device_count() > 0, "GPU tier needs a HIP device"
    return hnswindex.Index


@pytest.mark.parametrize("metric", ["sq_euclid", "cosine", "sq_euclid_i8"])
def test_threads_get_the_single_thread_answers(Index, metric):
    n, dim, T = 20000, 64, 6
    x = uniform(n, dim, 31)
    ix = Index(dim, metric); ix.set_collection_size(n); ix.set_min_nn(48)
    ix.add(x)
    sets = [uniform(300 + 1700 * t, dim, 100 + t) for t in range(T)]      # different sizes: different launch lengths
    sets[3] = uniform(9000, dim, 777)                                       # one large enough for the streamed upload
    want = [ix.knn_query(q, 10) for q in sets]
    got = [None] * T
    errs = []

    def worker(t):
        try:
            for _ in range(4):
                got[t] = ix.knn_query(sets[t], 10)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs
    for t in range(T):
        assert (got[t][0] == want[t][0]).all() and got[t][1].tobytes() == want[t][1].tobytes()
    st = ix.stats()
    assert st["search_launches"] >= T * 4


def test_queries_and_adds_interleave_safely(Index):
    # mixed types are not promised to overlap by the reference; here they are simply serialised (Add takes the handle
    # exclusively) and every query sees a consistent graph
    n, dim = 6000, 32
    x = uniform(n + 3000, dim, 5)
    ix = Index(dim); ix.set_collection_size(n + 3000)
    ix.add(x[:n])
    q = uniform(500, dim, 6)
    stop = threading.Event()
    bad = []

    def querier():
        while not stop.is_set():
            ids, d = ix.knn_query(q, 5)
            if not ((ids >= 0).all() and (np.diff(d, axis=1) >= 0).all()):
                bad.append(1)
    th = [threading.Thread(target=querier) for _ in range(3)]
    for t in th:
        t.start()
    for i in range(0, 3000, 500):
        ix.add(x[n + i:n + i + 500])
    stop.set()
    for t in th:
        t.join()
    assert not bad and ix.count == n + 3000
    one = Index(dim); one.set_collection_size(n + 3000)
    one.add(x[:n])
    for i in range(0, 3000, 500):
        one.add(x[n + i:n + i + 500])
    assert one.graph_hash() == ix.graph_hash()
    a, b = one.knn_query(q, 5), ix.knn_query(q, 5)
    assert (a[0] == b[0]).all()
