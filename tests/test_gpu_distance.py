"""GPU tier: the HIP distance kernels, through the C ABI (hnswdev_*), must be BIT-IDENTICAL
to the oracle's lane-ordered float32 arithmetic (north_star tolerance is 1e-5; these
kernels are held to 0 ulp) for every metric, including dim % 8 != 0 tails."""
import numpy as np
import pytest

import oracle
from common import normalize_f32, uniform

pytestmark = pytest.mark.gpu

METRICS = ["sq_euclid", "cosine", "ucosine"]
DIMS = [1, 5, 8, 9, 64, 96, 127, 128, 768]


@pytest.fixture(scope="module")
def net():
    import hnswindex
    assert hnswindex.net_amd.lib.hnswdev_device_count() > 0, "GPU tier needs a HIP device"
    return hnswindex.net_amd


@pytest.mark.parametrize("dim", DIMS)
@pytest.mark.parametrize("metric", METRICS)
def test_query_batch_bit_identical(net, metric, dim):
    n, nq = 3000, 37
    rows, q = uniform(n, dim, 100 + dim), uniform(nq, dim, 200 + dim)
    if metric == "ucosine":
        rows, q = normalize_f32(rows), normalize_f32(q)
    rng = np.random.default_rng(dim)
    counts = rng.integers(0, 150, nq)          # ragged, some empty, some > one slot (64)
    counts[0], counts[1] = 0, 1
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    ids = rng.integers(0, n, off[-1]).astype(np.int32)
    dev = net.DeviceBackend(dim, metric, capacity=n)
    dev.upload_rows(0, rows)
    got = dev.dist_query_batch(q, off, ids)
    want = np.concatenate([oracle.dist_query_rows(metric, rows, q[i], ids[off[i]:off[i + 1]]) for i in range(nq)])
    assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist()
    assert dev.stats()["evals"] == int(off[-1])


@pytest.mark.parametrize("dim", [7, 64, 128, 768])
@pytest.mark.parametrize("metric", METRICS)
def test_pair_batch_bit_identical(net, metric, dim):
    n = 2000
    rows = uniform(n, dim, 300 + dim)
    if metric == "ucosine":
        rows = normalize_f32(rows)
    rng = np.random.default_rng(dim + 1)
    a, b = rng.integers(0, n, 5001).astype(np.int32), rng.integers(0, n, 5001).astype(np.int32)
    a[:10] = b[:10]                             # identical rows: distance exactly 0 for sq_euclid
    dev = net.DeviceBackend(dim, metric, capacity=n)
    dev.upload_rows(0, rows)
    got = dev.dist_pair_batch(a, b)
    want = oracle.dist_pairs(metric, rows, a, b)
    assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist()
    if metric == "sq_euclid":
        assert (got[:10] == 0).all()


def test_special_values_and_denormals(net):
    dim = 128
    rows = uniform(64, dim, 1)
    rows[0] = 0.0                               # zero vector: cosine denominator guard
    rows[1] = 1e-30                             # squares underflow to denormals / zero
    rows[2] = 3e-20
    rows[3, ::2] = -rows[3, ::2]
    rows[4] = 1e18                              # squares overflow to +inf in float32
    q = rows[:8].copy()
    ids = np.tile(np.arange(8, dtype=np.int32), 8)
    off = (np.arange(9) * 8).astype(np.int32)
    for metric in METRICS:
        dev = net.DeviceBackend(dim, metric, capacity=64)
        dev.upload_rows(0, rows)
        got = dev.dist_query_batch(q, off, ids)
        want = np.concatenate([oracle.dist_query_rows(metric, rows, q[i], ids[off[i]:off[i + 1]]) for i in range(8)])
        assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist(), metric


def test_device_double_sqrt_is_correctly_rounded(net):
    # Math.Sqrt in CosineMetric.cs:88 is IEEE correctly rounded; the device's must be too
    rng = np.random.default_rng(7)
    x = np.concatenate([
        rng.random(200000) * 100, 10.0 ** rng.uniform(-60, 60, 200000),
        np.float32(rng.random(200000) * 50).astype(np.float64),
        [0.0, 1.0, 2.0, 4.0, 1e-300, 1e300, np.inf, 2.0 ** -1060],
        np.nextafter(np.arange(1, 2000, dtype=np.float64) ** 2, 0), np.nextafter(np.arange(1, 2000, dtype=np.float64) ** 2, np.inf),
    ])
    got = net.bindings.device_sqrt_rn(x)
    assert got.view(np.uint64).tolist() == np.sqrt(x).view(np.uint64).tolist()


def test_rows_round_trip_and_reserve_growth(net):
    dim = 64
    rows = uniform(5000, dim, 5)
    dev = net.DeviceBackend(dim, "sq_euclid", capacity=100)
    dev.upload_rows(0, rows[:100])
    dev.reserve(5000)                           # contents must survive the doubling resize
    dev.upload_rows(100, rows[100:])
    assert dev.download_rows(0, 5000).tobytes() == rows.tobytes()
    with pytest.raises(RuntimeError, match="capacity"):
        dev.upload_rows(4999, rows[:2])


def test_bad_ids_are_refused_not_dereferenced(net):
    # ids are guarded inside the kernels: a bad one yields NaN and an error return, never a fault
    dev = net.DeviceBackend(16, "sq_euclid", capacity=10)
    rows = uniform(10, 16, 1)
    dev.upload_rows(0, rows)
    with pytest.raises(RuntimeError, match="outside the uploaded data"):
        dev.dist_query_batch(uniform(1, 16, 2), np.array([0, 1], np.int32), np.array([10], np.int32))
    with pytest.raises(RuntimeError, match="outside uploaded rows"):
        dev.dist_pair_batch(np.array([0], np.int32), np.array([-1], np.int32))
    # the context keeps working, and the error belongs to it alone
    other = net.DeviceBackend(16, "sq_euclid", capacity=10)
    assert other.last_error() == "" and "outside uploaded rows" in dev.last_error()
    q = uniform(1, 16, 2)
    got = dev.dist_query_batch(q, np.array([0, 3], np.int32), np.array([9, 0, 4], np.int32))
    assert got.tobytes() == oracle.dist_query_rows("sq_euclid", rows, q[0], [9, 0, 4]).tobytes()


@pytest.mark.parametrize("metric", METRICS)
def test_step_api_double_buffered(net, metric):
    # The inner boundary as a foreign host drives it: queries uploaded once, two context-owned
    # pinned buffer sets, submit / wait, one set in flight while the other is filled.
    n, dim, nq, stride, nslots = 4000, 96, 64, 40, 512
    rows, q = uniform(n, dim, 21), uniform(nq, dim, 22)
    if metric == "ucosine":
        rows, q = normalize_f32(rows), normalize_f32(q)
    dev = net.DeviceBackend(dim, metric, capacity=n)
    dev.upload_rows(0, rows)
    dev.set_queries(q)
    rng = np.random.default_rng(5)
    sets = [dev.step_buffers(g, nslots, stride) for g in (0, 1)]
    addr = [(r.ctypes.data, d.ctypes.data) for r, d in sets]
    plans = []
    for step in range(6):
        g = step & 1
        rec, dist = sets[g]
        if step >= 2:       # this set's previous step must have landed before it is refilled
            dev.step_wait(g)
        used = int(rng.integers(1, nslots + 1))
        cnt = rng.integers(0, stride + 1, used).astype(np.int32)
        qidx = rng.integers(0, nq, used).astype(np.int32)
        pair = rng.random(used) < 0.3                      # some slots measure row <-> row (Distance(int, int))
        qidx[pair] = ~rng.integers(0, n, int(pair.sum())).astype(np.int32)
        ids = rng.integers(0, n, (used, stride)).astype(np.int32)
        rec[:used, 0], rec[:used, 1], rec[:used, 2:] = cnt, qidx, ids
        dev.step_submit(g, used)
        plans.append((g, used, cnt.copy(), qidx.copy(), ids.copy()))
        if step >= 1:       # verify the OTHER set's step while this one is in flight
            pg, pused, pcnt, pq, pids = plans[step - 1]
            dev.step_wait(pg)
            pdist = sets[pg][1]
            for s_ in range(0, pused, 7):
                v = rows[~pq[s_]] if pq[s_] < 0 else q[pq[s_]]
                want = oracle.dist_query_rows(metric, rows, v, pids[s_, :pcnt[s_]])
                assert want.tobytes() == pdist[s_, :pcnt[s_]].tobytes()
    dev.step_wait(0); dev.step_wait(1)
    # nothing was reallocated: the same pinned buffers every time
    assert [(r.ctypes.data, d.ctypes.data) for r, d in (dev.step_buffers(g, nslots, stride) for g in (0, 1))] == addr
    # the resident query set serves dist_query_batch without a re-upload
    off = np.arange(0, 5 * 30 + 1, 30, dtype=np.int32)
    cid = rng.integers(0, n, 150).astype(np.int32)
    got = dev.dist_query_batch(None, off, cid)
    for i in range(5):
        assert got[30 * i:30 * i + 30].tobytes() == oracle.dist_query_rows(metric, rows, q[i], cid[30 * i:30 * i + 30]).tobytes()
    # guards: a record naming a row that was never uploaded comes back NaN, wait reports it
    rec, dist = sets[0]
    rec[0, 0], rec[0, 1], rec[0, 2:4] = 2, 0, [5, n + 7]
    dev.step_submit(0, 1)
    with pytest.raises(RuntimeError, match="outside the uploaded data"):
        dev.step_wait(0)
    assert np.isnan(dist[0, 1]) and dist[0, 0] == oracle.dist_query_rows(metric, rows, q[0], [5])[0]
    rec[0, 0] = stride + 1                                  # more ids than the slot holds
    dev.step_submit(0, 1)
    with pytest.raises(RuntimeError):
        dev.step_wait(0)
    with pytest.raises(RuntimeError, match="already in flight|more slots"):
        dev.step_submit(0, nslots + 1)


def test_linearity_style_properties_at_full_row_size(net):
    # size-independent properties at the BASELINE row shape (dim 128): symmetry and
    # d(x,x) == 0, over a matrix far larger than any cache-resident toy
    n, dim = 200000, 128
    rows = uniform(n, dim, 9)
    dev = net.DeviceBackend(dim, "sq_euclid", capacity=n)
    dev.upload_rows(0, rows)
    rng = np.random.default_rng(3)
    a, b = rng.integers(0, n, 100000).astype(np.int32), rng.integers(0, n, 100000).astype(np.int32)
    ab, ba = dev.dist_pair_batch(a, b), dev.dist_pair_batch(b, a)
    assert ab.tobytes() == ba.tobytes()
    assert (dev.dist_pair_batch(a, a) == 0).all()
    sample = slice(0, 2000)
    assert ab[sample].tobytes() == oracle.dist_pairs("sq_euclid", rows, a[sample], b[sample]).tobytes()


@pytest.mark.parametrize("metric", METRICS)
def test_knn_search_on_a_host_supplied_graph(net, metric):
    # inner boundary, graph-resident form: a host (here: the oracle) owns the graph, hands it over
    # layer by layer, and gets KnnQuery results identical to its own CPU traversal
    n, dim, M = 5000, 64, 12
    x, q = uniform(n, dim, 401), uniform(300, dim, 402)
    if metric == "ucosine":
        x, q = normalize_f32(x), normalize_f32(q)
    ref = oracle.OracleIndex(dim, metric, max_edges=M, max_candidates=80, min_nn=48, collection_size=n)
    ref.add(x)
    lv = ref.levels()
    layers = []
    for L in range(int(lv.max()) + 1):
        stride = 2 * M + 2
        counts = np.full(n, -1, np.int32)
        edges = np.zeros((n, stride), np.int32)
        for i in np.nonzero(lv >= L)[0]:
            e = ref.edges(int(i), L)
            counts[i] = e.size
            edges[i, :e.size] = e
        layers.append((counts, edges))
    dev = net.DeviceBackend(dim, metric, capacity=n)
    dev.upload_rows(0, x)
    dev.set_graph(lv, layers, M)
    ids, d, flags = dev.knn_search(q, ref.entry_point, 48, 10)
    want_ids, want_d = ref.knn_query(q, 10)
    assert (flags == 0).all()
    assert (ids == want_ids).all() and d.tobytes() == want_d.tobytes()
    with pytest.raises(RuntimeError, match="bad argument"):
        dev.knn_search(q, n + 5, 48, 10)
    # RangeQuery through the same boundary: SearchLayerRange on the device, ascending by distance
    radius = {"sq_euclid": 6.5, "cosine": 0.16, "ucosine": 0.16}[metric]
    rids, rd, rflags = dev.range_search(q, ref.entry_point, radius)
    want_rids, want_rd = ref.range_query(q, radius)
    # (results of equal distance -- 1 - dot rounds many pairs together -- come in the order of the reference's heap array)
    assert (rflags == 0).all() and sum(len(a) for a in rids) > 100
    for a, b, c, e in zip(rids, rd, want_rids, want_rd):
        assert a.tolist() == c.tolist() and b.tobytes() == e.tobytes()
    st = dev.stats()
    assert st["range_launches"] == 1 and st["range_handbacks"] == 0
    with pytest.raises(RuntimeError, match="bad argument"):
        dev.range_search(q, -1, radius)
