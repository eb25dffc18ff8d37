"""BASELINE config 5: int8-quantised rows with one float scale per row, squared Euclidean distance from the
exact integer dot product.  The reference has no counterpart (it is generic over TDistance,
/root/reference/src/HNSWIndex/HNSWIndex.cs:6, and ships float metrics): the metric is the builder's own
definition (oracle/hnsw_oracle.c "int8 rows"), so what these tests pin is (i) that definition against an
independent numpy statement of it and (ii) the HIP path against the oracle, bit for bit."""
import numpy as np
import pytest

import oracle
from common import default_cap, uniform

DIMS = [4, 7, 33, 96, 100, 120, 128, 200, 768]


def np_quantize(x):
    """scale = max|x| / 127 (float32 division); q = clip(rint(x / scale)) (float32 division, half-even); q = 0 when the scale is not positive."""
    x = np.asarray(x, dtype=np.float32)
    m = np.abs(x).max(axis=-1).astype(np.float32)
    scale = (m / np.float32(127.0)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.rint((x / scale[..., None]).astype(np.float32))
    q = np.where(scale[..., None] > 0, np.clip(q, -127, 127), 0).astype(np.int8)
    return q, scale, (q.astype(np.int64) ** 2).sum(axis=-1).astype(np.int32)


def np_distance(qa, sa, na, qb, sb, nb):
    """(float)((A + B) - 2 C) with A = (sa sa) na, B = (sb sb) nb, C = (sa sb) dot, in IEEE double."""
    dot = int((qa.astype(np.int64) * qb.astype(np.int64)).sum())
    sa, sb = np.float64(sa), np.float64(sb)
    A, B, C = (sa * sa) * np.float64(na), (sb * sb) * np.float64(nb), (sa * sb) * np.float64(dot)
    return np.float32((A + B) - 2.0 * C)


@pytest.mark.parametrize("dim", DIMS)
def test_quantiser_and_distance_follow_the_stated_definition(dim):
    rng = np.random.default_rng(dim)
    x = np.concatenate([uniform(40, dim, 1), (uniform(40, dim, 2) - 0.5) * np.float32(3.0), rng.standard_normal((20, dim)).astype(np.float32) * 1e-3])
    x[5] = 0.0                                        # a zero vector: scale 0, every q 0
    x[6, :] = 0.0; x[6, 0] = -2.5                     # one element carries the maximum: q = -127
    q, s, n = oracle.i8_quantize(x)
    wq, ws, wn = np_quantize(x)
    assert (q == wq).all() and s.tobytes() == ws.tobytes() and (n == wn).all()
    assert (q[5] == 0).all() and s[5] == 0 and q[6, 0] == -127 and np.abs(q).max() <= 127
    for a, b in ((0, 1), (3, 50), (41, 42), (5, 7), (5, 5), (6, 6), (90, 2)):
        got = oracle.metric("sq_euclid_i8", x[a], x[b])
        assert got.tobytes() == np_distance(q[a], s[a], n[a], q[b], s[b], n[b]).tobytes()
        # it IS the squared distance of the dequantised vectors (float64 check) ...
        da, db = q[a].astype(np.float64) * np.float64(s[a]), q[b].astype(np.float64) * np.float64(s[b])
        assert abs(float(got) - ((da - db) ** 2).sum()) <= 1e-5 * max(1.0, ((da - db) ** 2).sum())
        # ... and close to the float distance: the quantisation step is scale / 2 per element
        err = np.sqrt(dim) * 0.5 * float(s[a] + s[b])
        true = np.sqrt(((x[a].astype(np.float64) - x[b]) ** 2).sum())
        assert abs(np.sqrt(max(float(got), 0.0)) - true) <= err + 1e-6
    assert oracle.metric("sq_euclid_i8", x[9], x[9]) == 0    # identical records: exactly zero


@pytest.mark.parametrize("dim", DIMS)
def test_the_avx2_form_of_the_int8_distance_is_the_scalar_one_bit_for_bit(dim):
    # the CPU baselines time the AVX2 form (bench.py: use_avx=True); it must be the spec's scalar byte loop to the bit
    rng = np.random.default_rng(1000 + dim)
    x = np.concatenate([uniform(64, dim, 3), (uniform(64, dim, 4) - 0.5) * np.float32(5.0), rng.standard_normal((16, dim)).astype(np.float32)])
    x[3] = 0.0
    for a, b in [(0, 1), (2, 100), (3, 7), (64, 65), (130, 9), (143, 143)] + [tuple(rng.integers(0, 144, 2)) for _ in range(40)]:
        assert oracle.metric("sq_euclid_i8", x[a], x[b], use_avx=True).tobytes() == oracle.metric("sq_euclid_i8", x[a], x[b], use_avx=False).tobytes()


def test_int8_index_on_the_oracle():
    x = uniform(3000, 96, 11)
    ix = oracle.OracleIndex(96, "sq_euclid_i8", collection_size=3000)
    ids = ix.add(x)
    got, d = ix.knn_query(x, 1)
    assert (got[:, 0] == ids).mean() > 0.85 and (d[got[:, 0] == ids, 0] == 0).all()   # the reference's self-recall window, on int8 rows
    q = uniform(300, 96, 12)
    a_ids, a_d = ix.knn_query(q, 10)
    b_ids, b_d = ix.knn_query(q, 10, threads=4)
    assert (a_ids == b_ids).all() and a_d.tobytes() == b_d.tobytes()
    # every reported distance is the metric of (stored record, query record)
    for i in range(0, 300, 30):
        assert oracle.dist_query_rows("sq_euclid_i8", x, q[i], a_ids[i]).tobytes() == a_d[i].tobytes()
    # neighbours agree with the float index most of the time (quantisation error << typical gaps)
    fx = oracle.OracleIndex(96, "sq_euclid", collection_size=3000)
    fx.add(x)
    f_ids, _ = fx.knn_query(q, 10)
    overlap = np.mean([len(set(a_ids[i]) & set(f_ids[i])) / 10 for i in range(300)])
    assert overlap > 0.8
    # batched schedule (what hnsw_add does for count > 1), threaded or not: one graph
    h = set()
    for t in (1, 4):
        b = oracle.OracleIndex(96, "sq_euclid_i8", collection_size=3000, allow_removals=False)
        b.add_batched(x, 65536, threads=t)
        h.add(b.graph_hash())
    assert len(h) == 1


# ------------------------------------------------------------------ GPU tier
@pytest.fixture(scope="module")
def net():
    import hnswindex
    assert hnswindex.net_amd.lib.hnswdev_device_count() > 0, "GPU tier needs a HIP device"
    return hnswindex.net_amd


@pytest.mark.gpu
@pytest.mark.parametrize("dim", DIMS)
def test_hip_distances_bit_identical(net, dim):
    n = 3000
    rows = np.concatenate([uniform(n // 2, dim, 31), (uniform(n - n // 2, dim, 32) - 0.5) * np.float32(2.0)])
    rows[17] = 0.0
    q = np.concatenate([uniform(20, dim, 33), uniform(20, dim, 34) - 0.5])
    dev = net.DeviceBackend(dim, "sq_euclid_i8", capacity=100)
    dev.upload_rows(0, rows[:100])
    dev.reserve(n)                                            # records survive the resize
    dev.upload_rows(100, rows[100:])
    # stored records: the dequantised rows come back as q * scale
    qq, ss, _ = oracle.i8_quantize(rows[:200])
    assert dev.download_rows(0, 200).tobytes() == (qq.astype(np.float32) * ss[:, None]).astype(np.float32).tobytes()
    rng = np.random.default_rng(dim)
    counts = rng.integers(0, 70, q.shape[0])
    counts[0], counts[1] = 0, 65
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    ids = rng.integers(0, n, off[-1]).astype(np.int32)
    ids[:5] = 17
    got = dev.dist_query_batch(q, off, ids)
    for i in range(q.shape[0]):
        want = oracle.dist_query_rows("sq_euclid_i8", rows, q[i], ids[off[i]:off[i + 1]])
        assert want.tobytes() == got[off[i]:off[i + 1]].tobytes()
    a, b = rng.integers(0, n, 5000).astype(np.int32), rng.integers(0, n, 5000).astype(np.int32)
    a[:10] = b[:10]
    pg = dev.dist_pair_batch(a, b)
    assert pg.tobytes() == oracle.dist_pairs("sq_euclid_i8", rows, a, b).tobytes()
    assert (pg[:10] == 0).all() and pg.tobytes() == dev.dist_pair_batch(b, a).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("traversal", ["device", "host"])
@pytest.mark.parametrize("dim,M,efc", [(96, 16, 100), (40, 8, 60), (128, 12, 80)])
def test_hip_index_matches_oracle(net, traversal, dim, M, efc):
    import hnswindex
    n = 4000
    x = uniform(n, dim, 41) if dim != 40 else uniform(n, dim, 41) - 0.5
    q = uniform(400, dim, 42) if dim != 40 else uniform(400, dim, 42) - 0.5
    ref = oracle.OracleIndex(dim, "sq_euclid_i8", max_edges=M, max_candidates=efc, min_nn=40, collection_size=1024)
    ix = hnswindex.Index(dim, "sq_euclid_i8")
    ix.set_max_edges(M); ix.set_max_candidates(efc); ix.set_min_nn(40); ix.set_collection_size(1024)
    ix.set_insert_batch(1); ix.set_device_traversal(traversal == "device")
    assert (ix.add(x[:1500]) == ref.add(x[:1500])).all()      # sequential, with two capacity doublings
    assert ix.graph_hash() == ref.graph_hash() and (ix.levels() == ref.levels()).all()
    for k in (1, 10, 40):
        a_ids, a_d = ix.knn_query(q, k)
        b_ids, b_d = ref.knn_query(q, k)
        assert (a_ids == b_ids).all() and a_d.tobytes() == b_d.tobytes()
    # the default (snapshot-batched) schedule on top
    ib = hnswindex.Index(dim, "sq_euclid_i8")
    ib.set_max_edges(M); ib.set_max_candidates(efc); ib.set_min_nn(40); ib.set_collection_size(n); ib.set_allow_removals(False)
    ib.set_device_traversal(traversal == "device")
    rb = oracle.OracleIndex(dim, "sq_euclid_i8", max_edges=M, max_candidates=efc, min_nn=40, collection_size=n, allow_removals=False)
    ib.add(x); rb.add_batched(x, default_cap())
    assert ib.graph_hash() == rb.graph_hash()
    a_ids, a_d = ib.knn_query(q, 10)
    b_ids, b_d = rb.knn_query(q, 10)
    assert (a_ids == b_ids).all() and a_d.tobytes() == b_d.tobytes()
    # range query and removal run on the same records
    r_ids, r_d = ib.range_query(q[:20], float(np.median(a_d[:, 3])))
    o_ids, o_d = rb.range_query(q[:20], float(np.median(a_d[:, 3])))
    assert all((r_ids[i] == o_ids[i]).all() and r_d[i].tobytes() == o_d[i].tobytes() for i in range(20))
    with pytest.raises(RuntimeError, match="NotSupported"):
        ib.serialize("/tmp/never_written.bin")
