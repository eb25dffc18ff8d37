"""GPU tier: HNSWIndex.Serialize / Deserialize (src/HNSWIndex/HNSWIndex.cs:210-229) through the
C ABI (`hnsw_mi355x_serialize` / `hnsw_mi355x_deserialize`).  Mirrors
GraphSerializationTests.EncodeDecodeTest (src/HNSWIndex.Tests/GraphSerializationTests.cs:17-49)
and loads snapshots of oracle-built graphs written by the test-side encoder (tests/pbnet.py)."""
import numpy as np
import pytest

import oracle
import pbnet
from common import normalize_f32, uniform
from test_snapshot_codec import oracle_snapshot

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Index():
    import hnswindex
    assert hnswindex.net_amd.lib.hnswdev_device_count() > 0, "GPU tier needs a HIP device"
    return hnswindex.Index


@pytest.mark.parametrize("traversal", ["device", "host"])
def test_encode_decode(Index, tmp_path, traversal):
    # GraphSerializationTests.cs:17-49: 2 000 x 128, add one at a time, every vector queried with k=5
    x = uniform(2000, 128, 65537)
    ix = Index(128)
    ix.set_device_traversal(traversal == "device")
    ix.set_insert_batch(1)
    ix.add(x)
    path = tmp_path / "index.bin"
    ix.serialize(path)
    Index(1).set_device_traversal(traversal == "device")
    back = Index.deserialize(path)
    assert back.dim == 128 and back.count == 2000 and back.graph_hash() == ix.graph_hash()
    assert back.entry_point == ix.entry_point and (back.ids() == ix.ids()).all()
    a, b = ix.knn_query(x, 5), back.knn_query(x, 5)
    assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()
    # the file is what the test-side decoder expects of the reference's contracts
    got = pbnet.decode(path.read_bytes())
    assert got["length"] == 2000 and got["count"] == 2000 and got["capacity"] == 65536 and got["entry"] == ix.entry_point
    assert np.array(got["items"], dtype=np.float32).tobytes() == x.tobytes()
    assert [len(nd["out"]) - 1 for nd in got["nodes"]] == ix.levels().tolist()


@pytest.mark.parametrize("metric", ["sq_euclid", "cosine", "ucosine"])
@pytest.mark.parametrize("packed", [False, True])
def test_loading_a_graph_built_elsewhere(Index, tmp_path, metric, packed):
    # the graph comes from the CPU restatement of the reference, through the reference's wire format
    params = dict(max_edges=8, max_candidates=60, min_nn=20, collection_size=1024, random_seed=99)
    x, q = uniform(700, 24, 5), uniform(150, 24, 6)
    if metric == "ucosine":
        x, q = normalize_f32(x), normalize_f32(q)
    ref = oracle.OracleIndex(24, metric, **params)
    ref.add(x)
    path = tmp_path / "ref.bin"
    path.write_bytes(oracle_snapshot(ref, x, params, packed=packed))
    for traversal in ("device", "host"):
        Index(1).set_device_traversal(traversal == "device")
        ix = Index.deserialize(path, metric)
        assert ix.graph_hash() == ref.graph_hash()
        want, got = ref.knn_query(q, 10), ix.knn_query(q, 10)
        assert (got[0] == want[0]).all() and got[1].tobytes() == want[1].tobytes()
        r_want, r_got = ref.range_query(q[:20], 0.9 if metric == "sq_euclid" else 0.1), ix.range_query(q[:20], 0.9 if metric == "sq_euclid" else 0.1)
        for (wi, wd), (gi, gd) in zip(zip(*r_want), zip(*r_got)):
            assert (np.asarray(wi) == np.asarray(gi)).all() and np.asarray(wd, dtype=np.float32).tobytes() == np.asarray(gd, dtype=np.float32).tobytes()


def test_add_after_load_restarts_the_level_sequence(Index, tmp_path):
    # GraphData's snapshot constructor builds a fresh Random(RandomSeed) (GraphData.cs:61)
    params = dict(max_edges=8, max_candidates=50, collection_size=2048, random_seed=777, allow_removals=False)
    x, more = uniform(600, 16, 11), uniform(300, 16, 12)
    ref = oracle.OracleIndex(16, **params)
    ref.add(x)
    path = tmp_path / "ref.bin"
    path.write_bytes(oracle_snapshot(ref, x, params))
    ix = Index.deserialize(path)
    Index(1).set_insert_batch(1)
    ix2 = Index.deserialize(path)
    # the same thing on the oracle: a fresh index (fresh RNG) given the saved graph, then Add
    lv = ix.levels()
    layers = [ix.export_edges(L, 18 if L == 0 else 10) for L in range(int(lv.max()) + 1)]
    cont = oracle.OracleIndex(16, **params)
    cont.import_graph(x, lv, ix.entry_point, layers)
    cont.add(more)
    ids = ix2.add(more)
    assert (ids == np.arange(600, 900)).all()
    assert ix2.graph_hash() == cont.graph_hash()
    assert ix2.levels()[600:].tolist() == oracle.random_levels(777, 1 / np.log(16), 300).tolist()


def test_removals_survive_a_round_trip(Index, tmp_path):
    x = uniform(900, 20, 31)
    q = uniform(100, 20, 32)
    a = Index(20)
    a.set_max_edges(6); a.set_max_candidates(40); a.set_collection_size(1024)
    a.add(x)
    a.remove([5, 700, 33, 34, 35])
    path = tmp_path / "a.bin"
    a.serialize(path)
    b = Index.deserialize(path)
    assert b.graph_hash() == a.graph_hash() and (b.ids() == a.ids()).all() and b.count == 895
    ra, rb = a.knn_query(q, 7), b.knn_query(q, 7)
    assert (ra[0] == rb[0]).all() and ra[1].tobytes() == rb[1].tobytes()
    # in-edge sets are rebuilt from the out-lists, so further removals agree too
    more = [1, 2, 3, 899, 450]
    a.remove(more); b.remove(more)
    assert b.graph_hash() == a.graph_hash() and (b.ids() == a.ids()).all()
    # vacated slots are reused in the same (LIFO) order
    extra = uniform(3, 20, 33)
    assert a.add(extra[:1]).tolist() == b.add(extra[:1]).tolist() == [450]


def test_errors(Index, tmp_path):
    with pytest.raises(RuntimeError, match="FileNotFound"):
        Index.deserialize(tmp_path / "missing.bin")
    (tmp_path / "junk.bin").write_bytes(b"\x0a\x05hello")
    with pytest.raises(RuntimeError, match="invalid snapshot|cannot be null"):
        Index.deserialize(tmp_path / "junk.bin")
    with pytest.raises(RuntimeError, match="Unsupported distance metric"):
        Index.deserialize(tmp_path / "junk.bin", "manhattan")
    ix = Index(8)
    ix.add(uniform(10, 8, 1))
    with pytest.raises(RuntimeError, match="IOException"):
        ix.serialize(tmp_path / "no_such_dir" / "x.bin")
