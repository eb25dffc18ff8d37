"""CPU tier: the Serialize / Deserialize wire codec (csrc/snapshot_io.h) against the test-side
restatement in tests/pbnet.py, on graphs built by the oracle.  No device involved: the library's
`hnswhost_test_snapshot_transcode` hook decodes a file and re-encodes it.

Reference: HNSWIndex.Serialize / Deserialize (src/HNSWIndex/HNSWIndex.cs:210-229) and
GraphSerializationTests.EncodeDecodeTest (src/HNSWIndex.Tests/GraphSerializationTests.cs:17-49).
No serialized fixture exists in the reference and protobuf-net cannot run here: the format is
pinned by the protocol-buffers spec and the contracts only ("parity unpinned")."""
import ctypes as ct

import numpy as np
import pytest

import oracle
import pbnet
from common import uniform


@pytest.fixture(scope="module")
def lib():
    import hnswindex
    L = hnswindex.net_amd.lib
    L.hnswhost_test_snapshot_transcode.argtypes = [ct.c_char_p, ct.c_char_p, ct.POINTER(ct.c_int), ct.POINTER(ct.c_uint64)]
    L.hnswhost_test_snapshot_transcode.restype = ct.c_int
    return L


def transcode(lib, src, dst):
    import hnswindex
    info = (ct.c_int * 8)()
    h = ct.c_uint64(0)
    rc = lib.hnswhost_test_snapshot_transcode(str(src).encode(), str(dst).encode() if dst else None, info, ct.byref(h))
    if rc < 0:
        raise RuntimeError(hnswindex.net_amd.last_error())
    return dict(zip("length dim count entry capacity max_edges allow_removals random_seed".split(), info)), h.value


def oracle_snapshot(ref, x, params, removed=(), capacity=None, packed=False, garbage=True, **kw):
    """What the reference's GraphDataSnapshot holds for the oracle's graph."""
    M = params.get("max_edges", 16)
    n = ref.length
    rng = np.random.default_rng(9)
    nodes = []
    for i in range(n):
        lists = {False: [], True: []}
        for incoming in (False, True):
            if incoming and not params.get("allow_removals", True):
                continue
            for l in range(ref.max_layer(i) + 1):
                e = ref.edges(i, l, incoming=incoming).tolist()
                cap = max(len(e), (2 * M if l == 0 else M) + 1)
                # EdgeList.Buffer is written whole; what lies beyond Count is stale (Node.cs:80-87)
                tail = rng.integers(0, n, cap - len(e)).tolist() if garbage else [0] * (cap - len(e))
                lists[incoming].append((e + tail, len(e)))
        nodes.append(dict(id=i, removed=i in removed, out=lists[False], inn=lists[True]))
    active = ref.active_ids().tolist()
    return pbnet.encode(params, nodes, active, x[:n].tolist(), list(removed)[::-1], ref.entry_point,
                        capacity if capacity is not None else max(n, params.get("collection_size", 65536)), n, len(active),
                        packed=packed, **kw)


def build(n=300, dim=12, seed=7, **params):
    x = uniform(n, dim, seed)
    ref = oracle.OracleIndex(dim, max_edges=params.get("max_edges", 16), max_candidates=params.get("max_candidates", 100),
                             allow_removals=params.get("allow_removals", True), random_seed=params.get("random_seed", 31337),
                             collection_size=params.get("collection_size", 65536))
    ref.add(x)
    return ref, x


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("allow_removals", [True, False])
def test_decode_then_encode_preserves_the_graph(lib, tmp_path, packed, allow_removals):
    params = dict(max_edges=6, max_candidates=40, min_nn=9, remove_max_candidates=33, collection_size=512, random_seed=4242,
                  allow_removals=allow_removals)
    ref, x = build(**params)
    src, dst = tmp_path / "a.bin", tmp_path / "b.bin"
    src.write_bytes(oracle_snapshot(ref, x, params, packed=packed))
    info, h = transcode(lib, src, dst)
    assert info == dict(length=300, dim=12, count=300, entry=ref.entry_point, capacity=512, max_edges=6,
                        allow_removals=1, random_seed=4242) or not allow_removals
    # AllowRemovals=false is a zero value: never on the wire, so it reloads as the initialiser `true`
    assert info["allow_removals"] == 1
    assert h == ref.graph_hash()
    got = pbnet.decode(dst.read_bytes())
    assert got["params"] == {**pbnet.DEFAULT_PARAMS, **params, "allow_removals": True}
    assert (got["entry"], got["capacity"], got["length"], got["count"]) == (ref.entry_point, 512, 300, 300)
    assert got["active"] == ref.active_ids().tolist() and got["removed"] == []
    assert np.array(got["items"], dtype=np.float32).tobytes() == x.tobytes()
    assert got["repeated_wire_types"] == {pbnet.VARINT, pbnet.FIXED32}  # protobuf-net default: not packed
    for i, nd in enumerate(got["nodes"]):
        assert nd["id"] == i and not nd["removed"] and len(nd["out"]) == ref.max_layer(i) + 1
        for l, (buf, cnt) in enumerate(nd["out"]):
            assert buf[:cnt] == ref.edges(i, l).tolist()
            assert len(buf) >= (12 if l == 0 else 6) + 1  # never a null Buffer for the reference (Node.cs:69)
        # the decoder reloaded AllowRemovals as true, so in-edge lists are written (GraphData.cs:227)
        assert len(nd["inn"]) == len(nd["out"])
        for l, (buf, cnt) in enumerate(nd["inn"]):
            assert sorted(buf[:cnt]) == sorted(ref.edges(i, l, incoming=True).tolist()) or not allow_removals
    # a second pass is a fixed point
    dst2 = tmp_path / "c.bin"
    _, h2 = transcode(lib, dst, dst2)
    assert h2 == h and dst2.read_bytes() == dst.read_bytes()


def test_snapshot_after_removals(lib, tmp_path):
    params = dict(max_edges=5, max_candidates=30, collection_size=256)
    ref, x = build(n=200, **params)
    gone = [17, 3, 150]
    ref.remove(gone)
    src, dst = tmp_path / "a.bin", tmp_path / "b.bin"
    src.write_bytes(oracle_snapshot(ref, x, params, removed=gone))
    info, h = transcode(lib, src, dst)
    assert info["count"] == 197 and info["length"] == 200 and h == ref.graph_hash()
    got = pbnet.decode(dst.read_bytes())
    assert got["removed"] == gone[::-1]  # ConcurrentStack enumerates from the top
    assert got["active"] == ref.active_ids().tolist()
    assert [i for i, nd in enumerate(got["nodes"]) if nd["removed"]] == sorted(gone)


def test_zero_defaults_and_negative_values(lib, tmp_path):
    # node 0 stays the entry point in a one-node index: EntryPointId == 0 is not on the wire
    x = uniform(1, 8, 3)
    ref = oracle.OracleIndex(8, random_seed=-5 & 0x7fffffff)
    ref.add(x)
    params = dict(random_seed=-5, collection_size=4)
    blob = oracle_snapshot(ref, x, params, capacity=4)
    assert 5 not in pbnet.decode(blob)["data_present"]
    src, dst = tmp_path / "a.bin", tmp_path / "b.bin"
    src.write_bytes(blob)
    info, _ = transcode(lib, src, dst)
    assert info["entry"] == 0 and info["random_seed"] == -5 and info["count"] == 1  # repaired, see snapshot_io.h
    got = pbnet.decode(dst.read_bytes())
    assert got["params"]["random_seed"] == -5 and 5 not in got["data_present"] and got["entry"] == -1
    # empty index
    src.write_bytes(pbnet.encode({}, [], [], [], [], -1, 65536, 0, 0))
    info, _ = transcode(lib, src, dst)
    assert info["length"] == 0 and info["entry"] == -1 and info["capacity"] == 65536
    got = pbnet.decode(dst.read_bytes())
    assert got["nodes"] == [] and got["entry"] == -1 and got["capacity"] == 65536


def test_malformed_snapshots_are_errors(lib, tmp_path):
    params = dict(max_edges=4, max_candidates=20, collection_size=64)
    ref, x = build(n=40, dim=4, **params)
    good = oracle_snapshot(ref, x, params)
    src = tmp_path / "a.bin"

    def fails(blob, text):
        src.write_bytes(blob)
        with pytest.raises(RuntimeError, match=text):
            transcode(lib, src, None)

    fails(good[:len(good) // 2], "invalid snapshot")
    fails(oracle_snapshot(ref, x, params, with_params=False), "Parameters cannot be null")  # HNSWIndex.cs:37-38
    fails(oracle_snapshot(ref, x, params, with_data=False), "Data cannot be null")          # HNSWIndex.cs:40-41
    fails(b"", "Parameters cannot be null")
    fails(oracle_snapshot(ref, x, params, capacity=10), "Capacity")
    n = ref.length

    def tweak(**kw):
        nodes = [dict(id=i, out=[(ref.edges(i, l).tolist(), len(ref.edges(i, l))) for l in range(ref.max_layer(i) + 1)]) for i in range(n)]
        args = dict(nodes=nodes, active=list(range(n)), items=x.tolist(), removed=[], entry=ref.entry_point, capacity=64, length=n, count=n)
        args.update(kw)
        return pbnet.encode(params, args["nodes"], args["active"], args["items"], args["removed"], args["entry"], args["capacity"],
                            args["length"], args["count"])

    transcode_ok = tweak()
    src.write_bytes(transcode_ok)
    assert transcode(lib, src, None)[1] == ref.graph_hash()
    bad_nodes = [dict(id=i, out=[([n + 5], 1)]) for i in range(n)]
    fails(tweak(nodes=bad_nodes), "edge id out of range")
    fails(tweak(nodes=[dict(id=i, out=[([1, 2], 3)]) for i in range(n)]), "Count beyond its Buffer")
    fails(tweak(nodes=[dict(id=i, out=[(list(range(12)), 12)]) for i in range(n)]), "longer than MaxEdges")
    fails(tweak(nodes=[dict(id=i, out=[]) for i in range(n)]), "without OutEdges")
    fails(tweak(items=x[:-1].tolist()), "differ in length")
    fails(tweak(items=[r[: 3 + (i % 2)] for i, r in enumerate(x.tolist())]), "different lengths")
    fails(tweak(active=[0, 0]), "ActiveNodes")
    fails(tweak(active=[n]), "ActiveNodes")
    fails(tweak(entry=n), "EntryPointId")
    fails(tweak(length=n - 1), "Length")
    fails(tweak(removed=[0]), "RemovedIndexes")  # an active id cannot be vacant
    # --- lists that would send a traversal out of bounds (post-pass over all lists) ---
    lv = [ref.max_layer(i) for i in range(n)]
    flat = next(i for i in range(n) if lv[i] == 0)                 # a node with layer 0 only
    tall = next(i for i in range(n) if lv[i] >= 1)                 # a node that has layer 1
    def nodes_with(edit):
        nodes = [dict(id=i, out=[(ref.edges(i, l).tolist(), len(ref.edges(i, l))) for l in range(lv[i] + 1)]) for i in range(n)]
        edit(nodes)
        return nodes
    def edge_to_missing_layer(nodes):                              # (tall, layer 1) -> a node without layer 1
        e, _ = nodes[tall]["out"][1]
        nodes[tall]["out"][1] = ([flat] + e[1:], max(1, len(e)))
    fails(tweak(nodes=nodes_with(edge_to_missing_layer)), "does not have that layer")
    def duplicate_edge(nodes):
        e, c = nodes[flat]["out"][0]
        nodes[flat]["out"][0] = ([e[0], e[0]] + e[2:], max(2, c))
    fails(tweak(nodes=nodes_with(duplicate_edge)), "duplicate id")
    # a live node pointing at a vacated slot; an entry point that was removed
    victim = next(i for i in range(n) if i != ref.entry_point and i != flat)
    def drop(nodes):
        nodes[victim]["removed"] = True
    live = [i for i in range(n) if i != victim]
    assert any(victim in ref.edges(i, 0).tolist() for i in live)
    fails(tweak(nodes=nodes_with(drop), active=live, removed=[victim], count=n - 1), "removed node")
    ep = ref.entry_point
    def drop_ep(nodes):
        nodes[ep]["removed"] = True
        for nd in nodes:
            nd["out"] = [([x for x in e[:c] if x != ep], len([x for x in e[:c] if x != ep])) for e, c in nd["out"]]
    fails(tweak(nodes=nodes_with(drop_ep), active=[i for i in range(n) if i != ep], removed=[ep], count=n - 1), "EntryPointId names a removed node")
    fails(tweak(nodes=nodes_with(drop)), "flagged IsRemoved")      # flagged but still listed active
    fails(tweak(active=live, count=n - 1), "neither in ActiveNodes nor flagged")

