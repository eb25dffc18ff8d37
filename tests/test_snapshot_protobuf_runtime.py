"""CPU tier: the snapshot files the PRODUCT writes (csrc/snapshot_io.h), parsed by the stock `google.protobuf`
runtime -- an independent decoder, not written by this repo.  The message descriptors are built programmatically
(descriptor_pb2 + message_factory, no protoc) from the reference's protobuf-net contracts:
  HNSWIndexSnapshot   /root/reference/src/HNSWIndex/HNSWIndexSnapshot.cs:12-16    (1 Parameters, 2 DataSnapshot)
  HNSWParameters      /root/reference/src/HNSWIndex/HNSWParameters.cs:12-55        (1..8)
  GraphDataSnapshot   /root/reference/src/HNSWIndex/GraphDataSnapshot.cs:13-35     (1 Nodes .. 8 Count)
  Node / EdgeList     /root/reference/src/HNSWIndex/Node.cs:9-36
  NestedArrayWrapper  /root/reference/src/HNSWIndex/NestedListWrapper.cs:14-21      (1 Values = float[])
protobuf-net maps int -> int32 (varint, sign-extended), bool -> bool, double -> double, float[] -> repeated float,
T[] of a contract type -> repeated message.  Still "parity unpinned" against protobuf-net itself (it cannot run
here), but the same-author codec tests/pbnet.py is no longer the only witness of what the product writes."""
import ctypes as ct

import numpy as np
import pytest

import oracle
from common import uniform
from test_snapshot_codec import build, oracle_snapshot, transcode

pb = pytest.importorskip("google.protobuf")
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory  # noqa: E402

F = descriptor_pb2.FieldDescriptorProto


def _messages():
    fd = descriptor_pb2.FileDescriptorProto(name="hnsw_snapshot.proto", package="hnswref", syntax="proto2")

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for num, fname, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = ".hnswref." + tname
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("EdgeList", [(1, "Buffer", F.TYPE_INT32, REP, None), (2, "Count", F.TYPE_INT32, OPT, None)])
    msg("Node", [(1, "Id", F.TYPE_INT32, OPT, None), (2, "IsRemoved", F.TYPE_BOOL, OPT, None),
                 (3, "OutEdges", F.TYPE_MESSAGE, REP, "EdgeList"), (4, "InEdges", F.TYPE_MESSAGE, REP, "EdgeList")])
    msg("NestedArrayWrapper", [(1, "Values", F.TYPE_FLOAT, REP, None)])
    msg("HNSWParameters", [(1, "MaxEdges", F.TYPE_INT32, OPT, None), (2, "DistributionRate", F.TYPE_DOUBLE, OPT, None),
                           (3, "MinNN", F.TYPE_INT32, OPT, None), (4, "MaxCandidates", F.TYPE_INT32, OPT, None),
                           (5, "RemoveMaxCandidates", F.TYPE_INT32, OPT, None), (6, "CollectionSize", F.TYPE_INT32, OPT, None),
                           (7, "RandomSeed", F.TYPE_INT32, OPT, None), (8, "AllowRemovals", F.TYPE_BOOL, OPT, None)])
    msg("GraphDataSnapshot", [(1, "Nodes", F.TYPE_MESSAGE, REP, "Node"), (2, "ActiveNodes", F.TYPE_INT32, REP, None),
                              (3, "Items", F.TYPE_MESSAGE, REP, "NestedArrayWrapper"), (4, "RemovedIndexes", F.TYPE_INT32, REP, None),
                              (5, "EntryPointId", F.TYPE_INT32, OPT, None), (6, "Capacity", F.TYPE_INT32, OPT, None),
                              (7, "Length", F.TYPE_INT32, OPT, None), (8, "Count", F.TYPE_INT32, OPT, None)])
    msg("HNSWIndexSnapshot", [(1, "Parameters", F.TYPE_MESSAGE, OPT, "HNSWParameters"), (2, "DataSnapshot", F.TYPE_MESSAGE, OPT, "GraphDataSnapshot")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("hnswref.HNSWIndexSnapshot"))


@pytest.fixture(scope="module")
def Snapshot():
    return _messages()


@pytest.fixture(scope="module")
def lib():
    import hnswindex
    L = hnswindex.net_amd.lib
    L.hnswhost_test_snapshot_transcode.argtypes = [ct.c_char_p, ct.c_char_p, ct.POINTER(ct.c_int), ct.POINTER(ct.c_uint64)]
    L.hnswhost_test_snapshot_transcode.restype = ct.c_int
    return L


@pytest.mark.parametrize("allow_removals", [True, False])
def test_stock_protobuf_runtime_reads_what_the_product_writes(lib, Snapshot, tmp_path, allow_removals):
    params = dict(max_edges=6, max_candidates=40, min_nn=9, remove_max_candidates=33, collection_size=512, random_seed=4242,
                  allow_removals=allow_removals)
    ref, x = build(**params)
    removed = ()
    if allow_removals:
        removed = (5, 17, 123)
        ref.remove(np.array(removed, dtype=np.int32))
    src, dst = tmp_path / "in.bin", tmp_path / "product.bin"
    src.write_bytes(oracle_snapshot(ref, x, params, removed=removed))
    transcode(lib, src, dst)                 # decoded and re-encoded by csrc/snapshot_io.h
    snap = Snapshot()
    snap.ParseFromString(dst.read_bytes())   # the stock runtime rejects malformed wire data
    from google.protobuf import unknown_fields
    assert len(unknown_fields.UnknownFieldSet(snap)) == 0 and len(unknown_fields.UnknownFieldSet(snap.DataSnapshot)) == 0  # every field is one of the contracts'
    p, d = snap.Parameters, snap.DataSnapshot
    assert (p.MaxEdges, p.MinNN, p.MaxCandidates, p.RemoveMaxCandidates, p.CollectionSize, p.RandomSeed) == (6, 9, 40, 33, 512, 4242)
    # protobuf-net omits a false bool and the field's initialiser is `true` (HNSWParameters.cs:54-55): a snapshot of
    # an AllowRemovals=false index reloads as true in the reference, and so it does here
    assert p.AllowRemovals is True and abs(p.DistributionRate - 0.36067376022224085) < 1e-15
    n = ref.length
    assert (d.Length, d.Count, d.Capacity) == (n, n - len(removed), 512)
    assert d.EntryPointId == ref.entry_point
    assert list(d.ActiveNodes) == ref.active_ids().tolist()
    assert list(d.RemovedIndexes) == list(removed)[::-1]          # ConcurrentStack enumerates top first
    assert len(d.Nodes) == n and len(d.Items) == n
    for i in (0, 1, 77, n - 1):
        assert np.array(d.Items[i].Values, dtype=np.float32).tobytes() == x[i].tobytes()
    for i, node in enumerate(d.Nodes):
        assert node.Id == i and node.IsRemoved == (i in removed)
        assert len(node.OutEdges) == ref.max_layer(i) + 1
        if i in removed:
            continue
        for l, el in enumerate(node.OutEdges):
            want = ref.edges(i, l).tolist()
            assert el.Count == len(want) and list(el.Buffer)[:el.Count] == want
            assert len(el.Buffer) >= (2 * 6 if l == 0 else 6) + 1     # never a null / short Buffer for a reference reader
        assert len(node.InEdges) == ref.max_layer(i) + 1              # written by transposing the out-lists
        if allow_removals:
            for l, el in enumerate(node.InEdges):
                assert sorted(list(el.Buffer)[:el.Count]) == sorted(ref.edges(i, l, incoming=True).tolist())


def test_stock_runtime_round_trip_into_the_product(lib, Snapshot, tmp_path):
    # the other direction: a file SERIALISED by the stock runtime (packed repeated scalars, its own field order)
    # is accepted by the product's reader and re-encodes to the same graph
    params = dict(max_edges=5, max_candidates=30, collection_size=300, random_seed=99, allow_removals=False)
    ref, x = build(n=200, dim=7, **params)
    a, b, c = tmp_path / "a.bin", tmp_path / "b.bin", tmp_path / "c.bin"
    a.write_bytes(oracle_snapshot(ref, x, params))
    _, h0 = transcode(lib, a, b)
    snap = Snapshot()
    snap.ParseFromString(b.read_bytes())
    c.write_bytes(snap.SerializeToString())
    info, h1 = transcode(lib, c, None)
    assert h1 == h0 == ref.graph_hash() and info["length"] == 200 and info["dim"] == 7
