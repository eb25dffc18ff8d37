"""Test-side restatement of the reference's snapshot wire format, written independently of
csrc/snapshot_io.h: protocol-buffers encoding of HNSWIndexSnapshot<float[],float>
(src/HNSWIndex/HNSWIndexSnapshot.cs:12-16, GraphDataSnapshot.cs:13-35, Node.cs:9-36,
HNSWParameters.cs:12-55, NestedListWrapper.cs:19-20) as protobuf-net 3.x writes it by default
(sub-messages length-delimited, int32 as sign-extended varint, zero-valued scalars omitted,
repeated scalars one tag per element -- `packed=True` produces the other legal encoding).
Pure Python: small cases only."""
import struct

VARINT, FIXED64, LEN, FIXED32 = 0, 1, 2, 5


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _tag(field, wt):
    return _varint((field << 3) | wt)


def _i32(field, v, omit_zero=True):
    if omit_zero and v == 0:
        return b""
    return _tag(field, VARINT) + _varint(v)


def _msg(field, body):
    return _tag(field, LEN) + _varint(len(body)) + body


def _rep_i32(field, vals, packed):
    if packed:
        return _msg(field, b"".join(_varint(int(v)) for v in vals)) if len(vals) else b""
    return b"".join(_tag(field, VARINT) + _varint(int(v)) for v in vals)


def _rep_f32(field, vals, packed):
    if packed:
        return _msg(field, struct.pack("<%df" % len(vals), *vals)) if len(vals) else b""
    return b"".join(_tag(field, FIXED32) + struct.pack("<f", float(v)) for v in vals)


DEFAULT_PARAMS = dict(max_edges=16, distribution_rate=0.36067376022224085, min_nn=5, max_candidates=100,
                      remove_max_candidates=100, collection_size=65536, random_seed=31337, allow_removals=True)


def encode(params, nodes, active, items, removed_stack_top_first, entry, capacity, length, count, packed=False,
           with_params=True, with_data=True):
    """nodes: list of dict(id, removed, out=[(buffer, count)], inn=[(buffer, count)])."""
    p = dict(DEFAULT_PARAMS)
    p.update(params)
    pb = (_i32(1, p["max_edges"]) + (_tag(2, FIXED64) + struct.pack("<d", p["distribution_rate"]) if p["distribution_rate"] != 0 else b"")
          + _i32(3, p["min_nn"]) + _i32(4, p["max_candidates"]) + _i32(5, p["remove_max_candidates"])
          + _i32(6, p["collection_size"]) + _i32(7, p["random_seed"]) + _i32(8, 1 if p["allow_removals"] else 0))

    def edge_list(buf, cnt):
        return _rep_i32(1, buf, packed) + _i32(2, cnt)

    db = b""
    for nd in nodes:
        body = _i32(1, nd["id"]) + _i32(2, 1 if nd.get("removed") else 0)
        body += b"".join(_msg(3, edge_list(b, c)) for b, c in nd["out"])
        body += b"".join(_msg(4, edge_list(b, c)) for b, c in nd.get("inn", []))
        db += _msg(1, body)
    db += _rep_i32(2, active, packed)
    for row in items:
        db += _msg(3, _rep_f32(1, row, packed))
    db += _rep_i32(4, removed_stack_top_first, packed)
    db += _i32(5, entry) + _i32(6, capacity) + _i32(7, length) + _i32(8, count)
    return (_msg(1, pb) if with_params else b"") + (_msg(2, db) if with_data else b"")


# ---- decoder ------------------------------------------------------------------------------
def _read_varint(b, i):
    v, s = 0, 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        s += 7
        if not c & 0x80:
            return v, i


def _s32(v):
    v &= (1 << 64) - 1
    if v >= 1 << 63:
        v -= 1 << 64
    return v


def _fields(b):
    i = 0
    while i < len(b):
        t, i = _read_varint(b, i)
        f, wt = t >> 3, t & 7
        if wt == VARINT:
            v, i = _read_varint(b, i)
        elif wt == FIXED64:
            v, i = b[i:i + 8], i + 8
        elif wt == LEN:
            n, i = _read_varint(b, i)
            v, i = b[i:i + n], i + n
        elif wt == FIXED32:
            v, i = b[i:i + 4], i + 4
        else:
            raise ValueError("wire type %d" % wt)
        yield f, wt, v


def _ints(wt, v):
    if wt == VARINT:
        return [_s32(v)]
    out, i = [], 0
    while i < len(v):
        x, i = _read_varint(v, i)
        out.append(_s32(x))
    return out


def _floats(wt, v):
    return list(struct.unpack("<%df" % (len(v) // 4), v))


def decode(b):
    """-> dict(params, present=set of scalar fields on the wire, nodes, active, items, removed, entry, ...)"""
    out = dict(params=dict(DEFAULT_PARAMS), params_present=set(), data_present=set(), nodes=[], active=[], items=[], removed=[],
               entry=-1, capacity=0, length=0, count=0, repeated_wire_types=set())
    names = {1: "max_edges", 3: "min_nn", 4: "max_candidates", 5: "remove_max_candidates", 6: "collection_size", 7: "random_seed"}
    for f, wt, v in _fields(b):
        if f == 1:
            for pf, pwt, pv in _fields(v):
                out["params_present"].add(pf)
                if pf == 2:
                    out["params"]["distribution_rate"] = struct.unpack("<d", pv)[0]
                elif pf == 8:
                    out["params"]["allow_removals"] = bool(pv)
                else:
                    out["params"][names[pf]] = _s32(pv)
        elif f == 2:
            for df, dwt, dv in _fields(v):
                if df == 1:
                    nd = dict(id=0, removed=False, out=[], inn=[])
                    for nf, nwt, nv in _fields(dv):
                        if nf == 1:
                            nd["id"] = _s32(nv)
                        elif nf == 2:
                            nd["removed"] = bool(nv)
                        else:
                            buf, cnt = [], 0
                            for ef, ewt, ev in _fields(nv):
                                if ef == 1:
                                    buf += _ints(ewt, ev)
                                    out["repeated_wire_types"].add(ewt)
                                else:
                                    cnt = _s32(ev)
                            nd["out" if nf == 3 else "inn"].append((buf, cnt))
                    out["nodes"].append(nd)
                elif df == 2:
                    out["active"] += _ints(dwt, dv)
                elif df == 3:
                    row = []
                    for f2, wt2, v2 in _fields(dv):
                        row += _floats(wt2, v2)
                        out["repeated_wire_types"].add(wt2)
                    out["items"].append(row)
                elif df == 4:
                    out["removed"] += _ints(dwt, dv)
                else:
                    out["data_present"].add(df)
                    out[{5: "entry", 6: "capacity", 7: "length", 8: "count"}[df]] = _s32(dv)
    return out
