// snapshot_io.h -- HNSWIndex.Serialize / Deserialize wire format.
//
// The reference writes `HNSWIndexSnapshot<float[],float>` with protobuf-net 3.2.52
// (src/HNSWIndex/HNSWIndex.cs:210-229, HNSWIndex.csproj:37).  protobuf-net is a third-party
// dependency that is not in /root/reference; what is restated here is the protocol-buffers wire
// format (public spec) applied to the reference's contracts:
//
//   HNSWIndexSnapshot  (HNSWIndexSnapshot.cs:12-16)   1: Parameters (message)   2: DataSnapshot (message)
//   HNSWParameters     (HNSWParameters.cs:12-55)      1: MaxEdges i32  2: DistributionRate f64  3: MinNN i32
//                                                     4: MaxCandidates i32  5: RemoveMaxCandidates i32
//                                                     6: CollectionSize i32  7: RandomSeed i32  8: AllowRemovals bool
//   GraphDataSnapshot  (GraphDataSnapshot.cs:13-35)   1: Nodes (repeated message)  2: ActiveNodes (repeated i32)
//                                                     3: Items (repeated NestedArrayWrapper)  4: RemovedIndexes (repeated i32)
//                                                     5: EntryPointId i32  6: Capacity i32  7: Length i32  8: Count i32
//   Node               (Node.cs:9-25)                 1: Id i32  2: IsRemoved bool  3: OutEdges (repeated EdgeList)
//                                                     4: InEdges (repeated EdgeList)
//   EdgeList           (Node.cs:33-36)                1: Buffer (repeated i32, the whole capacity)  2: Count i32
//   NestedArrayWrapper (NestedListWrapper.cs:19-20)   1: Values (repeated f32)
//
// protobuf-net conventions followed by the writer (PARITY UNPINNED: no serialized fixture exists
// in the reference and protobuf-net cannot run here; the reader is tolerant where they matter):
//   * sub-objects are length-delimited; int32 is a two's-complement varint (negative: 10 bytes);
//   * repeated scalars are NOT packed unless IsPacked is set (it is not): one tag per element.
//     The reader accepts packed and unpacked, as the protobuf spec requires;
//   * "implicit zero defaults": a scalar member equal to 0 / false is not written, and a reader
//     starts from the C# field initialisers (MaxEdges 16, ..., AllowRemovals true, EntryPointId -1).
//     Consequence in the reference: AllowRemovals=false reloads as true, and EntryPointId == 0
//     reloads as -1 (a reference index whose entry point is node 0 cannot be queried after
//     Deserialize).  This reader repairs the second case (absent EntryPointId with Count > 0 => 0);
//   * array elements are always written, including empty EdgeLists (tag + zero length).
// Loading follows GraphData's snapshot constructor (GraphData.cs:58-74): nodes and items are
// placed by POSITION in their arrays, the RNG restarts from RandomSeed, Count is not read.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "host_structs.h"

namespace hnsw {

struct SnapshotParams { // C# initialisers, HNSWParameters.cs:13-55
    int max_edges = 16;
    double distribution_rate = 0.36067376022224085;
    int min_nn = 5;
    int max_candidates = 100;
    int remove_max_candidates = 100;
    int collection_size = 65536;
    int random_seed = 31337;
    bool allow_removals = true;
};

namespace pbwire {

enum { VARINT = 0, FIXED64 = 1, LEN = 2, FIXED32 = 5 };

// ---- output: one code path both measures and writes ------------------------------------
struct Sink {
    FILE *f = nullptr; // nullptr: count only
    uint64_t n = 0;
    bool ok = true;
    char buf[1 << 16];
    size_t fill = 0;
    void flush()
    {
        if (f && fill && ok) ok = std::fwrite(buf, 1, fill, f) == fill;
        fill = 0;
    }
    inline void byte(uint8_t b)
    {
        ++n;
        if (!f) return;
        if (fill == sizeof buf) flush();
        buf[fill++] = (char)b;
    }
    void bytes(const void *p, size_t len)
    {
        n += len;
        if (!f) return;
        const char *c = static_cast<const char *>(p);
        while (len) {
            if (fill == sizeof buf) flush();
            size_t k = std::min(len, sizeof buf - fill);
            std::memcpy(buf + fill, c, k);
            fill += k; c += k; len -= k;
        }
    }
    inline void varint(uint64_t v)
    {
        while (v >= 0x80) { byte((uint8_t)(v | 0x80)); v >>= 7; }
        byte((uint8_t)v);
    }
    inline void tag(int field, int wt) { varint((uint64_t)((field << 3) | wt)); }
    inline void i32(int field, int v) { tag(field, VARINT); varint((uint64_t)(int64_t)v); } // sign-extended
    inline void i32_nz(int field, int v) { if (v != 0) i32(field, v); }                     // implicit zero default
    inline void f32(int field, float v) { tag(field, FIXED32); bytes(&v, 4); }
    inline void f64(int field, double v) { tag(field, FIXED64); bytes(&v, 8); }
};
inline size_t varint_size(uint64_t v) { size_t s = 1; while (v >= 0x80) { ++s; v >>= 7; } return s; }
inline size_t i32_size(int v) { return 1 + varint_size((uint64_t)(int64_t)v); } // fields < 16: one tag byte

// ---- input ------------------------------------------------------------------------------
struct Reader {
    const uint8_t *p, *end;
    bool ok = true;
    Reader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    bool done() const { return p >= end || !ok; }
    uint64_t varint()
    {
        uint64_t v = 0;
        for (int shift = 0; shift < 70; shift += 7) {
            if (p >= end) { ok = false; return 0; }
            uint8_t b = *p++;
            if (shift < 64) v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        ok = false;
        return 0;
    }
    bool tag(int &field, int &wt)
    {
        uint64_t t = varint();
        if (!ok) return false;
        field = (int)(t >> 3);
        wt = (int)(t & 7);
        return true;
    }
    Reader sub()
    {
        uint64_t len = varint();
        if (!ok || len > (uint64_t)(end - p)) { ok = false; return Reader(p, p); }
        Reader r(p, p + len);
        p += len;
        return r;
    }
    uint32_t fixed32()
    {
        if (end - p < 4) { ok = false; return 0; }
        uint32_t v; std::memcpy(&v, p, 4); p += 4;
        return v;
    }
    uint64_t fixed64()
    {
        if (end - p < 8) { ok = false; return 0; }
        uint64_t v; std::memcpy(&v, p, 8); p += 8;
        return v;
    }
    void skip(int wt)
    {
        switch (wt) {
        case VARINT: varint(); break;
        case FIXED64: fixed64(); break;
        case LEN: sub(); break;
        case FIXED32: fixed32(); break;
        default: ok = false; // groups are not used by these contracts
        }
    }
    // one occurrence of a repeated int32 member: a single varint or a packed run
    template <class F> void rep_i32(int wt, F &&push)
    {
        if (wt == VARINT) { int v = (int)(int64_t)varint(); if (ok) push(v); }
        else if (wt == LEN) { Reader r = sub(); while (ok && !r.done()) { int v = (int)(int64_t)r.varint(); if (r.ok) push(v); } ok = ok && r.ok; }
        else ok = false;
    }
    template <class F> void rep_f32(int wt, F &&push)
    {
        if (wt == FIXED32) { uint32_t u = fixed32(); float v; std::memcpy(&v, &u, 4); if (ok) push(v); }
        else if (wt == LEN) { Reader r = sub(); while (ok && !r.done()) { uint32_t u = r.fixed32(); float v; std::memcpy(&v, &u, 4); if (r.ok) push(v); } ok = ok && r.ok; }
        else ok = false;
    }
};

} // namespace pbwire

// ---- writer ---------------------------------------------------------------------------------
namespace snapshot_detail {
using pbwire::Sink;

// EdgeList body: Buffer padded with zeros to the capacity NewNode gives it (GraphData.cs:232),
// so that a reference reader never sees a null Buffer (EdgeList.Add dereferences it, Node.cs:69).
inline void edge_list_body(Sink &s, const int *ids, int cnt, int cap)
{
    for (int j = 0; j < cnt; ++j) s.i32(1, ids[j]);
    for (int j = cnt; j < cap; ++j) s.i32(1, 0);
    s.i32_nz(2, cnt);
}
inline uint64_t edge_list_size(const int *ids, int cnt, int cap)
{
    uint64_t n = 0;
    for (int j = 0; j < cnt; ++j) n += pbwire::i32_size(ids[j]);
    n += 2ull * (uint64_t)std::max(0, cap - cnt);
    if (cnt) n += pbwire::i32_size(cnt);
    return n;
}

struct InEdges { // in-edge lists by transposition (content only; Node.InEdges order is not observable)
    std::vector<int64_t> off; // per (node, layer) running offsets
    std::vector<int> ids;
    std::vector<int64_t> base; // first (node,layer) key per node
};
inline void transpose(const Graph &g, InEdges &t)
{
    const int n = g.length;
    t.base.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) t.base[(size_t)i + 1] = t.base[(size_t)i] + g.level[(size_t)i] + 1;
    const size_t keys = (size_t)t.base[(size_t)n];
    t.off.assign(keys + 1, 0);
    auto live = [&](int i) { return !g.removed[(size_t)i]; };
    for (int i = 0; i < n; ++i) {
        if (!live(i)) continue;
        for (int l = 0; l <= g.level[(size_t)i]; ++l) {
            const int *e = g.list(i, l);
            for (int j = 1; j <= e[0]; ++j)
                if (e[j] >= 0 && e[j] < n && g.level[(size_t)e[j]] >= l) ++t.off[(size_t)(t.base[(size_t)e[j]] + l) + 1];
        }
    }
    for (size_t k = 0; k < keys; ++k) t.off[k + 1] += t.off[k];
    t.ids.assign((size_t)t.off[keys], 0);
    std::vector<int64_t> cur(t.off.begin(), t.off.end() - 1);
    for (int i = 0; i < n; ++i) {
        if (!live(i)) continue;
        for (int l = 0; l <= g.level[(size_t)i]; ++l) {
            const int *e = g.list(i, l);
            for (int j = 1; j <= e[0]; ++j)
                if (e[j] >= 0 && e[j] < n && g.level[(size_t)e[j]] >= l) t.ids[(size_t)cur[(size_t)(t.base[(size_t)e[j]] + l)]++] = i;
        }
    }
}

inline void node_body(Sink &s, const Graph &g, int i, const InEdges *in)
{
    s.i32_nz(1, i);
    if (g.removed[(size_t)i]) { s.tag(2, pbwire::VARINT); s.byte(1); }
    for (int l = 0; l <= g.level[(size_t)i]; ++l) {
        const int *e = g.list(i, l);
        const int cap = std::max(e[0], g.max_edges_at(l) + 1);
        s.tag(3, pbwire::LEN);
        s.varint(edge_list_size(e + 1, e[0], cap));
        edge_list_body(s, e + 1, e[0], cap);
    }
    if (in) {
        for (int l = 0; l <= g.level[(size_t)i]; ++l) {
            const size_t k = (size_t)(in->base[(size_t)i] + l);
            const int *ids = in->ids.data() + in->off[k];
            const int cnt = (int)(in->off[k + 1] - in->off[k]);
            const int cap = std::max(cnt, g.max_edges_at(l) + 1);
            s.tag(4, pbwire::LEN);
            s.varint(edge_list_size(ids, cnt, cap));
            edge_list_body(s, ids, cnt, cap);
        }
    }
}

inline void params_body(Sink &s, const SnapshotParams &p)
{
    s.i32_nz(1, p.max_edges);
    if (p.distribution_rate != 0.0) s.f64(2, p.distribution_rate);
    s.i32_nz(3, p.min_nn);
    s.i32_nz(4, p.max_candidates);
    s.i32_nz(5, p.remove_max_candidates);
    s.i32_nz(6, p.collection_size);
    s.i32_nz(7, p.random_seed);
    if (p.allow_removals) { s.tag(8, pbwire::VARINT); s.byte(1); }
}

// rows: `length` x dim floats, row i = Items[i]
inline void data_body(Sink &s, const Graph &g, const float *rows, int dim, long long capacity, const InEdges *in)
{
    Sink cnt;
    for (int i = 0; i < g.length; ++i) { // 1: Nodes
        cnt.n = 0;
        node_body(cnt, g, i, in);
        s.tag(1, pbwire::LEN);
        s.varint(cnt.n);
        node_body(s, g, i, in);
    }
    for (int i = 0; i < g.count; ++i) s.i32(2, g.dense[(size_t)i]); // 2: ActiveNodes = dense[..count]
    const uint64_t item_len = 5ull * (uint64_t)dim;                 // 3: Items, each {1: Values}
    for (int i = 0; i < g.length; ++i) {
        s.tag(3, pbwire::LEN);
        s.varint(item_len);
        const float *r = rows + (size_t)i * dim;
        for (int j = 0; j < dim; ++j) s.f32(1, r[j]);
    }
    for (size_t i = g.removed_stack.size(); i-- > 0;) s.i32(4, g.removed_stack[i]); // 4: ConcurrentStack enumerates top first
    s.i32_nz(5, g.entry);
    s.i32_nz(6, (int)capacity);
    s.i32_nz(7, g.length);
    s.i32_nz(8, g.count);
}

} // namespace snapshot_detail

inline bool write_snapshot(const char *path, const SnapshotParams &p, const Graph &g, const float *rows, int dim, long long capacity,
                           std::string &err)
{
    using namespace snapshot_detail;
    if (capacity > 0x7fffffffLL) { err = "System.OverflowException: Capacity"; return false; }
    InEdges in;
    const InEdges *inp = nullptr;
    if (p.allow_removals) { transpose(g, in); inp = &in; } // NewNode allocates InEdges only then (GraphData.cs:227)
    Sink measure;
    params_body(measure, p);
    const uint64_t plen = measure.n;
    measure.n = 0;
    data_body(measure, g, rows, dim, capacity, inp);
    const uint64_t dlen = measure.n;
    FILE *f = std::fopen(path, "wb");
    if (!f) { err = std::string("System.IO.IOException: cannot create ") + path; return false; }
    Sink *s = new Sink();
    s->f = f;
    s->tag(1, pbwire::LEN); s->varint(plen); params_body(*s, p);
    s->tag(2, pbwire::LEN); s->varint(dlen); data_body(*s, g, rows, dim, capacity, inp);
    s->flush();
    bool ok = s->ok;
    delete s;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) err = std::string("System.IO.IOException: write failed: ") + path;
    return ok;
}

// ---- reader -----------------------------------------------------------------------------------
// Fills `g` (configured from the decoded MaxEdges), `rows` (length x dim) and the scalars.
inline bool read_snapshot(const uint8_t *buf, size_t len, SnapshotParams &p, Graph &g, std::vector<float> &rows, int &dim,
                          long long &capacity, std::string &err)
{
    using pbwire::Reader;
    auto bad = [&](const char *what) { err = std::string("ProtoBuf.ProtoException: invalid snapshot: ") + what; return false; };
    Reader root(buf, buf + len);
    const uint8_t *pb = nullptr, *pe = nullptr, *db = nullptr, *de = nullptr;
    while (!root.done()) {
        int f, wt;
        if (!root.tag(f, wt)) break;
        if ((f == 1 || f == 2) && wt == pbwire::LEN) {
            Reader r = root.sub();
            if (f == 1) { pb = r.p; pe = r.end; } else { db = r.p; de = r.end; }
        } else root.skip(wt);
    }
    if (!root.ok) return bad("truncated stream");
    // HNSWIndex.cs:37-41
    if (!pb) { err = "System.ArgumentNullException: Parameters cannot be null during deserialization. (Parameter 'Parameters')"; return false; }
    if (!db) { err = "System.ArgumentNullException: Data cannot be null during deserialization. (Parameter 'DataSnapshot')"; return false; }

    p = SnapshotParams();
    {
        Reader r(pb, pe);
        while (!r.done()) {
            int f, wt;
            if (!r.tag(f, wt)) break;
            if (wt == pbwire::VARINT && f >= 1 && f <= 8 && f != 2) {
                const int64_t v = (int64_t)r.varint();
                switch (f) {
                case 1: p.max_edges = (int)v; break;
                case 3: p.min_nn = (int)v; break;
                case 4: p.max_candidates = (int)v; break;
                case 5: p.remove_max_candidates = (int)v; break;
                case 6: p.collection_size = (int)v; break;
                case 7: p.random_seed = (int)v; break;
                case 8: p.allow_removals = v != 0; break;
                }
            } else if (wt == pbwire::FIXED64 && f == 2) {
                const uint64_t u = r.fixed64();
                std::memcpy(&p.distribution_rate, &u, 8);
            } else r.skip(wt);
        }
        if (!r.ok) return bad("parameters");
    }
    if (p.max_edges < 1 || p.max_edges > (1 << 20)) return bad("MaxEdges");

    // pass 1 over the data message: scalars, element counts
    int entry = -1, cap32 = 0, length = 0;
    bool has_entry = false;
    size_t n_nodes = 0, n_items = 0;
    {
        Reader r(db, de);
        while (!r.done()) {
            int f, wt;
            if (!r.tag(f, wt)) break;
            if (f == 1 && wt == pbwire::LEN) { r.sub(); ++n_nodes; }
            else if (f == 3 && wt == pbwire::LEN) { r.sub(); ++n_items; }
            else if (f >= 5 && f <= 8 && wt == pbwire::VARINT) {
                const int v = (int)(int64_t)r.varint();
                if (f == 5) { entry = v; has_entry = true; } else if (f == 6) cap32 = v; else if (f == 7) length = v;
            } else r.skip(wt);
        }
        if (!r.ok) return bad("data");
    }
    if (n_nodes > 0x7fffffffu || n_items != n_nodes) return bad("Nodes and Items differ in length");
    const int n = (int)n_nodes;
    if (length != n) return bad("Length does not match the number of nodes");
    if (cap32 < n) return bad("Capacity below Length");
    capacity = cap32;

    g = Graph();
    g.configure(p.max_edges);
    g.length = n;
    g.level.assign((size_t)n, 0);
    g.upper.assign((size_t)n, -1);
    g.adj0.assign((size_t)n * g.stride0, 0);
    g.removed.assign((size_t)n, 0);
    g.dense.assign((size_t)n, 0);
    g.sparse.assign((size_t)n, 0);
    dim = -1;
    rows.clear();

    // pass 2: nodes, items, id lists (arrays are filled by position: GraphDataSnapshot.cs:40-55)
    int node_pos = 0, item_pos = 0;
    std::vector<int> active, removed_wire, lst;
    Reader r(db, de);
    while (!r.done()) {
        int f, wt;
        if (!r.tag(f, wt)) break;
        if (f == 1 && wt == pbwire::LEN) {
            Reader nr = r.sub();
            const int i = node_pos++;
            int lvl = -1;
            while (!nr.done()) {
                int nf, nwt;
                if (!nr.tag(nf, nwt)) break;
                if (nf == 2 && nwt == pbwire::VARINT) g.removed[(size_t)i] = nr.varint() != 0;
                else if (nf == 3 && nwt == pbwire::LEN) {
                    Reader er = nr.sub();
                    ++lvl;
                    lst.clear();
                    int cnt = 0;
                    while (!er.done()) {
                        int ef, ewt;
                        if (!er.tag(ef, ewt)) break;
                        if (ef == 1) er.rep_i32(ewt, [&](int v) { lst.push_back(v); });
                        else if (ef == 2 && ewt == pbwire::VARINT) cnt = (int)(int64_t)er.varint();
                        else er.skip(ewt);
                    }
                    if (!er.ok) return bad("EdgeList");
                    if (cnt < 0 || cnt > (int)lst.size()) return bad("EdgeList.Count beyond its Buffer");
                    if (cnt > g.max_edges_at(lvl) + 1) return bad("EdgeList longer than MaxEdges + 1");
                    if (lvl == 1) { g.upper[(size_t)i] = (int64_t)g.pool.size(); }
                    if (lvl >= 1) g.pool.resize(g.pool.size() + (size_t)g.strideU, 0);
                    int *dst = lvl == 0 ? g.adj0.data() + (size_t)i * g.stride0 : g.pool.data() + g.upper[(size_t)i] + (size_t)(lvl - 1) * g.strideU;
                    dst[0] = cnt;
                    for (int j = 0; j < cnt; ++j) {
                        if (lst[(size_t)j] < 0 || lst[(size_t)j] >= n) return bad("edge id out of range");
                        dst[1 + j] = lst[(size_t)j];
                    }
                } else nr.skip(nwt); // Id (position decides), InEdges (rebuilt on demand)
            }
            if (!nr.ok) return bad("Node");
            if (lvl < 0) return bad("Node without OutEdges");
            if (lvl > 200) return bad("Node level");
            g.level[(size_t)i] = lvl;
        } else if (f == 2) {
            r.rep_i32(wt, [&](int v) { active.push_back(v); });
        } else if (f == 3 && wt == pbwire::LEN) {
            Reader ir = r.sub();
            const int i = item_pos++;
            size_t got = 0;
            const size_t base = dim < 0 ? 0 : (size_t)i * (size_t)dim;
            if (dim >= 0) rows.resize(base + (size_t)dim);
            while (!ir.done()) {
                int f2, wt2;
                if (!ir.tag(f2, wt2)) break;
                if (f2 == 1) ir.rep_f32(wt2, [&](float v) {
                    if (dim < 0) rows.push_back(v);
                    else if (got < (size_t)dim) rows[base + got] = v;
                    ++got;
                });
                else ir.skip(wt2);
            }
            if (!ir.ok) return bad("item");
            if (dim < 0) {
                if (got == 0 || got > (1u << 24)) return bad("empty item");
                dim = (int)got;
                rows.reserve((size_t)n * (size_t)dim);
            } else if (got != (size_t)dim) return bad("items of different lengths (this backend stores one dense matrix)");
        } else if (f == 4) {
            r.rep_i32(wt, [&](int v) { removed_wire.push_back(v); });
        } else r.skip(wt);
    }
    if (!r.ok) return bad("data");
    if (n == 0) dim = 0;

    // ActiveSet(int[] activeIds) (ActiveSet.cs:40-51).  The reference sizes `sparse` by the number
    // of active ids and throws IndexOutOfRange when an active id is beyond it (any snapshot taken
    // after a removal); this loader sizes by Length instead and accepts those.
    if ((int)active.size() > n) return bad("more active ids than nodes");
    std::vector<char> seen((size_t)n, 0);
    g.count = (int)active.size();
    for (int i = 0; i < g.count; ++i) {
        const int id = active[(size_t)i];
        if (id < 0 || id >= n || seen[(size_t)id]) return bad("ActiveNodes");
        seen[(size_t)id] = 1;
        g.dense[(size_t)i] = id;
        g.sparse[(size_t)id] = i;
    }
    for (size_t i = removed_wire.size(); i-- > 0;) { // wire order is top first; back() is the top here
        const int id = removed_wire[i];
        if (id < 0 || id >= n || seen[(size_t)id]) return bad("RemovedIndexes");
        g.removed_stack.push_back(id);
    }
    if (!has_entry && g.count > 0) entry = 0; // see header: zero is never on the wire
    if (g.count > 0 && (entry < 0 || entry >= n)) return bad("EntryPointId");
    if (entry >= n) return bad("EntryPointId");
    // Post-pass over every list, now that all levels and the active set are known (nodes are parsed
    // in order, so an edge could not be checked against its target's level while reading it).  The
    // traversals -- host and device -- index `pool + upper[target]` for an edge met on an upper layer
    // and trust lists to hold live, distinct ids: a snapshot that breaks that is refused here rather
    // than read out of bounds later.
    for (int i = 0; i < n; ++i) {
        if (seen[(size_t)i] && g.removed[(size_t)i]) return bad("an active node is flagged IsRemoved");
        if (!seen[(size_t)i] && !g.removed[(size_t)i]) return bad("a node is neither in ActiveNodes nor flagged IsRemoved");
    }
    if (g.count > 0 && !seen[(size_t)entry]) return bad("EntryPointId names a removed node");
    {
        std::vector<int> stamp((size_t)n, -1);
        int list_no = 0;
        for (int a = 0; a < g.count; ++a) {
            const int i = g.dense[(size_t)a];
            for (int layer = 0; layer <= g.level[(size_t)i]; ++layer, ++list_no) {
                const int *l = g.list(i, layer);
                for (int e = 1; e <= l[0]; ++e) {
                    const int t = l[e];
                    if (g.level[(size_t)t] < layer) return bad("edge to a node that does not have that layer");
                    if (!seen[(size_t)t]) return bad("edge from a live node to a removed node");
                    if (stamp[(size_t)t] == list_no) return bad("duplicate id in an EdgeList");
                    stamp[(size_t)t] = list_no;
                }
            }
        }
    }
    g.entry = entry;
    return true;
}

} // namespace hnsw
