// tools/host_sanitize/harness.cpp -- the library's pure-host code under AddressSanitizer + UBSan (CPU only; GPU ASan is not
// available on this pool): csrc/snapshot_io.h parses untrusted files, csrc/host_structs.h and csrc/range_replay.h hold the
// restated BCL pieces.  Built and run by tests/test_host_sanitizers.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I hnswindex.net_amd/csrc harness.cpp
//   harness parse <file>...            decode each file (errors are fine: only a sanitizer report is a failure)
//   harness fuzz <seed file> <iterations> <rng seed>   structure-aware mutations of a valid snapshot
//   harness structs <rng seed>         heaps, the restated Span.Sort (NaN / -0 / ties), System.Random, the range replay
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_structs.h"
#include "range_replay.h"
#include "snapshot_io.h"

using namespace hnsw;

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> b;
    FILE *f = std::fopen(path, "rb");
    if (!f) return b;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    b.resize((size_t)(n > 0 ? n : 0));
    if (n > 0 && std::fread(b.data(), 1, (size_t)n, f) != (size_t)n) b.clear();
    std::fclose(f);
    return b;
}

// decode; whatever decodes is walked the way a traversal would walk it, re-encoded and decoded again
static int exercise(const std::vector<uint8_t> &buf, bool verbose)
{
    SnapshotParams sp;
    Graph g;
    std::vector<float> rows;
    int dim = 0;
    long long cap = 0;
    std::string err;
    if (!read_snapshot(buf.data(), buf.size(), sp, g, rows, dim, cap, err)) {
        if (verbose) std::printf("rejected: %s\n", err.c_str());
        return 0;
    }
    // every list the reader let through must be walkable: ids inside the graph, layers that exist
    uint64_t walked = 0;
    for (int i = 0; i < g.length; ++i) {
        if (g.removed[(size_t)i]) continue;
        for (int l = 0; l <= g.level[(size_t)i]; ++l) {
            const int *e = g.list(i, l);
            for (int t = 1; t <= e[0]; ++t) walked += (uint64_t)g.level[(size_t)e[t]] + (uint64_t)g.list(e[t], l)[0];
        }
    }
    const uint64_t h1 = graph_hash_of(g);
    const char *tmp = "/tmp/hnsw_sanitize_roundtrip.bin";
    if (!write_snapshot(tmp, sp, g, rows.data(), dim, cap, err)) { if (verbose) std::printf("write refused: %s\n", err.c_str()); return 0; }
    const std::vector<uint8_t> again = slurp(tmp);
    SnapshotParams sp2;
    Graph g2;
    std::vector<float> rows2;
    int dim2 = 0;
    long long cap2 = 0;
    if (!read_snapshot(again.data(), again.size(), sp2, g2, rows2, dim2, cap2, err)) { std::printf("ROUND TRIP LOST: %s\n", err.c_str()); return 2; }
    const bool same_rows = rows2.size() == rows.size() && (rows.empty() || !std::memcmp(rows2.data(), rows.data(), rows.size() * sizeof(float))); // (bits: a mutated item may be NaN)
    if (graph_hash_of(g2) != h1 || !same_rows || dim2 != dim) {
        std::printf("ROUND TRIP CHANGED THE GRAPH (hash %d, rows %d, dim %d / %d)\n", (int)(graph_hash_of(g2) == h1), (int)same_rows, dim, dim2);
        return 2;
    }
    // the range replay on the decoded layer 0 (everything within an infinite range of a made-up distance field)
    if (g.entry >= 0 && g.count > 0) {
        struct Hit { int id; float dist; };
        std::vector<Hit> found;
        for (int i = 0; i < g.length && found.size() < 200; ++i)
            if (!g.removed[(size_t)i]) found.push_back(Hit{i, (float)((i * 2654435761u) % 7u)}); // plenty of equal distances
        std::vector<NodeDist> out;
        replay_range_heaps([&](int id) { return g.list(id, 0); }, g.max_edges_at(0), g.entry, 1e9f, found.data(), (int)found.size(), out);
        walked += out.size();
    }
    if (verbose) std::printf("ok: length %d dim %d count %d hash %llu walked %llu\n", g.length, dim, g.count, (unsigned long long)h1, (unsigned long long)walked);
    return 0;
}

struct Rng {
    uint64_t s;
    uint32_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }
    uint32_t below(uint32_t n) { return n ? next() % n : 0; }
};

// where the length-delimited fields start: a mutation that hits a tag / length byte goes deeper than a random flip
static void collect_len_fields(const uint8_t *p, const uint8_t *end, int depth, std::vector<size_t> &at, const uint8_t *base)
{
    pbwire::Reader r(p, end);
    while (!r.done() && r.ok) {
        const uint8_t *here = r.p;
        int f, wt;
        if (!r.tag(f, wt)) break;
        if (wt == pbwire::LEN) {
            at.push_back((size_t)(here - base));
            pbwire::Reader sub = r.sub();
            if (r.ok && depth < 4 && sub.end - sub.p < (1 << 20)) collect_len_fields(sub.p, sub.end, depth + 1, at, base);
        } else r.skip(wt);
    }
}

static int fuzz(const char *seed_path, int iters, uint64_t seed)
{
    const std::vector<uint8_t> good = slurp(seed_path);
    if (good.empty()) { std::printf("cannot read %s\n", seed_path); return 3; }
    if (exercise(good, true) != 0) return 2;
    std::vector<size_t> fields;
    collect_len_fields(good.data(), good.data() + good.size(), 0, fields, good.data());
    Rng rng{seed * 0x9E3779B97F4A7C15ull + 1};
    int decoded = 0;
    for (int it = 0; it < iters; ++it) {
        std::vector<uint8_t> b = good;
        const int nmut = 1 + (int)rng.below(4);
        for (int m = 0; m < nmut && !b.empty(); ++m) {
            const size_t pos = !fields.empty() && rng.below(3) ? std::min(b.size() - 1, fields[rng.below((uint32_t)fields.size())] + rng.below(3)) : rng.below((uint32_t)b.size());
            switch (rng.below(8)) {
            case 0: b[pos] ^= (uint8_t)(1u << rng.below(8)); break;                                   // bit flip
            case 1: b[pos] = (uint8_t)rng.next(); break;                                              // byte
            case 2: b.resize(pos); break;                                                             // truncate
            case 3: for (size_t i = pos; i < std::min(b.size(), pos + 10); ++i) b[i] = 0xff; break;   // a varint that never ends
            case 4: b.insert(b.begin() + (long)pos, (size_t)(1 + rng.below(16)), (uint8_t)rng.next()); break; // insert
            case 5: b.erase(b.begin() + (long)pos, b.begin() + (long)std::min(b.size(), pos + 1 + rng.below(32))); break; // delete
            case 6: { const size_t len = std::min(b.size() - pos, (size_t)(1 + rng.below(64))); std::vector<uint8_t> cut(b.begin() + (long)pos, b.begin() + (long)(pos + len)); b.insert(b.begin() + (long)rng.below((uint32_t)b.size()), cut.begin(), cut.end()); break; } // splice a copy elsewhere
            default: b[pos] = (uint8_t)(rng.below(2) ? 0x7f : 0x80); break;                           // length / sign boundary
            }
        }
        const int rc = exercise(b, false);
        if (rc != 0) { std::printf("iteration %d failed\n", it); return rc; }
        SnapshotParams sp; Graph g; std::vector<float> rows; int dim = 0; long long cap = 0; std::string err;
        decoded += read_snapshot(b.data(), b.size(), sp, g, rows, dim, cap, err) ? 1 : 0;
    }
    std::printf("fuzz: %d mutated snapshots, %d still decoded, no fault\n", iters, decoded);
    return 0;
}

static int structs(uint64_t seed)
{
    Rng rng{seed * 0x9E3779B97F4A7C15ull + 7};
    // heaps: random push / pop scripts, tie-heavy keys
    for (int rep = 0; rep < 200; ++rep) {
        BinaryHeap<FartherFirst> hf; BinaryHeap<CloserFirst> hc;
        hf.reset(1 + (int)rng.below(8)); hc.reset(1 + (int)rng.below(8));
        for (int i = 0; i < 400; ++i) {
            const NodeDist v{(int)rng.below(1000), (float)rng.below(16) * 0.25f};
            if (rng.below(3)) { hf.push(v); hc.push(v); }
            else { if (hf.count > 0) hf.pop(); if (hc.count > 0) hc.pop(); }
        }
        if (hf.count > 0) (void)hf.peek(); if (hc.count > 0) (void)hc.peek();
    }
    // the restated Span.Sort: every size class (insertion sort, median of three, heapsort fallback), NaN / -0 / ties
    for (int rep = 0; rep < 300; ++rep) {
        const int n = (int)rng.below(rep % 10 == 0 ? 3000 : 70);
        std::vector<NodeDist> k((size_t)n);
        for (int i = 0; i < n; ++i) {
            float d = (float)rng.below(rep % 3 == 0 ? 5 : 100000) * 0.5f;
            const uint32_t odd = rng.below(50);
            if (odd == 0) d = std::nanf(""); else if (odd == 1) d = -0.0f; else if (odd == 2) d = -d; else if (odd == 3) d = INFINITY;
            k[(size_t)i] = NodeDist{i, d};
        }
        if (rep % 7 == 0) for (int i = 0; i < n; ++i) k[(size_t)i].dist = (float)(n - i); // descending: the quicksort's bad case
        dotnet_sort(k.data(), n);
        for (int i = 1; i < n; ++i) if (float_compare_to(k[(size_t)i - 1].dist, k[(size_t)i].dist) > 0) { std::printf("sort order broken\n"); return 2; }
    }
    // System.Random + level draw
    for (int s = 0; s < 50; ++s) {
        DotnetRandom r((int)rng.next());
        long long acc = 0;
        for (int i = 0; i < 2000; ++i) acc += level_from_uniform(r.next_single(), 0.36067376022224085);
        if (acc < 0) return 2;
    }
    std::printf("structs: heaps, sort, random: no fault\n");
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && !std::strcmp(argv[1], "parse")) {
        for (int i = 2; i < argc; ++i) {
            std::printf("%s: ", argv[i]);
            const int rc = exercise(slurp(argv[i]), true);
            if (rc) return rc;
        }
        return 0;
    }
    if (argc >= 5 && !std::strcmp(argv[1], "fuzz")) return fuzz(argv[2], std::atoi(argv[3]), (uint64_t)std::atoll(argv[4]));
    if (argc >= 3 && !std::strcmp(argv[1], "structs")) return structs((uint64_t)std::atoll(argv[2]));
    std::printf("usage: harness parse <file>... | fuzz <seed file> <iterations> <rng seed> | structs <rng seed>\n");
    return 64;
}
