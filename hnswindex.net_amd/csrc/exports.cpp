// exports.cpp -- C ABI (A): the reference's 16 `hnsw_*` exports
// (/root/reference/bindings/HNSWIndex.Native/HNSWIndexExports.cs:27-273), same names,
// signatures, return codes and padding, over the MI355X-backed HnswIndex.
#include <algorithm>
#include <thread>
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../../include/hnsw_mi355x.h"
#include "hnsw_index.h"
#include "diag.h"
#include "range_replay.h"
#include "snapshot_io.h"

using hnsw::HnswIndex;
using hnsw::Params;

namespace {
// Exports.cs:11-12,16: process-global last error and pending parameters (unsynchronised in
// the reference; a mutex here costs nothing and changes no observable behaviour).
std::mutex g_mu;
std::string g_last_error;
Params g_pending;

void set_error(const std::string &s)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_last_error = s;
}
} // namespace

#define API extern "C" __attribute__((visibility("default")))
// exclusive use of one handle (HnswIndex::mutex; hnsw_knn_query calls share it, see hnsw_index.h)
#define LOCK_INDEX(h) std::unique_lock<std::shared_mutex> index_lock_(static_cast<HnswIndex *>(h)->mutex())

API int hnsw_get_last_error_utf8(void *buf, int buf_len) // :27-39
{
    std::lock_guard<std::mutex> lk(g_mu);
    const int need = (int)g_last_error.size();
    if (buf_len > 0 && buf != nullptr) {
        int to_write = std::max(0, buf_len - 1);
        int written = std::min(need, to_write);
        std::memcpy(buf, g_last_error.data(), (size_t)written);
        static_cast<char *>(buf)[written] = 0;
    }
    return need;
}

static int parse_metric(const char *distance_metric) // :47-60
{
    if (!distance_metric) return -1;
    if (!std::strcmp(distance_metric, "sq_euclid")) return HNSWDEV_SQ_EUCLID;
    if (!std::strcmp(distance_metric, "cosine")) return HNSWDEV_COSINE;
    if (!std::strcmp(distance_metric, "ucosine")) return HNSWDEV_UCOSINE;
    if (!std::strcmp(distance_metric, "sq_euclid_i8")) return HNSWDEV_SQ_EUCLID_I8; // not in the reference: int8 rows (BASELINE config 5)
    return -1;
}

API void *hnsw_create(const char *distance_metric) // :41-65
{
    const int metric = parse_metric(distance_metric);
    if (metric < 0) {
        set_error(std::string("System.ArgumentException: Unsupported distance metric: ") + (distance_metric ? distance_metric : "(null)"));
        return nullptr;
    }
    Params p;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        p = g_pending;
    }
    std::string err;
    HnswIndex *ix = HnswIndex::create(metric, p, err);
    if (!ix) { set_error(err); return nullptr; }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_pending = Params(); // :61 reset parameters for next instance
    }
    return ix;
}

API void hnsw_free(void *handle) // :67-73
{
    if (!handle) return;
    delete static_cast<HnswIndex *>(handle);
}

API int hnsw_add(void *handle, const float *vectors, int count, int dim, int *out_ids) // :75-100
{
    if (!handle) return 0;
    if (!vectors || count <= 0 || dim <= 0) return 0;
    LOCK_INDEX(handle);
    std::string err;
    int n = static_cast<HnswIndex *>(handle)->add(vectors, count, dim, out_ids, err);
    if (n < 0) { set_error(err); return -1; }
    return n;
}

API int hnsw_remove(void *handle, const int *ids, int count) // :102-117
{
    if (!handle) return 0;
    if (!ids || count <= 0) return 0;
    LOCK_INDEX(handle);
    std::string err;
    if (static_cast<HnswIndex *>(handle)->remove(ids, count, err) < 0) { set_error(err); return -1; }
    return 0;
}

API int hnsw_knn_query(void *handle, const float *vectors, int count, int dim, int k, int *out_ids, float *out_dists) // :119-149
{
    if (!handle) return 0;
    if (count <= 0) return 0;
    if (!vectors || !out_ids || !out_dists || dim <= 0) { set_error("System.ArgumentNullException: hnsw_knn_query"); return -1; }
    std::string err;
    HnswIndex *ix = static_cast<HnswIndex *>(handle);
    {   // same-type calls overlap (README.md:64-65): shared lock, a query lane per call
        std::shared_lock<std::shared_mutex> rd(ix->mutex());
        int rc = 0;
        if (ix->knn_query_concurrent(vectors, count, dim, k, out_ids, out_dists, rc, err)) {
            if (rc < 0) { set_error(err); return -1; }
            return 0;
        }
    }
    LOCK_INDEX(handle);
    int rc = ix->knn_query(vectors, count, dim, k, out_ids, out_dists, err);
    if (rc < 0) { set_error(err); return -1; }
    return 0;
}

API void hnsw_free_results(void **ids_array, void **dists_array, int count);

API int hnsw_range_query(void *handle, const float *vectors, int count, int dim, float range, void **out_ids, void **out_dists, int *counts) // :151-197
{
    if (!handle) return 0;
    if (count <= 0) return 0;
    if (!vectors || !out_ids || !out_dists || !counts || dim <= 0) { set_error("System.ArgumentNullException: hnsw_range_query"); return -1; }
    for (int i = 0; i < count; ++i) { out_ids[i] = nullptr; out_dists[i] = nullptr; counts[i] = 0; }
    std::string err;
    std::vector<std::vector<hnsw::NodeDist>> res;
    {
        LOCK_INDEX(handle);
        if (static_cast<HnswIndex *>(handle)->range_query(vectors, count, dim, range, res, err) < 0) { set_error(err); return -1; }
    }
    const bool trace = hnsw::diag("trace", 0) != 0;
    const auto t_out0 = std::chrono::steady_clock::now();
    struct OutTimer { bool on; std::chrono::steady_clock::time_point t0; int count; ~OutTimer() { if (on) fprintf(stderr, "[hnsw trace] hnsw_range_query: handing out %d per-query arrays %.4fs\n", count, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); } } out_timer{trace, t_out0, count};
    // callee-allocated per-query arrays (Marshal.AllocHGlobal :172-173), freed by hnsw_free_results: a call of thousands of queries hands out
    // millions of results, so the copies are spread over host threads (malloc is thread-safe; a failed allocation fails the call as before)
    std::atomic<int> next{0};
    std::atomic<bool> oom{false};
    auto work = [&]() {
        for (int i0; (i0 = next.fetch_add(256, std::memory_order_relaxed)) < count;)
            for (int i = i0, e = std::min(count, i0 + 256); i < e; ++i) {
                const int n = (int)res[(size_t)i].size();
                if (n > 0) {
                    int *ids = static_cast<int *>(std::malloc(sizeof(int) * (size_t)n));
                    float *ds = static_cast<float *>(std::malloc(sizeof(float) * (size_t)n));
                    if (!ids || !ds) { std::free(ids); std::free(ds); oom.store(true); continue; }
                    for (int j = 0; j < n; ++j) { ids[j] = res[(size_t)i][(size_t)j].id; ds[j] = res[(size_t)i][(size_t)j].dist; }
                    out_ids[i] = ids;
                    out_dists[i] = ds;
                    counts[i] = n;
                }
            }
    };
    {
        const int nth = count >= 2048 ? (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 8u) : 1;
        std::vector<std::thread> pool;
        for (int t = 1; t < nth; ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
    }
    if (oom.load()) {
        hnsw_free_results(out_ids, out_dists, count);
        for (int j = 0; j < count; ++j) counts[j] = 0;
        set_error("System.OutOfMemoryException: hnsw_range_query");
        return -1;
    }
    return 0;
}

API void hnsw_free_results(void **ids_array, void **dists_array, int count) // :199-217
{
    if (!ids_array && !dists_array) return;
    for (int i = 0; i < count; ++i) {
        if (ids_array && ids_array[i]) { std::free(ids_array[i]); ids_array[i] = nullptr; }
        if (dists_array && dists_array[i]) { std::free(dists_array[i]); dists_array[i] = nullptr; }
    }
}

#define SETTER(name, field, type)                      \
    API int name(type v)                               \
    {                                                  \
        std::lock_guard<std::mutex> lk(g_mu);          \
        g_pending.field = v;                           \
        return 0;                                      \
    }
SETTER(hnsw_set_collection_size, collection_size, int)             // :219
SETTER(hnsw_set_max_edges, max_edges, int)                         // :226
SETTER(hnsw_set_max_candidates, max_candidates, int)               // :233
SETTER(hnsw_set_remove_max_candidates, remove_max_candidates, int) // :240
SETTER(hnsw_set_random_seed, random_seed, int)                     // :254
SETTER(hnsw_set_min_nn, min_nn, int)                               // :261
SETTER(hnsw_set_allow_removals, allow_removals, bool)              // :268
SETTER(hnsw_mi355x_set_device, device, int)
SETTER(hnsw_mi355x_set_insert_batch, insert_batch, int)
SETTER(hnsw_mi355x_set_remove_batch, remove_batch, int)
SETTER(hnsw_mi355x_set_search_slots, search_slots, int)
SETTER(hnsw_mi355x_set_host_threads, host_threads, int)
SETTER(hnsw_mi355x_set_device_traversal, device_traversal, int)
SETTER(hnsw_mi355x_set_devices, devices, int)

// The knobs as one versioned struct (include/hnsw_mi355x.h): the documented way to configure the backend.
API int hnsw_mi355x_default_options(hnsw_mi355x_options *out)
{
    if (!out) return -1;
    const Params d;
    *out = hnsw_mi355x_options{(uint32_t)sizeof(hnsw_mi355x_options), 0, 1, d.insert_batch, d.remove_batch, d.host_threads, d.search_slots, d.device_traversal, nullptr};
    return 0;
}
API int hnsw_mi355x_set_options(const hnsw_mi355x_options *opt)
{
    if (!opt || opt->struct_size < 8 || opt->struct_size > 4096) { set_error("hnsw_mi355x_set_options: struct_size not set"); return -1; }
    hnsw_mi355x_options o;
    (void)hnsw_mi355x_default_options(&o);
    std::memcpy(&o, opt, std::min<size_t>(opt->struct_size, sizeof o)); // an older caller's shorter struct: the rest keeps the defaults
    if (o.insert_batch == -1 || o.remove_batch < 1 || o.devices < 0 || o.devices > 64 || o.device < -1 || o.host_threads < 0 || o.search_slots < 0) {
        set_error("hnsw_mi355x_set_options: value out of range");
        return -1;
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_pending.device = o.device; g_pending.devices = o.devices; g_pending.insert_batch = o.insert_batch; g_pending.remove_batch = o.remove_batch;
        g_pending.host_threads = o.host_threads; if (o.search_slots > 0) g_pending.search_slots = o.search_slots; g_pending.device_traversal = o.device_traversal;
    }
    if (opt->struct_size >= offsetof(hnsw_mi355x_options, diagnostics) + sizeof(const char *)) hnsw::set_diag_string(o.diagnostics);
    return 0;
}

API int hnsw_set_distribution_rate(float dist_rate) // :247 -- crosses the ABI as float, widened to double
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_pending.distribution_rate = (double)dist_rate;
    return 0;
}

// ---- introspection / counters ----
// Which sources this binary was compiled from: sha256 over csrc/* and include/* (build.py: source_id()), also findable
// in the file itself behind the tag.  __graft_entry__.build() rebuilds when it differs from the tree's, and the test
// tier asserts that the library that got loaded is the tree's.
#ifndef HNSW_MI355X_BUILD_ID_STR
#define HNSW_MI355X_BUILD_ID_STR "unidentified"
#endif
extern "C" __attribute__((visibility("default"), used)) const char hnsw_mi355x_build_id_tag[] = "HNSW_MI355X_BUILD_ID=" HNSW_MI355X_BUILD_ID_STR;
API const char *hnsw_mi355x_build_id(void) { return hnsw_mi355x_build_id_tag + sizeof("HNSW_MI355X_BUILD_ID=") - 1; }

API int hnsw_mi355x_count(void *h)
{
    if (!h) return 0;
    LOCK_INDEX(h);
    return static_cast<HnswIndex *>(h)->count();
}
// measurement aid: queries resident in HBM across calls (bench.py's timed region)
API int hnsw_mi355x_set_queries(void *h, const float *queries, int count, int dim)
{
    if (!h || !queries || count <= 0 || dim <= 0) return -1;
    LOCK_INDEX(h);
    std::string err;
    if (static_cast<HnswIndex *>(h)->set_resident_queries(queries, count, dim, err) < 0) { set_error(err); return -1; }
    return 0;
}
API int hnsw_mi355x_resident_count(void *h)
{
    if (!h) return 0;
    LOCK_INDEX(h);
    return static_cast<HnswIndex *>(h)->resident_count();
}
API int hnsw_mi355x_knn_query_resident(void *h, int k, int *out_ids, float *out_dists)
{
    if (!h || !out_ids || !out_dists) return -1;
    LOCK_INDEX(h);
    std::string err;
    if (static_cast<HnswIndex *>(h)->knn_query_resident(k, out_ids, out_dists, err) < 0) { set_error(err); return -1; }
    return 0;
}
API int hnsw_mi355x_index_set_insert_batch(void *h, int max_batch)
{
    if (!h || max_batch == -1) { set_error("insert batch must be >= 1, 0 (the host's hardware threads), or -W with W >= 2"); return -1; }
    LOCK_INDEX(h);
    static_cast<HnswIndex *>(h)->set_insert_batch(max_batch);
    return 0;
}
API int hnsw_mi355x_host_parallelism(void) { return HnswIndex::host_parallelism(); }
API int hnsw_mi355x_index_insert_batch(void *h)
{
    if (!h) return 0;
    LOCK_INDEX(h);
    return static_cast<HnswIndex *>(h)->insert_batch_cap();
}
API int hnsw_mi355x_exact_window_stats(void *h, uint64_t out[4])
{
    if (!h || !out) return -1;
    LOCK_INDEX(h);
    static_cast<HnswIndex *>(h)->exact_window_stats(out);
    return 0;
}
API int hnsw_mi355x_dim(void *h)
{
    if (!h) return 0;
    LOCK_INDEX(h);
    return static_cast<HnswIndex *>(h)->dim();
}
API int hnsw_mi355x_length(void *h)
{
    if (!h) return 0;
    LOCK_INDEX(h);
    return static_cast<HnswIndex *>(h)->graph().length;
}
API int hnsw_mi355x_active_ids(void *h, int *out, int cap)
{
    if (!h || !out) return -1;
    LOCK_INDEX(h);
    const hnsw::Graph &g = static_cast<HnswIndex *>(h)->graph();
    int n = std::min(g.count, cap);
    std::memcpy(out, g.dense.data(), sizeof(int) * (size_t)n);
    return g.count;
}
API int hnsw_mi355x_entry_point(void *h)
{
    if (!h) return -1;
    LOCK_INDEX(h);
    return static_cast<HnswIndex *>(h)->graph().entry;
}
API int hnsw_mi355x_node_max_layer(void *h, int id)
{
    if (!h) return -1;
    LOCK_INDEX(h);
    const hnsw::Graph &g = static_cast<HnswIndex *>(h)->graph();
    return (id < 0 || id >= g.length) ? -1 : g.level[(size_t)id];
}
API int hnsw_mi355x_get_out_edges(void *h, int id, int layer, int *out, int cap)
{
    if (!h) return -1;
    LOCK_INDEX(h);
    const hnsw::Graph &g = static_cast<HnswIndex *>(h)->graph();
    if (id < 0 || id >= g.length || layer < 0 || layer > g.level[(size_t)id]) return -1;
    const int *l = g.list(id, layer);
    int n = std::min(l[0], cap);
    if (out && n > 0) std::memcpy(out, l + 1, sizeof(int) * (size_t)n);
    return l[0];
}
API int hnsw_mi355x_export_levels(void *h, int *out, int cap)
{
    if (!h || !out) return -1;
    LOCK_INDEX(h);
    const hnsw::Graph &g = static_cast<HnswIndex *>(h)->graph();
    int n = std::min(g.length, cap);
    std::memcpy(out, g.level.data(), sizeof(int) * (size_t)n);
    return g.length;
}
// counts[id] = out-degree of (id, layer) or -1 if the node has no such layer;
// edges[id*stride ...] = its ids (stride >= max degree + 1).
API int hnsw_mi355x_export_edges(void *h, int layer, int *counts, int *edges, int stride, int cap)
{
    if (!h || !counts || !edges || layer < 0) return -1;
    LOCK_INDEX(h);
    const hnsw::Graph &g = static_cast<HnswIndex *>(h)->graph();
    int n = std::min(g.length, cap);
    for (int i = 0; i < n; ++i) {
        if (g.level[(size_t)i] < layer) { counts[i] = -1; continue; }
        const int *l = g.list(i, layer);
        if (l[0] > stride) return -1;
        counts[i] = l[0];
        std::memcpy(edges + (size_t)i * stride, l + 1, sizeof(int) * (size_t)l[0]);
    }
    return g.length;
}
// HNSWIndex.Serialize / Deserialize (src/HNSWIndex/HNSWIndex.cs:210-229); the reference's C ABI
// does not export them, its C# API does.
API int hnsw_mi355x_serialize(void *h, const char *path_utf8)
{
    if (!h) return 0;
    LOCK_INDEX(h);
    std::string err;
    if (static_cast<HnswIndex *>(h)->serialize(path_utf8, err) < 0) { set_error(err); return -1; }
    return 0;
}
API void *hnsw_mi355x_deserialize(const char *distance_metric, const char *path_utf8)
{
    const int metric = parse_metric(distance_metric);
    if (metric < 0) {
        set_error(std::string("System.ArgumentException: Unsupported distance metric: ") + (distance_metric ? distance_metric : "(null)"));
        return nullptr;
    }
    Params p;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        p = g_pending; // backend knobs only; the HNSW parameters come from the snapshot
    }
    std::string err;
    HnswIndex *ix = HnswIndex::deserialize(metric, p, path_utf8, err);
    if (!ix) { set_error(err); return nullptr; }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_pending = Params();
    }
    return ix;
}
API int hnsw_mi355x_import_nodes(void *h, const float *rows, int n, int dim, const int *levels, int entry_point)
{
    if (!h) return 0;
    LOCK_INDEX(h);
    std::string err;
    if (static_cast<HnswIndex *>(h)->import_nodes(rows, n, dim, levels, entry_point, err) < 0) { set_error(err); return -1; }
    return 0;
}
API int hnsw_mi355x_import_edges(void *h, int layer, const int *counts, const int *edges, int stride)
{
    if (!h) return 0;
    LOCK_INDEX(h);
    std::string err;
    if (static_cast<HnswIndex *>(h)->import_edges(layer, counts, edges, stride, err) < 0) { set_error(err); return -1; }
    return 0;
}
API uint64_t hnsw_mi355x_graph_hash(void *h)
{
    if (!h) return 0;
    LOCK_INDEX(h);
    return static_cast<HnswIndex *>(h)->graph_hash();
}
API int hnsw_mi355x_get_stats(void *h, hnswdev_stats *out)
{
    if (!h || !out) return -1;
    LOCK_INDEX(h);
    static_cast<HnswIndex *>(h)->collect_stats(out); // the primary context plus the query lanes
    return 0;
}
API int hnsw_mi355x_device_count(void *h)
{
    if (!h) return 0;
    LOCK_INDEX(h);
    return static_cast<HnswIndex *>(h)->device_count();
}
API int hnsw_mi355x_get_stats_at(void *h, int context, hnswdev_stats *out)
{
    if (!h || !out) return -1;
    LOCK_INDEX(h);
    hnsw::Device *d = static_cast<HnswIndex *>(h)->device_at(context);
    if (!d) { std::memset(out, 0, sizeof(*out)); return context >= 0 && context < static_cast<HnswIndex *>(h)->device_count() ? 0 : -1; }
    d->get_stats(out);
    return 0;
}
API int hnsw_mi355x_reset_stats(void *h)
{
    if (!h) return -1;
    LOCK_INDEX(h);
    static_cast<HnswIndex *>(h)->reset_all_stats();
    return 0;
}
API int hnsw_mi355x_set_profiling(void *h, int enabled)
{
    if (!h) return -1;
    LOCK_INDEX(h);
    static_cast<HnswIndex *>(h)->set_profiling(enabled != 0);
    return 0;
}

// ---- host-logic test hooks (no device involved): the restated BCL pieces, so that the CPU
// test tier can compare them with the oracle's independent restatement ----
// Snapshot codec without a device: decode `in_path`, re-encode to `out_path`.
// info = {length, dim, count, entry, capacity, max_edges, allow_removals, random_seed}
API int hnswhost_test_snapshot_transcode(const char *in_path, const char *out_path, int *info, uint64_t *graph_hash)
{
    std::string err;
    FILE *f = std::fopen(in_path, "rb");
    if (!f) { set_error("cannot open input"); return -1; }
    std::vector<uint8_t> buf;
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize((size_t)std::max(0L, sz));
    if (sz > 0 && std::fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { std::fclose(f); set_error("short read"); return -1; }
    std::fclose(f);
    hnsw::SnapshotParams sp;
    hnsw::Graph g;
    std::vector<float> rows;
    int dim = 0;
    long long cap = 0;
    if (!hnsw::read_snapshot(buf.data(), buf.size(), sp, g, rows, dim, cap, err)) { set_error(err); return -1; }
    if (info) {
        info[0] = g.length; info[1] = dim; info[2] = g.count; info[3] = g.entry; info[4] = (int)cap;
        info[5] = sp.max_edges; info[6] = sp.allow_removals ? 1 : 0; info[7] = sp.random_seed;
    }
    if (graph_hash) *graph_hash = hnsw::graph_hash_of(g);
    if (out_path && !hnsw::write_snapshot(out_path, sp, g, rows.data(), dim, cap, err)) { set_error(err); return -1; }
    return 0;
}

API int hnswhost_test_diag(const char *name, int dflt) { return name ? hnsw::diag(name, dflt) : dflt; } // what csrc/diag.h answers right now
API void hnswhost_test_random_next(int seed, int n, int *out)
{
    hnsw::DotnetRandom r(seed);
    for (int i = 0; i < n; ++i) out[i] = r.internal_sample();
}
// NextSingle's redraw rule on an injected stream of InternalSample() values
API float hnswhost_test_next_single_from_samples(const int *samples, int n, int *used)
{
    for (int i = 0; i < n; ++i) {
        const float f = hnsw::DotnetRandom::single_of_sample(samples[i]);
        if (f < 1.0f) { *used = i + 1; return f; }
    }
    *used = n;
    return -1.0f;
}
API void hnswhost_test_random_levels(int seed, double rate, int n, int *out)
{
    hnsw::DotnetRandom r(seed);
    for (int i = 0; i < n; ++i) out[i] = hnsw::level_from_uniform(r.next_single(), rate);
}
API void hnswhost_test_sort(int *ids, float *dists, int n)
{
    std::vector<hnsw::NodeDist> k((size_t)n);
    for (int i = 0; i < n; ++i) k[(size_t)i] = hnsw::NodeDist{ids[i], dists[i]};
    hnsw::dotnet_sort(k.data(), n);
    for (int i = 0; i < n; ++i) { ids[i] = k[(size_t)i].id; dists[i] = k[(size_t)i].dist; }
}
// RangeQuery's order among equal distances (range_replay.h) on a graph handed in as layer-0 lists [count, ids...]
// of `stride` ints per node: `found` = the query's result set in ANY order; out_ids = the reference's order.
API int hnswhost_test_range_replay(const int *adj0, int stride, int max_edges0, int entry, float range, const int *found_ids, const float *found_d,
                                   int m, int *out_ids, float *out_d)
{
    struct Hit { int id; float dist; };
    std::vector<Hit> found((size_t)m);
    for (int i = 0; i < m; ++i) found[(size_t)i] = Hit{found_ids[i], found_d[i]};
    std::vector<hnsw::NodeDist> out;
    hnsw::replay_range_heaps([&](int id) { return adj0 + (size_t)id * (size_t)stride; }, max_edges0, entry, range, found.data(), m, out);
    for (size_t i = 0; i < out.size(); ++i) { out_ids[i] = out[i].id; out_d[i] = out[i].dist; }
    return (int)out.size();
}
API int hnswhost_test_heap_script(int closer_first, const int *ops, const float *d, int n, int *out_ids, float *out_d, int *popped_ids, int *n_popped)
{
    hnsw::BinaryHeap<hnsw::FartherFirst> hf;
    hnsw::BinaryHeap<hnsw::CloserFirst> hc;
    hf.reset(4); hc.reset(4);
    int np = 0;
    for (int i = 0; i < n; ++i) {
        if (ops[i] >= 0) { hnsw::NodeDist v{ops[i], d[i]}; if (closer_first) hc.push(v); else hf.push(v); }
        else if ((closer_first ? hc.count : hf.count) > 0) {
            hnsw::NodeDist v = closer_first ? hc.pop() : hf.pop();
            if (popped_ids) popped_ids[np] = v.id;
            ++np;
        }
    }
    int c = closer_first ? hc.count : hf.count;
    for (int i = 0; i < c; ++i) {
        const hnsw::NodeDist &v = closer_first ? hc.buf[(size_t)i] : hf.buf[(size_t)i];
        out_ids[i] = v.id; out_d[i] = v.dist;
    }
    if (n_popped) *n_popped = np;
    return c;
}
