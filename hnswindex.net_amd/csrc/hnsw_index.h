// hnsw_index.h -- host-side mirror of HNSWIndex<float[],float> for the Add / KnnQuery path
// (src/HNSWIndex/HNSWIndex.cs:20-29,55-78,107-137), with every distance evaluated by the
// gfx950 backend through the lock-step engine.
#pragma once
#include <memory>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "device_backend.h"
#include "host_structs.h"
#include "search_engine.h"

namespace hnsw {

// src/HNSWIndex/HNSWParameters.cs:13-55 plus the backend knobs.
struct Params {
    int max_edges = 16;
    double distribution_rate = 0.36067376022224085; // 1 / Math.Log(16)
    int min_nn = 5;
    int max_candidates = 100;
    int remove_max_candidates = 100;
    int collection_size = 65536;
    int random_seed = 31337;
    bool allow_removals = true;
    // backend
    int device = -1;         // -1: HNSW_MI355X_DEVICE or 0
    int insert_batch = 0;     // cap of a snapshot batch (which is also <= linked/16).  0 (default) = the host's hardware threads:
                              // B items searching one snapshot and linking in id order is an interleaving the reference's
                              // Parallel.For (HNSWIndex.cs:70-78) can produce iff B <= its threads, so the default stays inside
                              // the reference's outcome set on this host; larger caps (the 65 536-item snapshots of rounds 1-4)
                              // are opt-in.  1 = strictly sequential inserts; -W = the sequential graph through speculative
                              // windows of W items (insert_exact_window)
    int remove_batch = 1;     // > 1: hnsw_remove takes removals with disjoint neighbourhoods together (snapshot batches of up to this many)
    int search_slots = 16384;
    int host_threads = 0;    // 0: min(hardware threads, 16)
    int device_traversal = 1; // 1: graph-resident search kernel; 0: host lock-step traversal
    int devices = 0;          // 0: HNSW_MI355X_DEVICES or 1; > 1: KnnQuery shards its queries over this many device contexts (replicas of rows + graph)
};

class HnswIndex {
public:
    static HnswIndex *create(int metric, const Params &p, std::string &err);
    ~HnswIndex();

    // hnsw_add: returns number of ids written or -1.
    int add(const float *vectors, int count, int dim, int *out_ids, std::string &err);
    // hnsw_knn_query: 0 or -1.
    int knn_query(const float *queries, int count, int dim, int k, int *out_ids, float *out_dists, std::string &err);

    // Measurement aid: upload a query set once (resident in HBM), then run KnnQuery on it any
    // number of times without host->device traffic for the inputs.
    int set_resident_queries(const float *queries, int count, int dim, std::string &err, bool streamed = false);
    int knn_query_resident(int k, int *out_ids, float *out_dists, std::string &err);

    // hnsw_range_query (HNSWIndex.RangeQuery, src/HNSWIndex/HNSWIndex.cs:144-168): per query the
    // in-range results ordered by distance.  Host lock-step traversal.
    int range_query(const float *queries, int count, int dim, float range, std::vector<std::vector<NodeDist>> &out, std::string &err);

    // hnsw_remove (HNSWIndex.Remove, src/HNSWIndex/HNSWIndex.cs:83-102), ids in order.
    int remove(const int *ids, int count, std::string &err);

    // HNSWIndex.Serialize / Deserialize (src/HNSWIndex/HNSWIndex.cs:210-229): the reference's
    // protobuf-net snapshot (csrc/snapshot_io.h).  `backend` supplies the device knobs only;
    // the HNSW parameters come from the file.
    int serialize(const char *path, std::string &err);
    static HnswIndex *deserialize(int metric, const Params &backend, const char *path, std::string &err);

    // hnsw_mi355x_import_nodes / _edges: a graph built elsewhere into an empty index (see include/hnsw_mi355x.h)
    int import_nodes(const float *rows, int n, int dim, const int *levels, int entry_point, std::string &err);
    int import_edges(int layer, const int *counts, const int *edges, int stride, std::string &err);

    int count() const { return graph_.count; }
    int resident_count() const { return resident_queries_; }
    int device_count() const { return p_.devices; }
    Device *device_at(int g) { return g == 0 ? dev_.get() : (g - 1 < (int)replicas_.size() ? replicas_[(size_t)g - 1].get() : nullptr); }
    int dim() const { return dim_; }
    // The host copy of the graph.  After a device-linked Add the neighbour lists live in the HBM
    // mirror only and are fetched back here on demand.
    const Graph &graph() { std::string e; (void)refresh_host_lists(e); return graph_; }
    Device *device() { return dev_.get(); }
    uint64_t graph_hash();
    void set_insert_batch(int v) { p_.insert_batch = v; }
    // Threads this process may run on (std::thread::hardware_concurrency(): the affinity mask on Linux) = what bounds the items a
    // Parallel.For on this host holds in flight.  (.NET also clamps Environment.ProcessorCount to a cgroup CPU quota; a
    // container host that wants that bound passes it to hnsw_mi355x_set_insert_batch itself.)
    static int host_parallelism();
    int insert_batch_cap() const { return p_.insert_batch == 0 ? host_parallelism() : p_.insert_batch; }
    void exact_window_stats(uint64_t out[4]) const { out[0] = xw_rounds_; out[1] = xw_searches_; out[2] = xw_alone_; out[3] = xw_linked_; }
    void set_profiling(bool on)
    {
        profiling_ = on;
        if (dev_) dev_->set_profiling(on);
        for (auto &r : replicas_) r->set_profiling(on);
        for (auto &l : lanes_) if (l) l->set_profiling(on);
    }

    // Calls on one handle.  The reference promises that operations of one type may overlap on an index
    // (/root/reference/README.md:64-65; BatchKnnQuery / Add(List) are Parallel.For, HNSWIndex.cs:70-78,129-137).  Here:
    //   * hnsw_knn_query calls DO overlap: each takes the index lock shared and a query lane of its own -- a Device view
    //     (stream, resident query set, per-wave scratch) that borrows the rows and the graph mirror -- so two host
    //     threads' launches run side by side on the GPU, the head of one filling the tail of the other
    //     (knn_query_concurrent; whatever it cannot serve -- sharded indices, host traversal, a graph mirror that is
    //     not current, hand-backs -- goes through the exclusive path);
    //   * everything else (Add, Remove, RangeQuery, ...) takes the lock exclusively: those calls already fan out over
    //     the whole GPU and own the index's staging buffers.
    // Calls on different handles run concurrently.
    std::shared_mutex &mutex() { return mu_; }
    // 1: answered (rc = the call's return value); 0: not eligible, take the exclusive path.  Call with the lock held SHARED.
    int knn_query_concurrent(const float *queries, int count, int dim, int k, int *out_ids, float *out_dists, int &rc, std::string &err);
    void collect_stats(hnswdev_stats *out);
    void reset_all_stats();

private:
    std::shared_mutex mu_;
    static constexpr int kLanes = 2;
    std::unique_ptr<Device> lanes_[kLanes];
    bool lane_busy_[kLanes] = {false, false};
    std::mutex lane_mu_;
    std::condition_variable lane_cv_;
    HnswIndex() = default;
    // A device failure in the middle of an Add leaves appended nodes without rows / links: the
    // index then refuses every further call with the original message.
    std::string failed_msg_;
    bool failed(std::string &err) const { if (failed_msg_.empty()) return false; err = failed_msg_; return true; }
    int fail(const std::string &why, std::string &err)
    {
        failed_msg_ = "HNSWIndex MI355X backend: the index is unusable after a failed Add / Remove (" + why + ")";
        err = failed_msg_;
        return -1;
    }
    bool ensure_dim(int dim, std::string &err);
    bool ensure_capacity(long long need, std::string &err);
    bool insert_batch(const std::vector<int> &bid, std::string &err);
    bool insert_exact_window(const std::vector<int> &fresh, int &p, int W, bool background, std::string &err);
    // Selected neighbour ids per (batch item, layer).  Device results are read in place from the
    // context's pinned buffers (layer 0: slot = item; layer L >= 1: slot upper_base[item] + L - 1);
    // items processed on the host (lock-step mode, hand-backs) carry their own lists.
    struct Selection {
        int n = 0, n_upper = 0;
        Device::InsertResults dev{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
        std::vector<int> upper_base;
        std::vector<char> has_own;
        std::vector<std::vector<std::vector<int>>> own;
        inline void get(int i, int layer, const int *&ids, int &cnt) const
        {
            if (has_own[(size_t)i]) { const std::vector<int> &v = own[(size_t)i][(size_t)layer]; ids = v.data(); cnt = (int)v.size(); return; }
            if (layer == 0) { ids = dev.sel0 + (size_t)i * dev.sel_stride; cnt = dev.cnt0[i]; }
            else { const size_t s = (size_t)(upper_base[(size_t)i] + layer - 1); ids = dev.selU + s * dev.sel_stride; cnt = dev.cntU[s]; }
        }
    };
    bool search_half_lockstep(const std::vector<int> &bid, const std::vector<int> &items, Selection &sel, std::string &err);
    bool search_half_device(const std::vector<int> &bid, Selection &sel, std::string &err);
    bool link_half_lockstep(const std::vector<int> &bid, const Selection &sel, std::string &err);
    bool link_half_device(const std::vector<int> &bid, const Selection &sel, std::string &err);
    bool link_prefix_begin(const std::vector<int> &bid, const Selection &sel, int set, std::string &err);
    bool sync_graph(std::string &err);
    // ---- query sharding over several devices inside one process (hnsw_mi355x_set_devices) ----
    // BatchKnnQuery is a Parallel.For over independent read-only searches (HNSWIndex.cs:129-137): with n contexts,
    // context g answers queries [g nq / n, (g + 1) nq / n) on its own replica of the rows and the graph mirror and
    // writes its slice of the caller's arrays.  The primary (dev_) builds; replicas are brought up to date device to
    // device (Device::clone_from) when the graph has changed since they were last used.
    std::vector<std::unique_ptr<Device>> replicas_; // contexts 1 .. devices - 1
    std::vector<uint64_t> replica_epoch_;
    uint64_t graph_epoch_ = 1;                       // bumped by everything that changes rows or lists
    std::vector<long long> shard_lo_;                // resident query set: shard bounds (devices + 1 entries)
    bool sharded_resident_ = false;
    Device *context(int g) { return g == 0 ? dev_.get() : replicas_[(size_t)g - 1].get(); }
    bool ensure_replicas(bool clone, std::string &err);
    int knn_query_sharded(int k, int *out_ids, float *out_dists, std::string &err);
    bool refresh_host_lists(std::string &err);
    int knn_query_device(const float *queries, int count, int k, int *out_ids, float *out_dists, std::string &err);
    int knn_query_lockstep(const int *which, int count, int k, int *out_ids, float *out_dists, std::string &err);
    int range_query_lockstep(const int *which, int count, float range, std::vector<std::vector<NodeDist>> &out, std::string &err);
    int range_query_device(int count, float range, std::vector<std::vector<NodeDist>> &out, std::string &err);
    int remove_batched(const int *ids, int count, std::string &err);

    int metric_ = 0;
    int dim_ = 0; // fixed by the first add (the reference takes it from the arrays)
    Params p_;
    Graph graph_;
    DotnetRandom rng_;
    std::unique_ptr<Device> dev_;
    std::unique_ptr<LockStepEngine> engine_;
    LockStepEngine *engine();
    int engine_stride_ = 0;
    long long capacity_ = 0; // GraphData.Capacity (doubling, GraphData.cs:98-111)
    int skipped_ = 0;
    int device_ordinal_ = 0;
    int threads_ = 1;
    bool profiling_ = false;
    int resident_queries_ = 0;
    bool graph_dirty_ = true;       // HBM mirror needs a full re-upload
    // in-edge sets for Remove (content only, see remove()): built by transposing the out-lists on the first removal
    // after anything else changed the graph, kept current by the removals themselves
    std::vector<std::vector<int>> in0_;
    std::unordered_map<uint64_t, std::vector<int>> inU_;
    bool in_valid_ = false;
    // exact-window Add: per adjacency list the sequence number of the last insert that wrote it (0 = never);
    // seq_ = inserts linked so far.  Only compared within one Add call; monotone across calls.
    std::vector<uint32_t> mod0_;
    std::unordered_map<uint64_t, uint32_t> modU_;
    uint32_t seq_ = 0;
    uint64_t xw_rounds_ = 0, xw_searches_ = 0, xw_alone_ = 0, xw_linked_ = 0, xw_pairs_ = 0;
    double xw_prefix_ema_ = 16.0;                          // items linked per round, recent average (sizes the next window)
    struct XwChange { int t, e; bool known; };             // first change of a layer-0 list in the current round: window item, selection entry
    std::unordered_map<int, XwChange> xw_first_;
    std::unordered_map<int, int> xw_second_;               // ... and the item that touches it next
    std::vector<int> pa_, pb_;                             // id<->id distances asked for (reader row, gained / lost row)
    std::vector<uint32_t> pfar_;
    std::vector<std::pair<int, int>> powner_;              // (reader item, writer item)
    std::vector<float> pd_; // exact-window statistics (hnsw_mi355x_exact_window_stats)
    bool host_lists_stale_ = false; // the HBM mirror holds newer neighbour lists than graph_ (device-linked Add)
    long long dev_pool_len_ = 0;    // pool ints already mirrored
    std::vector<int> grp_of_node0_; // link half: group index per layer-0 neighbour (-1 = none)
    std::vector<int> lk_rows_, lk_node_, lk_layer_, lk_cnt_, lk_off_, lk_items_, lk_fill_, lk_out_; // link half: per-batch work arrays
    std::vector<std::pair<int, int>> lk_seq_;
};

} // namespace hnsw
