// device_backend.h -- internal C++ face of the gfx950 distance backend.
// The C ABI (include/hnsw_mi355x.h, hnswdev_*) and the host driver (search_engine.cpp,
// hnsw_index.cpp) both sit on this class.  Nothing here computes a distance on the CPU.
#pragma once
#include <cstddef>
#include <atomic>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <vector>
#include <thread>

#include "../../include/hnsw_mi355x.h"

namespace hnsw {

// Last error: one process-wide string (hnswdev_last_error: creation failures have no context yet)
// and one per context (hnswdev_ctx_last_error).  set_dev_error() files the message under the context
// the calling thread is currently inside (ErrorScope), if any.
void set_dev_error(const std::string &msg);
std::string get_dev_error();
class Device;
struct ErrorScope { // RAII: the calling thread is inside a call on `d`
    explicit ErrorScope(Device *d);
    ~ErrorScope();
    Device *prev;
};

// One lock-step "step" worth of work for up to nslots concurrent searches.
// The host driver fills one packed record per slot in pinned host memory,
//     rec[s*rec_stride + 0]      = cnt   : number of candidate ids this step
//     rec[s*rec_stride + 1]      = qidx  : >= 0 resident query index; < 0: ~row id (id<->id)
//     rec[s*rec_stride + 2 ...]  = ids   : candidate row ids (capacity = stride)
// launch_step() moves the used prefix to HBM with ONE async H2D copy, runs the kernel on
// device-resident inputs, and brings the distances back with ONE async D2H copy -- all on
// the context's stream.  (Letting the kernel read pinned host memory directly was measured
// 2-3x slower: tools/kbench.hip, DESIGN.md "Step buffers".)
//   slot s evaluates metric(row[ids[c]], Q(s)) for c < cnt
struct StepBuffers {
    enum { kHeader = 16 }; // floats in front of the distances; word 0 = the kernel's guard flag, so one D2H copy brings both
    int nslots = 0, stride = 0, rec_stride = 0;
    int *rec = nullptr;     // pinned host, nslots * rec_stride
    float *dist = nullptr;  // pinned host, nslots * stride (dist - kHeader is the allocation)
    int *d_rec = nullptr;   // HBM mirrors
    float *d_dist = nullptr; // kHeader + nslots * stride
    void *done = nullptr;      // hipEvent_t recorded after the D2H copy
    void *t0 = nullptr, *t1 = nullptr; // hipEvent_t pair around the kernel when profiling
    bool timed = false, in_flight = false;
    uint64_t evals = 0;
};

// One traversal for the graph-resident search kernel: greedy descent from `entry` at
// `entry_layer` down to (exclusive) `search_layer`, then the beam search at `search_layer`.
struct SearchJob {
    int qref;         // >= 0 resident query index; < 0: ~row id
    int entry;        // entry node id
    int entry_layer;  // layer the descent starts at (== search_layer: no descent)
    int search_layer; // layer of the beam search
    int aux;          // insert search: index of the item's first upper-layer output slot (-1: none)
    int stop_layer;   // insert search: the last layer this job searches (0 = all the way down; the exact-window Add
                      // runs the upper layers of a multi-layer item ahead of time, and its layer 0 as a job of its own)
};
// RangeQuery: state of a job's result list after the finishing kernels (dk_range_finish.h).  The list in the arena is ...
constexpr int kRangeFinal = 0;    // ... in the reference's order
constexpr int kRangeTied = 2;     // ... ascending, with equal distances in it: to be replayed (range_replay_kernel did not: the host does)
constexpr int kRangeHostSort = 5; // ... as found: too long for the device ranking, or a -0 distance (host: sort, replay if tied)

struct SearchHit {
    int id;
    float dist;
};

class Device {
public:
    static Device *create(int device, int dim, int metric, long long capacity);
    // A second context on the same GPU that BORROWS the primary's stored rows and graph mirror and owns everything
    // a query needs besides (stream, resident query set, per-wave scratch, staging): two hnsw_knn_query calls on one
    // index then run side by side, the head of one launch filling the tail of the other.  rebind() refreshes the
    // borrowed pointers (the primary may have grown); the caller guarantees the primary is not being written.
    static Device *create_view(Device *primary);
    void rebind(const Device *primary);
    ~Device();

    int dim() const { return dim_; }
    int metric() const { return metric_; }
    long long capacity() const { return capacity_; }

    bool reserve(long long capacity);
    bool upload_rows(int first_id, int n, const float *rows);
    bool download_rows(int first_id, int n, float *rows);
    // The same upload in the background: a helper thread stages the rows through its own pinned buffers
    // and stream while the caller already works on the rows that have landed.  begin() returns at once;
    // wait(upto) blocks until every row with id < upto is resident (upto < 0: all of them, and the
    // helper has finished).  `rows` stays borrowed until wait(-1) returned.  Float metrics only.
    bool upload_rows_begin(int first_id, int n, const float *rows);
    bool upload_rows_wait(long long upto);
    // Replaces the resident query set (nq x dim); norms for cosine computed on device.
    bool set_queries(const float *queries, int nq);
    // The same, with only the first `head` rows uploaded now: the rest follows behind the traversal launch of the next
    // search_batch (which must come next), the kernel waiting for rows that have not landed yet.  Falls back to
    // set_queries where that does not apply (cosine / int8 rows, head >= nq).
    bool set_queries_streamed(const float *queries, int nq, int head);
    // Forgets a tail that no launch picked up (an error between the two calls): the resident set is then empty.
    void cancel_streamed() { if (tail_.n > 0) { tail_.n = 0; n_queries_ = 0; } }
    // ---- replicas (query sharding over the GPUs of one node, one context per GPU in one process) ----
    // Makes this context a replica of `src`: stored rows (only those it does not hold yet when `rows_from` >= 0 says
    // where they start to differ), per-row norms and the whole graph mirror, copied device to device
    // (hipMemcpyPeerAsync: over xGMI between two GPUs).  pool_len: used ints of src's upper-layer pool.
    bool clone_from(Device *src, long long pool_len);
    // Rows [first, first + n) of src's resident query set land at [at, at + n) of this context's (which is grown to
    // `total` rows and then counts `total` queries): the lock-step fallback gathers every shard on the primary.
    bool adopt_queries(Device *src, long long first, long long n, long long at, long long total);
    int ordinal() const { return device_; }

    StepBuffers *alloc_step(int nslots, int stride);
    void free_step(StepBuffers *sb);
    // Asynchronous: one kernel over slots [0, nslots_used).  `evals` = sum of cnt (for stats).
    bool launch_step(StepBuffers *sb, int nslots_used, uint64_t evals);
    bool wait_step(StepBuffers *sb);
    bool sync();
    // Makes this context's HIP device current on the calling thread.
    bool bind_thread() { return bind(); }

    // --- graph-resident traversal (DESIGN.md "Graph-resident search") ---
    // Mirrors the host adjacency in HBM (layer 0: n x stride0 ints [count, e...]; upper layers:
    // per-node offset into a pool of strideU-int blocks).  Full replace.
    long long graph_nodes() const { return g_n_; }
    // capacity of the kernels' id / distance scratch: the longest adjacency list, rounded up to 8
    int nbcap() const { int m = (g_stride0_ > g_strideU_ ? g_stride0_ : g_strideU_) - 1; m = (m + 7) & ~7; return m < 8 ? 8 : m; }
    // Can the graph-resident kernels run this shape?  (LDS budget for beam width k at this dim;
    // MaxEdges <= 63.)  When not, callers use the host lock-step traversal instead.
    bool traversal_fits(int k, bool with_heuristic, int max_edges) const;
    bool set_graph(const int *adj0, long long n, int stride0, const int *level, const int64_t *upper, const int *pool,
                   long long pool_len, int strideU);
    // Runs njobs KnnQuery traversals with beam width k and returns the first k_out results of
    // the stable distance order (padded with -1 / NaN); out_flag: 1 where the candidate heap
    // outgrew LDS + spill capacity (caller re-runs that job on the lock-step path).  Synchronous.
    // two_heap: the exact two-heap traversal for every job (jobs with aux == -2: the entry point is filtered out of the
    // results and the output is the result heap's ARRAY, the removal search's return value)
    bool search_batch(const SearchJob *jobs, int njobs, int k, int k_out, int *out_ids, float *out_d, int *out_flag, bool keep_repeat_flag = false,
                      bool two_heap = false);
    // search_batch for KnnQuery's own jobs -- resident query i from the entry point, i = 0 .. nq-1: the job array is written
    // on this side and stays on the device while entry point and top layer do not change (a call then uploads no jobs)
    bool search_queries(int nq, int entry, int entry_layer, int k, int k_out, int *out_ids, float *out_d, int *out_flag);
    // Remove, second half, for the `n` affected nodes of one (removed node, layer) step (graph_relink_kernel): per node
    // the new neighbour selection out_sel[i * sel_stride ..][0 .. out_cnt[i]); out_flag[i] = 1: this node's answer
    // depends on the heap-array order of the candidates (the caller repeats the step on the lock-step path).
    // Reads the HBM graph mirror; writes nothing to it.  Synchronous.
    // heap_order: `cands` are in the reference's heap-array order (search_batch with two_heap): nothing is flagged.
    // Jobs i < n: affected[i] at layer[i], un-linking removed[i], with the search candidates of step[i]:
    // cands[cand_off[s] ..][0 .. cand_cnt[s]) for the nsteps (removed node, layer) steps.  max_edges0 = MaxEdges(0).
    bool relink_batch(const int *affected, const int *layer, const int *removed, const int *step, int n, const int *cands, const int *cand_off,
                      const int *cand_cnt, int nsteps, int max_edges0, int *out_sel, int *out_cnt, int *out_flag, int sel_stride,
                      bool heap_order = false);
    // Overwrites adjacency lists of the mirror: records [node, layer, count, ids...] of `row_stride` ints; the lists are
    // marked as NOT being a heuristic's ordered output (the link kernel's tested-prefix shortcut starts from 0).
    bool patch_lists(const int *recs, int nrows, int row_stride);
    // Insert, search half, fused on the device: for every job (new item) the descent from
    // (entry, entry_layer) to search_layer = the item's first layer, then on every layer from there
    // down to 0 the traversal with beam k (= MaxCandidates) + RelativeNeighborPruning, the next
    // layer entering at selected[0].  jobs[].qref must be ~item_id; jobs[].aux = the item's first
    // upper-layer output slot (layer L >= 1 goes to slot aux + L - 1), n_upper = slots in total.
    // Results stay in pinned host buffers owned by the context (valid until the next call):
    //   sel0 [njobs x sel_stride] / cnt0 [njobs]: layer 0;  selU [n_upper x sel_stride] / cntU;
    //   flag [njobs]: 1 = handed back.  sel_stride = max_edges0 = MaxEdges(0) = 2M.
    struct InsertResults {
        const int *sel0, *cnt0, *selU, *cntU, *flag;
        int sel_stride;
    };
    // read_log_cap > 0 (the reference-exact windowed Add): every job also records which adjacency lists its searches
    // read -- *read_log = njobs records of read_log_cap ints [n, entries...], a marker -(layer + 1) in front of each
    // layer's node ids, n > read_log_cap - 1 on overflow -- and the selected ids come back with the flags (one wait).
    // The dry run of every selected entry's back-edge append (graph_link_dry_sel_kernel) follows in the same stream:
    // dry0 / dryU, shaped like sel0 / selU, 0 = the append would leave that neighbour's list reading as it does.
    struct WindowExtras {
        int read_log_cap;       // in
        const int *upper_owner; // in: job index per upper-layer output slot (n_upper)
        const int *read_log, *dry0, *dryU; // out: pinned, valid until the next call
        const int *drop0;    // out: per layer-0 selection entry, up to three ids the neighbour's list would lose (dry0's code says how many)
        const int *repeated; // out: per job, 1 = a layer was answered by the exact two-heap traversal (equal distances)
    };
    bool insert_search_batch(const SearchJob *jobs, int njobs, int k, int max_edges0, int n_upper, InsertResults *res, WindowExtras *win = nullptr);
    // insert_search_batch brings back the flags only; this fetches the selected ids into the arrays `res`
    // names (the device-side link half never needs them on the host).
    bool fetch_insert_selections(const InsertResults *res);
    // Keeps the HBM graph mirror in step with nodes appended on the host since the last call:
    // levels / upper offsets of nodes [first, first+n) and the pool tail [pool_from, pool_len).
    // Returns false (with no error set) when capacity is exceeded: caller falls back to set_graph.
    bool graph_append_nodes(long long first, long long n, const int *level, const int64_t *upper, const int *pool,
                            long long pool_from, long long pool_len, bool *need_full_sync);
    // Insert, link half, on the HBM graph mirror (one launch):
    //  rows:   nrows records [node, layer, cnt, ids...] (row_stride ints): OutEdges[layer] = selected
    //  groups: for group g, node g_node[g] / layer g_layer[g] receives the back-edge appends
    //          g_items[g_off[g] .. g_off[g+1]) in order, pruning on overflow (PruneOverflow).
    //  out_lists: ngroups x list_stride ints [cnt, ids...]: the final adjacency list of every group.
    bool link_batch(const int *rows, int nrows, int row_stride, const int *g_node, const int *g_layer, const int *g_off,
                    const int *g_items, int ngroups, int max_edges0, int *out_lists, int list_stride);
    // The same in two halves, so that the host can prepare the next sub-batch while this one runs:
    // begin copies the inputs to one of two pinned staging sets and enqueues copy-in, kernels and
    // copy-out on the stream; finish waits for that set and returns its lists (valid until the set
    // is used again).  Sub-batches are applied in the order they were begun.
    bool link_batch_begin(int set, const int *rows, int nrows, int row_stride, const int *g_node, const int *g_layer, const int *g_off,
                          const int *g_items, int ngroups, int max_edges0, int list_stride, bool want_lists = true);
    bool link_batch_finish(int set, const int **out_lists);
    // Dry run of n single appends [node, layer, item] on the mirror as it stands (graph_link_dry_kernel): changed[i] = 1
    // when list (node, layer) would hold another sequence of ids afterwards.  Nothing is written.  Synchronous.
    bool link_dry_run(const int *jobs3, int n, int max_edges0, int *changed);
    // Link half of the batch whose insert_search_batch just ran (its jobs and selections are still on
    // the device): own lists, grouping of the back-edge appends and the appends / prunes, all on the
    // device, nothing copied back.  Requires that no job of that batch was handed back.
    bool link_batch_planned(int njobs, int n_upper, int max_edges0);
    // The adjacency mirror back to the host (adj0: n x stride0 ints, pool: pool_len ints).
    bool download_graph(int *adj0, long long n, int *pool, long long pool_len);

    // ---- the inner boundary as a foreign host drives it (hnswdev_step_* / hnswdev_dist_*) ----
    // Two step-buffer sets owned by the context (pinned host + HBM mirror), (re)allocated only when
    // they grow: the host fills set A's records while the GPU works on set B.  The lock-step engine
    // of this library is a client of exactly these calls.
    bool step_buffers(int set, int nslots, int stride, int **rec, float **dist, bool internal = false);
    bool step_submit(int set, int nslots_used); // async: H2D, kernel, D2H on the context's stream
    bool step_wait(int set);                    // distances of that set are in its pinned array
    long long resident_queries() const { return n_queries_; }
    // Synchronous conveniences on the same buffers (ids are guarded on the device).  queries ==
    // nullptr: the resident set (hnswdev_set_queries) is used, otherwise it is replaced first.
    bool dist_query_batch(const float *queries, int nq, const int *offsets, const int *ids, float *out);
    bool dist_pair_batch(const int *a, const int *b, int n, float *out);

    // C-ABI staging of a host graph given layer by layer (hnswdev_graph_*)
    bool graph_begin(int n, int max_edges, const int *levels);
    bool graph_set_layer(int layer, const int *counts, const int *edges, int stride);
    bool graph_commit();
    bool knn_search(const float *queries, int nq, int entry_point, int k_beam, int k_out, int *out_ids, float *out_d, int *out_flag);
    // RangeQuery on the device (graph_range_kernel): per job the results within `range`, UNSORTED, at
    // found[off[i] .. off[i] + cnt[i]); flag[i] = 1: handed back (cnt 0).  jobs[].qref must name a resident query.
    struct RangeResults {
        std::vector<unsigned long long> off;
        std::vector<int> cnt, flag, entry; // entry[i]: the layer-0 entry node FindEntryPointQuery reached
        std::vector<int> state;            // kRangeFinal: found[] holds the reference's order; kRangeTied: ascending, equal distances to replay; kRangeHostSort: as found
        SearchHit *found = nullptr;        // the lists, in pinned memory the context owns: valid until its next range_batch
        size_t found_n = 0;
    };
    bool range_batch(const SearchJob *jobs, int njobs, float range, RangeResults *res);
    // the C ABI's form: sorted per query, equal distances handed back; results kept until the next call
    bool range_search(const float *queries, int nq, int entry_point, float range, int *out_counts, int *out_flags);
    bool range_results(int *out_ids, float *out_d);

    void set_profiling(bool on) { profiling_ = on; }

    // search launches of a call that shares the chip with another call (query lanes) keep no idle waves behind as shadows

    void set_shadows_allowed(bool on) { shadows_allowed_ = on; }
    void get_stats(hnswdev_stats *out);
    void reset_stats();

    // C-ABI callers: one call at a time per context (hnswdev_* exports hold this), and the
    // context's own last-error string.
    std::mutex &mutex() { return mu_; }
    void set_error(const std::string &msg) { std::lock_guard<std::mutex> lk(err_mu_); err_ = msg; }
    std::string error() { std::lock_guard<std::mutex> lk(err_mu_); return err_; }

private:
    Device() = default;
    std::mutex mu_, err_mu_;
    std::string err_;
    bool bind();
    int device_ = 0, dim_ = 0, metric_ = 0;
    int pitch_ = 0; // 32-bit words per stored row / resident query (== dim_ for the float metrics; the int8 record otherwise)
    float *q_stage_ = nullptr; // int8: float staging area on the device (quantise on upload, dequantise on download)
    size_t q_stage_cap_ = 0;
    long long capacity_ = 0;
    long long n_rows_hw_ = 0; // high-water mark of uploaded rows (id validation)
    float *d_rows_ = nullptr;
    double *d_row_sn_ = nullptr; // cosine: sqrt((double)|row|^2_f32)
    float *d_queries_ = nullptr;
    double *d_q_sn_ = nullptr;
    long long q_capacity_ = 0, n_queries_ = 0;
    // graph mirror
    int *g_adj0_ = nullptr, *g_level_ = nullptr, *g_pool_ = nullptr;
    int64_t *g_upper_ = nullptr;
    long long g_n_ = 0, g_cap_n_ = 0, g_pool_cap_ = 0;
    // per adjacency list: how many leading entries are the ordered, mutually tested output of a
    // RelativeNeighborPruning run (graph_link_kernel's shortcut); 0 = unknown
    int *g_tested0_ = nullptr, *g_testedU_ = nullptr;
    int g_stride0_ = 0, g_strideU_ = 0;
    // search scratch
    unsigned *s_visited_ = nullptr;
    size_t s_visited_bytes_ = 0;
    int *s_jobctr_ = nullptr; // persistent launches: next job
    int *lp_slot_[3] = {nullptr, nullptr, nullptr}; // device-side link grouping: per list slot count / fill / offset
    long long lp_slots_ = 0;
    int *lp_grp_[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // per group node / layer / start / count; items as filed; items in batch order
    size_t lp_grp_cap_[6] = {0, 0, 0, 0, 0, 0};
    int *lp_counters_ = nullptr;
    int last_insert_jobs_ = 0, last_insert_upper_ = 0, last_insert_stride_ = 0; // what insert_search_batch left on the device
    int fetch_njobs_ = 0, fetch_nupper_ = 0;                                    // ... and what fetch_insert_selections would copy
    int *s_vistab_ = nullptr; // per-wave visited-id hash tables
    size_t s_vistab_cap_ = 0;
    int s_vistab_each_ = 0;
    bool visited_table(size_t vis_bytes_per_job, int k, int **out, int *out_cap, int min_cap = 512, bool allow_hash = true);
    int num_cu_ = 256;
    // Persistent launches never use more than 16 one-wave blocks per CU (the traversal kernels need
    // >= 128 VGPRs): the per-wave scratch (visited bitsets, spill areas, logs) is sized for that.
    int max_slots() const { return num_cu_ * max_waves_per_cu(); }
    static int max_waves_per_cu(); // persistent waves per CU the per-wave scratch is sized for
    SearchJob *s_jobs_ = nullptr;
    SearchHit *s_hits_ = nullptr;
    int *s_cnt_ = nullptr, *s_flag_ = nullptr;
    unsigned long long *s_evals_ = nullptr;
    size_t s_jobs_cap_ = 0, s_hits_cap_ = 0;
    int *s_sel_ = nullptr, *s_lcnt_ = nullptr, *s_selU_ = nullptr, *s_cntU_ = nullptr, *s_iflag_ = nullptr;
    size_t s_sel_cap_ = 0, s_lcnt_cap_ = 0, s_selU_cap_ = 0, s_cntU_cap_ = 0, s_iflag_cap_ = 0;
    void *h_res_ = nullptr; // pinned: results of insert_search_batch
    size_t h_res_cap_ = 0;
    int *s_rl_ = nullptr; // relink / patch staging on the device
    size_t s_rl_cap_ = 0;
    int *s_order_ = nullptr; // insert search: processing order of a batch's jobs
    size_t s_order_cap_ = 0;
    int *s_rlog_ = nullptr; // insert search: per-job read logs (exact-window Add)
    size_t s_rlog_cap_ = 0;
    bool is_view_ = false;
    struct QueryTail { const float *src = nullptr; long long first = 0, n = 0; } tail_; // set_queries_streamed: rows still on the host
    int *h_ready_ = nullptr, *d_ready_ = nullptr; // rows of the query set that have landed (host memory, read by the kernel)
    void *copy_stream_ = nullptr;
    bool upload_tail();
    int *s_dry_ = nullptr;  // link_dry_run: [jobs | flags]
    size_t s_dry_cap_ = 0;
    int *s_wdry_ = nullptr; // windowed insert search: upper_owner
    size_t s_wdry_cap_ = 0;
    int *s_win_ = nullptr;  // windowed insert search: every output of the launch, laid out like the pinned result block
    size_t s_win_cap_ = 0;
    SearchHit *s_spill_ = nullptr;
    size_t s_spill_cap_ = 0;
    SearchHit *s_arena_ = nullptr; // range search: the launch's results, packed
    size_t s_arena_cap_ = 0;
    unsigned long long *s_roff_ = nullptr, *s_arena_used_ = nullptr;
    int *s_rentry_ = nullptr;
    SearchHit *h_range_ = nullptr;     // RangeQuery results on the host (pinned, grown on demand, filled by ONE device-to-host copy per launch)
    size_t h_range_cap_ = 0;
    bool range_host_room(size_t entries, size_t keep);
    int *s_rstate_ = nullptr, *s_rtied_ = nullptr, *s_rfin_ctr_ = nullptr; // RangeQuery's finishing kernels: per-job state, the tied jobs, two job counters
    SearchHit *s_rlists_ = nullptr; // range search: long per-wave result lists for the few jobs that outgrow s_spill_'s
    size_t s_rlists_cap_ = 0;
    double range_hint_ = 48.0;      // results per query of the last range search (sizes the next arena)
    size_t s_roff_cap_ = 0;
    std::vector<SearchHit> abi_range_; // hnswdev_range_search's results until hnswdev_range_results
    int *s_lk_[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t s_lk_cap_[5] = {0, 0, 0, 0, 0};
    bool ensure_search_scratch(long long chunk, long long slots, int k, size_t vis_bytes_per_job);
    void *pinned_stage(size_t bytes);
    bool staged_upload(float *dst, const float *src, size_t bytes);
    void *up_pin_[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // staged_upload: two pinned chunks per helper thread
    void *up_ev_[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool up_busy_[8] = {false, false, false, false, false, false, false, false};
    struct HostGraphStage;
    HostGraphStage *hg_ = nullptr;
    void *h_stage_ = nullptr;
    size_t h_stage_cap_ = 0;
    void *ev0_ = nullptr, *ev1_ = nullptr, *ev2_ = nullptr;
    struct LinkSet { // pinned staging of one in-flight link sub-batch
        int *h_in = nullptr, *h_out = nullptr;
        size_t in_cap = 0, out_cap = 0;
        unsigned long long *h_ev = nullptr;
        void *ev_start = nullptr, *ev_stop = nullptr, *ev_done = nullptr;
        bool busy = false, timed = false;
        int ngroups = 0;
    } lset_[2];
    struct BgUpload {
        std::thread th;
        std::atomic<long long> resident{0}; // rows with id < resident have landed
        std::atomic<bool> failed{false}, active{false};
        std::string err;
        void *stream = nullptr, *pin[2] = {nullptr, nullptr}, *ev[2] = {nullptr, nullptr};
        size_t pin_bytes = 0;
    } bg_;
    StepBuffers *abi_sb_[4] = {nullptr, nullptr, nullptr, nullptr}; // context-owned step-buffer sets: 0/1 handed out by hnswdev_step_buffers, 2/3 private to dist_query_batch (so it never moves buffers a caller holds)
    int *d_guard_ = nullptr;                      // guard flag of pair_distance_kernel
    int *pair_dev_ = nullptr;                     // dist_pair_batch: [a | b | out] on the device
    size_t pair_dev_cap_ = 0;
    void *stream_ = nullptr;
    bool profiling_ = false;
    hnswdev_stats stats_{};
    bool shadows_allowed_ = true;
    int uj_len_ = 0, uj_entry_ = -1, uj_layer_ = -1; // s_jobs_ holds search_queries' jobs 0 .. uj_len_-1 for that entry point
    bool search_batch_impl(const SearchJob *jobs, int njobs, int k, int k_out, int *out_ids, float *out_d, int *out_flag, bool keep_repeat_flag,
                           bool two_heap, int u_entry, int u_layer);
};

} // namespace hnsw
