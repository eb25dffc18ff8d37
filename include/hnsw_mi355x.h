/*
 * hnsw_mi355x.h -- C ABI of the MI355X-native distance backend for HNSWIndex.Net's
 * Add / KnnQuery hot path.
 *
 * Two boundaries, one shared library (artifacts/native/linux-x64/HNSWIndex.Native.so):
 *
 *  (A) OUTER boundary -- the reference's own 16 cdecl exports, same names, argument order,
 *      C types, return codes and padding, so the reference's ctypes wrapper
 *      (bindings/bindings.py:45-119) and any C host bind unchanged.  Each prototype cites
 *      the [UnmanagedCallersOnly] export it replaces in
 *      /root/reference/bindings/HNSWIndex.Native/HNSWIndexExports.cs.
 *
 *  (B) INNER boundary -- the batched candidate-distance backend a C# (or any) host
 *      P/Invokes in place of the scalar-pair delegate
 *      `Func<float[],float[],float> distFnc` (src/HNSWIndex/HNSWIndex.cs:20) that
 *      GraphData.Distance invokes one pair at a time (src/HNSWIndex/GraphData.cs:255-277).
 *      The reference has no batched hook; this is the hook.  INTEGRATION.md shows the
 *      C# binding.
 *
 * Conventions: plain pointers and sizes only; inputs are borrowed for the duration of the
 * call; outputs are caller-allocated unless stated; no exception crosses the boundary.
 * Every distance is computed by hand-written HIP kernels on gfx950; there is no CPU
 * fallback -- with no HIP device the calls fail and say so.
 */
#ifndef HNSW_MI355X_H
#define HNSW_MI355X_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* =====================================================================================
 * (A) Outer boundary: the reference's 16 exports
 * ===================================================================================== */

/* HNSWIndexExports.cs:27-39  GetLastErrorUtf8.  Copies at most buf_len-1 bytes of the last
 * error (UTF-8) and NUL-terminates; returns the byte count needed (without the NUL). */
int hnsw_get_last_error_utf8(void *buf, int buf_len);

/* :41-65  Create.  metric: "sq_euclid" | "cosine" | "ucosine".  Consumes and resets the
 * process-global pending parameters set by hnsw_set_* (:16, :61).  Returns 0 on error. */
void *hnsw_create(const char *distance_metric);

/* :67-73  Free. handle == 0 is ignored. */
void hnsw_free(void *handle);

/* :75-100  Add.  vectors: count x dim row-major float32.  Writes the assigned ids to
 * out_ids[count]; returns the number written, 0 for a null handle / null vectors /
 * count <= 0 / dim <= 0, -1 on error.  The reference inserts with Parallel.For
 * (src/HNSWIndex/HNSWIndex.cs:70-78): a scheduler-dependent interleaving.  Here the batch
 * is inserted in id order with snapshot-batched searches (DESIGN.md "Add"); a call with
 * count == 1 -- the reference's own recipe for deterministic builds,
 * bindings/__tests__/parameters_test.py:65-68 -- is exactly the sequential
 * HNSWIndex.Add(item) (HNSWIndex.cs:55-65). */
int hnsw_add(void *handle, const float *vectors, int count, int dim, int *out_ids);

/* :102-117  Remove -> HNSWIndex.Remove(List<int>) (HNSWIndex.cs:83-102); the ids are removed in
 * the order given (the reference uses Parallel.For).  Returns 0; 0 for a null handle / null ids /
 * count <= 0; -1 on error (removals disabled: InvalidOperationException; unknown id). */
int hnsw_remove(void *handle, const int *ids, int count);

/* :119-149  KnnQuery -> BatchKnnQuery (HNSWIndex.cs:129-137).  out_ids / out_dists are
 * count x k row-major; rows with fewer than k results are padded with id -1, dist NaN
 * (:144).  Returns 0 on success (and for a null handle), -1 on error.
 * Threads (the reference: operations of one type may overlap on an index, README.md:64-65): calls on one handle
 * from several host threads run side by side on the GPU, each on a query lane of its own; a call of 32 768
 * queries or more has the handle to itself and uploads all but its first rows behind its launch; every other
 * export takes the handle exclusively.  Any mix of calls from any number of threads is safe. */
int hnsw_knn_query(void *handle, const float *vectors, int count, int dim, int k, int *out_ids, float *out_dists);

/* :151-197  RangeQuery -> BatchRangeQuery (HNSWIndex.cs:144-168).  For query i, out_ids[i] /
 * out_dists[i] receive callee-allocated arrays of counts[i] results ordered by distance (null when
 * counts[i] == 0); release them with hnsw_free_results.  Returns 0, or -1 on error. */
int hnsw_range_query(void *handle, const float *vectors, int count, int dim, float range, void **out_ids,
                     void **out_dists, int *counts);

/* :199-217  FreeRangeResults.  Frees (with free()) whatever hnsw_range_query allocated. */
void hnsw_free_results(void **ids_array, void **dists_array, int count);

/* :219-273  pending-parameter setters (defaults: src/HNSWIndex/HNSWParameters.cs:13-55).
 * They mutate a process-global parameter block that the NEXT hnsw_create consumes. */
int hnsw_set_collection_size(int collection_size);       /* :219 */
int hnsw_set_max_edges(int max_edges);                   /* :226 */
int hnsw_set_max_candidates(int max_candidates);         /* :233 */
int hnsw_set_remove_max_candidates(int max_candidates);  /* :240 */
int hnsw_set_distribution_rate(float dist_rate);         /* :247 */
int hnsw_set_random_seed(int seed);                      /* :254 */
int hnsw_set_min_nn(int min_nn);                         /* :261 */
int hnsw_set_allow_removals(bool allow_removals);        /* :268 */

/* ---- additions next to the reference's surface (not in the reference) ---------------- */

/* THE BACKEND'S KNOBS, all of them (version 1 = this layout; struct_size = sizeof(hnsw_mi355x_options) names the version, a
 * library that knows a longer struct fills the rest with defaults).  Pending like hnsw_set_*: consumed by the next hnsw_create /
 * hnsw_mi355x_deserialize and reset to the defaults afterwards.  The hnsw_mi355x_set_<knob>() functions below set one field each.
 * Nothing else configures the product: the HNSW_MI355X_* environment switches of rounds 1-4 are gone. */
typedef struct hnsw_mi355x_options {
    uint32_t struct_size;      /* sizeof(hnsw_mi355x_options) */
    int32_t device;            /* first HIP device ordinal (default 0) */
    int32_t devices;           /* device contexts hnsw_knn_query shards its queries over (default 1; see hnsw_mi355x_set_devices) */
    int32_t insert_batch;      /* hnsw_add's schedule: 0 = snapshot batches of at most the host's hardware threads (default: inside the
                                * reference's Parallel.For outcome set), 1 = one item after the other, B > 1 = that cap, -W = the sequential
                                * graph through exact windows (see hnsw_mi355x_set_insert_batch) */
    int32_t remove_batch;      /* hnsw_remove's schedule: 1 = sequential (default), B > 1 = disjoint neighbourhoods together */
    int32_t host_threads;      /* host worker threads, 0 = min(hardware threads, 16) */
    int32_t search_slots;      /* concurrent searches of the host lock-step driver (default 16384) */
    int32_t device_traversal;  /* 1 = graph traversal on the device (default), 0 = on the host, distances batched to the device */
    const char *diagnostics;   /* NULL (default), or test hooks "name=value,..." (csrc/diag.h; process-wide, replaces the environment
                                * variable HNSW_MI355X_DIAG while set): forces code paths for the test tiers, never needed by a caller */
} hnsw_mi355x_options;
/* Fills *out with the defaults (struct_size set).  0 / -1. */
int hnsw_mi355x_default_options(hnsw_mi355x_options *out);
/* Sets every pending knob from *opt (opt->struct_size must be set; fields beyond it keep their defaults).  0 / -1 on a bad value. */
int hnsw_mi355x_set_options(const hnsw_mi355x_options *opt);

/* Pending, like hnsw_set_*: HIP device ordinal for the next hnsw_create (default 0). */
int hnsw_mi355x_set_device(int device);
/* Pending: cap B on the snapshot batch of hnsw_add: B consecutive items search the graph as it stands, then link in id
 * order (a batch also never exceeds 1/16 of the linked graph -- 1/4 of it during the first min(65 536, final count / 16)
 * inserts).  That is an interleaving the reference's HNSWIndex.Add(List) -- a Parallel.For over the items,
 * src/HNSWIndex/HNSWIndex.cs:70-78 -- can produce iff B <= the threads it runs on, so:
 *   0 (default)  B = hnsw_mi355x_host_parallelism(), the hardware threads of this host: the graph stays inside the
 *                reference's outcome set on this machine;
 *   1            strictly sequential inserts, HNSWIndex.Add(item) per item (HNSWIndex.cs:55-65);
 *   B > 1        that cap, legal for a Parallel.For host with >= B threads; caps far beyond any host (the 65 536 of rounds
 *                1-4, ~10x the build rate) are this build's own schedule -- opt-in, checked only against its CPU restatement;
 *   -W (W >= 2)  the graph of strictly sequential inserts built through speculative windows: W consecutive items search
 *                one snapshot and record the adjacency lists they read; in item order, an item whose lists nobody has
 *                written since is linked, the first one that is not ends the round and searches again.  Same graph as
 *                max_batch = 1, bit for bit (DESIGN.md "exact window").
 * See DESIGN.md "Add". */
int hnsw_mi355x_set_insert_batch(int max_batch);
/* The same knob on an existing index (takes effect with the next hnsw_add): lets one index be continued under
 * another schedule, as bench.py's add_modes do. */
int hnsw_mi355x_index_set_insert_batch(void *handle, int max_batch);
/* The threads this process may run on (affinity mask) -- the default cap above -- and the cap an index is using. */
int hnsw_mi355x_host_parallelism(void);
int hnsw_mi355x_index_insert_batch(void *handle);
/* Counters of the exact-window schedule since the index was created: out[0] rounds (dependent search launches),
 * out[1] insert searches run (>= items: the re-searched ones count again), out[2] items inserted alone (entry-point
 * moves, hand-backs), out[3] items linked through windows. */
int hnsw_mi355x_exact_window_stats(void *handle, uint64_t out[4]);
/* Pending: hnsw_remove's schedule.  1 (default): the ids one after the other, exactly HNSWIndex.Remove(int) per id.
 * B > 1: the deterministic counterpart of Remove(List<int>) = Parallel.For under region locks (HNSWIndex.cs:95-101,
 * GraphLocker.cs:28-72): removals whose neighbourhoods (the node, its out- and in-neighbours on every layer) are
 * disjoint are taken together, up to B per batch out of the first 8 B remaining ids, all searching the graph as it
 * stands before the batch; the others wait, in order; the entry point is always removed alone.  With
 * hnsw_mi355x_set_device_traversal(0) removals stay sequential whatever B says.  See DESIGN.md 9. */
int hnsw_mi355x_set_remove_batch(int max_batch);
/* Pending: number of concurrent search slots of the lock-step driver (default 16384) and
 * host worker threads (default: min(hardware threads, 16)). */
int hnsw_mi355x_set_search_slots(int slots);
int hnsw_mi355x_set_host_threads(int threads);
/* Pending: 1 (default) = KnnQuery traverses on the device (graph mirrored in HBM, heaps in LDS);
 * 0 = traversal on the host, distances batched to the device step by step.  Same results. */
int hnsw_mi355x_set_device_traversal(int enabled);

/* Pending: number of device contexts of the next index (default 1).  With n > 1,
 * hnsw_knn_query -- BatchKnnQuery, a Parallel.For over independent searches, src/HNSWIndex/HNSWIndex.cs:129-137 via
 * bindings/HNSWIndex.Native/HNSWIndexExports.cs:119-149 -- shards its queries over n GPUs of the node inside this
 * one process: context g (device ordinal hnsw_mi355x_set_device + g, modulo the devices present) holds a replica of
 * the rows and of the graph mirror, copied device to device (hipMemcpyPeerAsync over xGMI) whenever the graph has
 * changed, answers queries [g nq / n, (g + 1) nq / n) and writes that slice of the caller's out arrays.  Same ids and
 * distance bits as one device.  Add / Remove / RangeQuery run on the first context. */
int hnsw_mi355x_set_devices(int n);
/* The sha256 (hex) of the sources this binary was compiled from -- csrc and include, by name and content, plus the compiler
 * flags -- or that string with "+variant" for a diagnostic build.  hnswindex.net_amd/build.py: source_id(). */
const char *hnsw_mi355x_build_id(void);
int hnsw_mi355x_device_count(void *handle);
/* hnswdev_stats of context `context` (0 = the primary). */
struct hnswdev_stats;
int hnsw_mi355x_get_stats_at(void *handle, int context, struct hnswdev_stats *out);

/* Measurement aid: hnsw_mi355x_set_queries uploads a query set (count x dim) once; every later
 * hnsw_mi355x_knn_query_resident(k) is hnsw_knn_query on that set with the inputs already in HBM
 * (out arrays: count x k). */
int hnsw_mi355x_set_queries(void *handle, const float *queries, int count, int dim);
int hnsw_mi355x_knn_query_resident(void *handle, int k, int *out_ids, float *out_dists);
/* Rows of the query set currently resident (hnsw_knn_query leaves its own queries resident, hnsw_range_query leaves
 * none): the out arrays of hnsw_mi355x_knn_query_resident must hold this many rows of k. */
int hnsw_mi355x_resident_count(void *handle);

/* Graph introspection for parity checks (reads host state only). */
int hnsw_mi355x_count(void *handle);   /* HNSWIndex.Count: live items */
int hnsw_mi355x_length(void *handle);  /* slots ever allocated (ids are < length) */
/* HNSWIndex.Ids(): the live ids in ActiveSet order; returns Count. */
int hnsw_mi355x_active_ids(void *handle, int *out, int cap);
int hnsw_mi355x_entry_point(void *handle);
/* Row length fixed by the first add (or by the loaded snapshot); 0 before that. */
int hnsw_mi355x_dim(void *handle);
int hnsw_mi355x_node_max_layer(void *handle, int id);
/* Copies up to cap out-edge ids of (id, layer); returns the edge count or -1. */
int hnsw_mi355x_get_out_edges(void *handle, int id, int layer, int *out, int cap);
uint64_t hnsw_mi355x_graph_hash(void *handle);

/* HNSWIndex.Serialize(filePath) / HNSWIndex.Deserialize(distFnc, filePath)
 * (src/HNSWIndex/HNSWIndex.cs:210-229): the reference's protobuf-net snapshot of
 * HNSWIndexSnapshot<float[],float> (HNSWIndexSnapshot.cs:12-16, GraphDataSnapshot.cs:13-35,
 * Node.cs:9-36, HNSWParameters.cs:12-55).  The reference's C ABI does not export these; its C#
 * API has them.  serialize: 0 / -1.  deserialize: a handle for hnsw_* calls, or 0 with the
 * message in hnsw_get_last_error_utf8; HNSW parameters come from the file, the pending
 * hnsw_mi355x_set_* backend knobs are consumed as by hnsw_create. */
int hnsw_mi355x_serialize(void *handle, const char *path_utf8);
void *hnsw_mi355x_deserialize(const char *distance_metric_utf8, const char *path_utf8);
/* Loading a graph that was built elsewhere -- by another replica of this index (multi-GPU: build once, broadcast,
 * import; hnswindex.net_amd/distributed.py::replicate_index) or by a host that owns one -- into a handle that holds
 * nothing yet.  hnsw_mi355x_import_nodes: rows (n x dim float32, id == row index), levels[i] = MaxLayer of node i
 * (Node.cs:27), the entry point (GraphData.EntryPointId); then one hnsw_mi355x_import_edges per layer 0..max(levels)
 * in the layout of hnsw_mi355x_export_edges (counts ignored where the node lacks the layer; ids in EdgeList order,
 * Node.cs:31-107).  Lists are validated (ids in range and on that layer, no duplicates, length <= MaxEdges(layer)).
 * Afterwards the index behaves as if it had inserted the n items itself: the level generator is advanced by n draws
 * (GraphData.cs:211-219), so later Adds continue exactly as on the index the graph came from.  0 / -1. */
int hnsw_mi355x_import_nodes(void *handle, const float *rows, int n, int dim, const int *levels, int entry_point);
int hnsw_mi355x_import_edges(void *handle, int layer, const int *counts, const int *edges, int stride);
/* Bulk forms: levels of nodes [0, min(count, cap)); returns count. */
int hnsw_mi355x_export_levels(void *handle, int *out, int cap);
/* counts[id] = out-degree of (id, layer), -1 where the node has no such layer;
 * edges[id*stride ..] = the ids, in adjacency order.  Returns count or -1. */
int hnsw_mi355x_export_edges(void *handle, int layer, int *counts, int *edges, int stride, int cap);

/* Counters of the index's device context (see hnswdev_stats). */
struct hnswdev_stats;
int hnsw_mi355x_get_stats(void *handle, struct hnswdev_stats *out);
int hnsw_mi355x_reset_stats(void *handle);
int hnsw_mi355x_set_profiling(void *handle, int enabled);

/* =====================================================================================
 * (B) Inner boundary: batched candidate-distance backend
 *     replaces GraphData.Distance(int, TVector) / Distance(int, int)
 *     (src/HNSWIndex/GraphData.cs:255-277) at the 14 call sites of SURVEY.md 8a.
 * ===================================================================================== */

/* 0-2: the reference's three float metrics (HNSWIndexExports.cs:47-60).  3: squared Euclidean distance on
 * int8-quantised rows with one float scale per row (BASELINE config 5; no reference counterpart) -- rows and
 * queries still cross the boundary as float32 and are quantised on the device, q = rint(x / scale), scale =
 * max|x| / 127; the distance is that of the dequantised vectors, computed from the exact int32 dot product
 * (metric name "sq_euclid_i8" for hnsw_create). */
enum { HNSWDEV_SQ_EUCLID = 0, HNSWDEV_COSINE = 1, HNSWDEV_UCOSINE = 2, HNSWDEV_SQ_EUCLID_I8 = 3 };

typedef struct hnswdev_stats {
    uint64_t launches;      /* distance-kernel launches */
    uint64_t evals;         /* distance evaluations (one candidate row read each) */
    uint64_t timed_launches;/* launches bracketed by HIP events (profiling on) */
    uint64_t timed_evals;   /* evaluations inside those launches */
    double kernel_ms;       /* sum of HIP-event durations of the timed launches */
    uint64_t row_bytes;     /* dim * sizeof(float): algorithmic bytes per evaluation */
    /* graph-resident search kernel (traversal on the device) */
    uint64_t search_launches;
    uint64_t search_evals;        /* distance evaluations inside those launches (device-counted) */
    uint64_t search_timed_launches;
    uint64_t search_timed_evals;
    double search_kernel_ms;      /* HIP-event durations of the timed search launches */
    uint64_t search_overflows;    /* traversals handed back to the lock-step path */
    uint64_t search_repeats;      /* traversals repeated on the device with the exact two-heap variant (equal distances) */
    /* Add's two halves, counted separately as well (they are also part of the search_* totals above):
     * graph_insert_search_kernel (descent + per-layer search + RelativeNeighborPruning) and the link half
     * (link_plan / link_offsets / link_order + graph_link_kernel: appends and PruneOverflow) */
    uint64_t insert_launches, insert_evals, insert_timed_launches, insert_timed_evals;
    double insert_kernel_ms;
    uint64_t link_launches, link_evals, link_timed_launches, link_timed_evals;
    double link_kernel_ms;
    uint64_t visited_hash_launches; /* traversal launches whose visited sets were per-wave hash tables (graphs above 4M nodes) */
    /* graph_range_kernel (RangeQuery on the device; also part of the search_* totals) */
    uint64_t range_launches, range_evals, range_timed_launches, range_timed_evals;
    double range_kernel_ms;
    uint64_t range_handbacks;       /* range traversals handed back (more results than a wave's list holds, visited table full) */
    uint64_t replica_bytes;         /* bytes this context received from another one (rows + graph mirror of a replica) */
    uint64_t tie_windows;           /* searches that met open candidates of equal distance and were shown to be order-free (no exact re-run) */
    uint64_t peer_direct_copies;    /* replica / query-set copies between contexts whose devices have peer access enabled (one device: counted here) */
    uint64_t peer_staged_copies;    /* ... and those the runtime had to stage through host memory (no peer access between the two devices) */
    uint64_t lat_launches;          /* traversal launches that ran the latency variant of their kernel (fewer jobs than its resident waves) */
    uint64_t range_device_ordered;  /* RangeQuery result lists (of two or more entries) whose ORDER the device completed: ranked, and replayed where distances tie */
    uint64_t range_host_ordered;    /* ... and those handed to the host for it (beyond 2 048 entries, a -0 distance, a replay the device gave up) */
    uint64_t insert_tie_reruns;     /* Add searches answered by the exact two-heap traversal because equal distances could show in what the insert
                                     * consumes in order (Span.Sort among equal keys, Heuristic.cs:22; heap layout at the far end of the list): the inserts
                                     * whose outcome rests on BCL tie behaviour this build restates from memory -- the "parity unpinned" exposure as a number
                                     * (also counted in search_repeats) */
    uint64_t lean_launches;         /* traversal launches that ran the lean form of their kernel (no visited sets: the default wherever lists hold <= 64 ids and rows <= 1 KB) */
} hnswdev_stats;

/* All return 0 on success, < 0 on error (message via hnswdev_ctx_last_error / hnswdev_last_error).
 * Calls on ONE context are serialised by the library (a mutex per context); different contexts
 * run concurrently. */

/* Creates a context on HIP device `device` holding up to `capacity` rows of `dim` float32
 * in one contiguous row-major HBM matrix (id == row index, GraphData.cs:95-115). */
int hnswdev_create(int device, int dim, int metric, long long capacity, void **ctx);
int hnswdev_destroy(void *ctx);
/* Grows the matrix (contents preserved); the counterpart of the doubling resize at
 * GraphData.cs:98-111. */
int hnswdev_reserve(void *ctx, long long capacity);
/* Copies rows [first_id, first_id+n) host -> HBM (and, for cosine, computes each row's
 * f32 squared norm in the reference's lane order and its double sqrt on the device). */
int hnswdev_upload_rows(void *ctx, int first_id, int n, const float *rows);
/* Reads rows back (parity checks). */
int hnswdev_download_rows(void *ctx, int first_id, int n, float *rows);

/* Uploads the query set of a batch of searches ONCE (nq x dim host floats; for cosine the norms are
 * computed on the device); records then name a query by its row index in this set.  Replaces the
 * previous resident set. */
int hnswdev_set_queries(void *ctx, const float *queries, int nq);

/* ---- the batched step, asynchronous and double-buffered ---------------------------------------
 * This is what replaces the scalar delegate inside the traversal loops (GraphNavigator.cs:70,
 * :163, :231; Heuristic.cs:34; GraphConnector.cs:233): the host advances MANY traversals together;
 * each step every live traversal states one record -- which vector (a resident query, or a stored
 * row) against which candidate rows -- and ONE launch evaluates all of them.
 *
 * The context owns two buffer sets (set = 0 | 1) in pinned host memory with an HBM mirror; they
 * are (re)allocated only when nslots grows or stride changes, never per step, and no other call
 * moves them: the pointers stay valid until the next hnswdev_step_buffers for that set that asks
 * for more (or hnswdev_destroy).  Layout of set `set`:
 *     rec [s * (stride + 2) + 0]      = cnt   number of candidate ids of slot s (0: idle slot)
 *     rec [s * (stride + 2) + 1]      = qidx  >= 0: resident query index;  < 0: ~row_id (id<->id)
 *     rec [s * (stride + 2) + 2 ...]  = ids   candidate row ids, cnt <= stride
 *     dist[s * stride + c]            = metric(row[ids[c]], that vector) after hnswdev_step_wait
 * hnswdev_step_submit enqueues ONE host->HBM copy of the used records, ONE kernel and ONE copy of
 * the distances back, on the context's stream, and returns; hnswdev_step_wait blocks until that
 * set's distances have landed.  While one set is in flight the host consumes / fills the other.
 * A record naming a row or query that was never uploaded is not dereferenced: its distances come
 * back NaN and hnswdev_step_wait returns -1. */
int hnswdev_step_buffers(void *ctx, int set, int nslots, int stride, int **rec, float **dist);
int hnswdev_step_submit(void *ctx, int set, int nslots_used);
int hnswdev_step_wait(void *ctx, int set);

/* Synchronous conveniences over the same two sets (nothing is allocated per call once they exist):
 * Distance(int a, TVector b) for many (query, candidate-list) pairs at once:
 * out[j] = metric(row[cand_ids[j]], queries[i]) for cand_offsets[i] <= j < cand_offsets[i+1].
 * queries: nq x dim host floats, uploaded as the resident set -- or NULL to use the set already
 * resident (hnswdev_set_queries), so that a batch of searches uploads its queries once;
 * cand_offsets: nq+1 ints. */
int hnswdev_dist_query_batch(void *ctx, const float *queries, int nq, const int *cand_offsets, const int *cand_ids,
                             float *out);
/* Distance(int a, int b): out[j] = metric(row[a_ids[j]], row[b_ids[j]]); synchronous. */
int hnswdev_dist_pair_batch(void *ctx, const int *a_ids, const int *b_ids, int n, float *out);

/* ---- graph-resident traversal: SearchLayerQuery + FindEntryPointQuery
 *      (src/HNSWIndex/GraphNavigator.cs:39-82,194-256) for a batch of queries in one launch ---- */

/* Describes the host graph to the context: n nodes, MaxEdges = max_edges, levels[i] = MaxLayer of
 * node i (Node.cs:27).  Then one hnswdev_graph_set_layer per layer 0..max(levels), then commit. */
int hnswdev_graph_begin(void *ctx, int n, int max_edges, const int *levels);
/* counts[i] = OutEdges[layer].Count of node i (ignored where levels[i] < layer);
 * edges[i*stride .. i*stride+counts[i]) = its ids in EdgeList order (Node.cs:31-107). */
int hnswdev_graph_set_layer(void *ctx, int layer, const int *counts, const int *edges, int stride);
/* Uploads the staged graph to HBM (replaces the previous one). */
int hnswdev_graph_commit(void *ctx);
/* KnnQuery for nq queries (nq x dim floats) from entry point `entry_point` (GraphData.EntryPoint):
 * beam width k_beam = max(MinNN, k) (HNSWIndex.cs:115), first k_out results of the stable distance
 * order (HNSWIndex.cs:121) into out_ids / out_dists (nq x k_out, padded with -1 / NaN).
 * out_flags[i] = 1: query i met a case the device path hands back (candidate heap beyond its
 * capacity, NaN or -0 distance) -- evaluate it with hnswdev_dist_query_batch instead. */
int hnswdev_knn_search(void *ctx, const float *queries, int nq, int entry_point, int k_beam, int k_out, int *out_ids,
                       float *out_dists, int *out_flags);

/* RangeQuery for nq queries from `entry_point`: FindEntryPointQuery + SearchLayerRange at layer 0
 * (HNSWIndex.cs:144-156, GraphNavigator.cs:262-325), no filter.  out_counts[i] = results of query i, kept in the
 * context until the next range search and copied out by hnswdev_range_results: concatenated in query order, each
 * query's results in RangeQuery's order -- ascending by distance, results of EQUAL distance in the order of the
 * reference's heap array (replayed on the committed graph) -- out_ids / out_dists hold sum(out_counts) entries.
 * out_flags[i] = 1: handed back (count 0; on graphs above 4M nodes, a traversal that outgrows its visited table):
 * evaluate it with hnswdev_dist_query_batch instead. */
int hnswdev_range_search(void *ctx, const float *queries, int nq, int entry_point, float range, int *out_counts, int *out_flags);
int hnswdev_range_results(void *ctx, int *out_ids, float *out_dists);

int hnswdev_sync(void *ctx);
int hnswdev_set_profiling(void *ctx, int enabled);
int hnswdev_get_stats(void *ctx, hnswdev_stats *out);
int hnswdev_reset_stats(void *ctx);
/* Last error, process-wide (creation failures have no context yet) ... */
int hnswdev_last_error(char *buf, int buf_len);
/* ... and of one context: calls on different contexts never overwrite each other's message.
 * Both copy at most buf_len-1 bytes, NUL-terminate and return the byte count needed. */
int hnswdev_ctx_last_error(void *ctx, char *buf, int buf_len);
/* Number of visible HIP devices (>= 0), or < 0 on error. */
int hnswdev_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* HNSW_MI355X_H */
