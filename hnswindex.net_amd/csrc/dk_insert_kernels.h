// dk_insert_kernels.h -- device code, part of device_kernels.h: Add, search half: insert_job, graph_insert_search_kernel.
#pragma once
#include "dk_sorted_top.h"
#include "dk_pool_top.h"
#include "dk_traverse_exact.h"
#include "dk_heuristic.h"

namespace hnsw {

// Insert, search half, fused: for one new item, GraphConnector.AddNewConnections' whole loop
// (GraphConnector.cs:172-181): FindEntryPoint, then for every layer of the item ConnectAtLayer's
// SearchLayer + RelativeNeighborPruning (:189-190) with the next layer's entry = selected[0]
// (:216).  One launch serves every layer of every item of a batch (the few multi-layer items
// clear their visited bitset between layers).  Output per (job, layer): the selected ids in
// selection order (layer 0 -> slot `job`; layer L >= 1 -> upper slot jobs[].aux + L - 1).
// jobs[].search_layer = the item's first layer min(level, top).
// graph_insert_search_kernel's parameter list as a struct (same order, same types: the kernarg segment's layout) -- see kernarg_load
struct InsertKernArgs {
    const float *rows; const double *row_sn; int dim; const int *adj0; int stride0; const int64_t *upper; const int *pool; int strideU;
    const SearchJob *jobs; int k, cand_cap; ND *spill; int spill_cap, max_edges0; unsigned *visited; long long vis_words; int *vis_tab; int vis_tab_cap;
    int *out_sel0, *out_cnt0, *out_selU, *out_cntU; int sel_stride; int *out_flag; unsigned long long *eval_counter; int nbcap, njobs; int *job_counter;
    int overlap; const int *order; int *read_log; int read_log_cap;
};
static_assert(offsetof(InsertKernArgs, jobs) == 64 && offsetof(InsertKernArgs, out_sel0) == 128 && offsetof(InsertKernArgs, read_log) == 216 && sizeof(InsertKernArgs) == 232,
              "InsertKernArgs must mirror graph_insert_search_kernel's parameter list");
#define HNSW_KAI(field) kernarg_load<decltype(InsertKernArgs::field)>((unsigned)offsetof(InsertKernArgs, field))
// The latency variant reads the arguments that only the start / the end of a job or of a layer use where they are used (kernarg_load) instead of
// carrying them through the traversal: its logic wave is bound by its own instruction stream, and spilled SGPRs are instructions in it.
// Measured (profiles/r5_lean_ab.log, 7): 1M build under the default cap 106.3 -> 110.1 k adds/s, B = 1 1 007 -> 1 031 adds/s, ladder +3-4 %; the plain
// form (large snapshots) keeps its arguments in registers: 0.640 -> 0.630 of peak with the reloads, inside the noise on the wrong side.

template <int METRIC, int NS, bool HASHED, int FORM = kFormPlain>
__device__ __forceinline__ void insert_job(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                           const int *__restrict__ adj0, int stride0, const int64_t *__restrict__ upper,
                           const int *__restrict__ pool, int strideU, const SearchJob *__restrict__ jobs, int k,
                           int cand_cap, ND *__restrict__ spill, int spill_cap, int max_edges0, VisitedSet<HASHED> &V,
                           int *__restrict__ out_sel0, int *__restrict__ out_cnt0, int *__restrict__ out_selU,
                           int *__restrict__ out_cntU, int sel_stride, int *__restrict__ out_flag,
                           unsigned long long *__restrict__ eval_counter, int nbcap, unsigned char *smem, int job, int overlap_and_flags,
                           int *__restrict__ read_log, int read_log_cap, TeamPort *port = nullptr, bool *v_dirty_out = nullptr)
{
    constexpr bool LAT = FORM == kFormLat, LEAN = FORM == kFormLean;
    constexpr bool KA = LAT;
    if constexpr (KA) { jobs = HNSW_KAI(jobs); cand_cap = HNSW_KAI(cand_cap); nbcap = HNSW_KAI(nbcap); read_log = HNSW_KAI(read_log); read_log_cap = HNSW_KAI(read_log_cap); }
    bool v_dirty = false; // the visited set has marks in it (the sorted traversal without a visited set -- oflags bit 3 -- leaves none)
    const bool novis = LEAN || (!LAT && (overlap_and_flags & 8) != 0);
    const SearchLds L = carve_lds(smem, k, cand_cap, dim, nbcap);
    const int lane = threadIdx.x & 63;
    const int overlap = overlap_and_flags & 1; // bit 0: overlapped form (bit 1: the MFMA-prefiltered heuristic is allowed)
    SearchJob jb = jobs[job];
    const GraphView G{adj0, stride0, upper, pool, strideU};
    const int item = ~jb.qref;
    const float *q = rows + (size_t)item * dim;
    double sb = 0.0;
    if (METRIC == M_COS) sb = row_sn[item];
    for (int i = lane; i < dim; i += 64) L.qs[i] = q[i];
    if constexpr (LAT) { if (lane == 0) port->m->sb = sb; }
    unsigned long long evals = 0;
    bool ok = true, repeat = false;
    // exact-window Add: record [n, entries...] of this job's read log (n beyond the capacity = overflow)
    ReadLog RL{read_log ? read_log + (size_t)job * read_log_cap + 2 : nullptr, 0, (read_log_cap - 2) / 2};
#ifdef EXP_PHASE_CLOCKS
    const long long ph_j0 = __builtin_readcyclecounter();
#endif
    const int first_layer = jb.search_layer, last_layer = jb.stop_layer;
    for (int layer = first_layer; layer >= last_layer && ok; --layer) {
        if (layer != first_layer && v_dirty) { V.clear(lane); v_dirty = false; } // a fresh SearchLayer: new visited list (VisitedListPool.cs:74-106)
        int top_n = 0;
        const int rl_n0 = RL.n;
        if constexpr (KA) max_edges0 = HNSW_KAI(max_edges0);
        const int max_edges = layer == 0 ? max_edges0 : (max_edges0 >> 1); // GraphData.MaxEdges :247-250
        const bool exact_only = (overlap_and_flags & 0x200) != 0; // the launch runs the exact two-heap traversal only (beams beyond 512 entries)
        bool exact = exact_only, order_tie = false;
        const unsigned long long ev0 = evals;
        if (!exact_only) {
            bool tie = false;
            if constexpr (LAT) ok = traverse_pool<METRIC, NS, HASHED>(rows, row_sn, dim, sb, G, jb, k, k, V, L, lane, top_n, tie, evals, RL, &order_tie, nullptr, port);
            else ok = traverse_sorted<METRIC, NS, HASHED, LEAN>(rows, row_sn, dim, sb, G, jb, k, k, V, L, lane, top_n, tie, evals, overlap_and_flags & 9, RL, &order_tie); // Span.Sort consumes all
            v_dirty = v_dirty || !novis;
            if (!ok) break;
            // equal distances where the heap layout shows, or fewer candidates than MaxEdges (the heuristic
            // then returns them in HEAP order, Heuristic.cs:13-18): this layer again, exact traversal
            exact = tie || top_n < max_edges;
        }
        int rc = 0;
        for (;;) {
            if (exact) {
                if (!exact_only) {
                    repeat = true;
                    evals = ev0;
                    top_n = 0;
                    RL.n = rl_n0; // the same lists are read again
                    if (v_dirty) V.clear(lane);
                }
                if constexpr (KA) { cand_cap = HNSW_KAI(cand_cap); spill_cap = HNSW_KAI(spill_cap); spill = HNSW_KAI(spill) + (size_t)blockIdx.x * spill_cap; }
                ok = traverse<METRIC, HASHED>(rows, row_sn, dim, sb, G, jb, k, cand_cap, spill, spill_cap, V, L, lane, top_n, evals, RL, nullptr, nullptr,
                                              LAT || overlap != 0);
                v_dirty = true;
                if (!ok) break;
            }
#ifdef EXP_PHASE_CLOCKS
            const long long ph_h0 = __builtin_readcyclecounter();
#endif
            // the candidate heap's LDS area is idle now: the grouped heuristic borrows it
            if constexpr (KA) cand_cap = HNSW_KAI(cand_cap);
            rc = relative_neighbor_pruning<METRIC, NS == 8>(rows, row_sn, dim, L.top, top_n, max_edges, L, lane, evals, !exact,
#ifdef HNSW_NO_GROUPED
                                                            nullptr, 0);
#else
                                                            reinterpret_cast<float *>(L.cand), sizeof(ND) * (size_t)cand_cap, (overlap_and_flags & 2) != 0);
#endif
#ifdef EXP_PHASE_CLOCKS
            if (lane == 0) atomicAdd(&g_phase[8], (unsigned long long)(__builtin_readcyclecounter() - ph_h0)); // heuristic cycles
#endif
            if (exact || !order_tie) break;
            // Equal distances somewhere in the ascending candidate list, and nothing else open: the SET is the
            // reference's, but Span.Sort (Heuristic.cs:22) leaves such a group in an order only the heap array knows.
            // The greedy pass (:23-40) shows that order only if two members of a group get past the ids accepted before
            // the group (one may then turn the other away, or both enter the list in that order).  A member that was NOT
            // accepted just now, with no member of its group accepted before it, was turned away by ids of smaller
            // distance -- in any order.  So when every member but the last of each group was rejected, the outcome is
            // the reference's whatever its order was (one candidate in seven is accepted on uniform data: most groups
            // are harmless -- 2.1 % of the inserts at C2 used to start over, a third of a percent still do).
            wave_sync();
            bool shows = false;
            for (int p0 = 0; p0 < top_n; p0 += 64) {
                const int pp = p0 + lane;
                if (pp >= 1 && pp < top_n && __float_as_uint(L.top[pp].dist) == __float_as_uint(L.top[pp - 1].dist)) {
                    const int first = L.top[pp - 1].id;
                    for (int a = 0; a < rc; ++a) shows = shows || L.acc[a] == first;
                }
            }
            if (__ballot(shows) == 0ull) break;
            exact = true;
        }
        if (!ok) break;
        if constexpr (KA) { out_sel0 = HNSW_KAI(out_sel0); out_cnt0 = HNSW_KAI(out_cnt0); out_selU = HNSW_KAI(out_selU); out_cntU = HNSW_KAI(out_cntU); sel_stride = HNSW_KAI(sel_stride); }
        int *osel = layer == 0 ? out_sel0 + (size_t)job * sel_stride : out_selU + (size_t)(jb.aux + layer - 1) * sel_stride;
        for (int i = lane; i < rc; i += 64) osel[i] = L.acc[i];
        if (lane == 0) { if (layer == 0) out_cnt0[job] = rc; else out_cntU[jb.aux + layer - 1] = rc; }
        const int next_entry = __builtin_amdgcn_readfirstlane(L.acc[0]); // :216 selected[0] -> bestPeer of the next layer (:179)
        jb.entry = next_entry;
        jb.entry_layer = layer - 1;
        jb.search_layer = layer - 1;
        wave_sync();
    }
    if (v_dirty_out) *v_dirty_out = v_dirty || LAT; // (the latency variants' memory wave marks as it goes)
    if constexpr (KA) { out_flag = HNSW_KAI(out_flag); eval_counter = HNSW_KAI(eval_counter); read_log = HNSW_KAI(read_log); read_log_cap = HNSW_KAI(read_log_cap); }
    if (lane == 0) {
        out_flag[job] = ok ? (repeat ? 2 : 0) : 1; // 2: informational (a layer was answered by the exact traversal)
        if (read_log) read_log[(size_t)job * read_log_cap] = RL.n;
        atomicAdd(eval_counter, evals);
#ifdef EXP_PHASE_CLOCKS
        atomicAdd(&g_phase[9], (unsigned long long)(__builtin_readcyclecounter() - ph_j0)); // whole insert job
#endif
    }
}

template <int METRIC, int NS, bool HASHED, int FORM = kFormPlain>
__global__ void __launch_bounds__(FORM == kFormLat ? 128 : 64) __attribute__((amdgpu_waves_per_eu(HNSW_WAVES(FORM == kFormLat ? (NS <= 4 ? 2 : 1) : NS <= 4 ? 3 : 2)))) // up to 256 candidates: 168 VGPRs, three waves per SIMD (LAT: see graph_search_kernel)
graph_insert_search_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                           const int *__restrict__ adj0, int stride0, const int64_t *__restrict__ upper,
                           const int *__restrict__ pool, int strideU, const SearchJob *__restrict__ jobs, int k,
                           int cand_cap, ND *__restrict__ spill, int spill_cap, int max_edges0, unsigned *__restrict__ visited, long long vis_words,
                           int *__restrict__ vis_tab, int vis_tab_cap, int *__restrict__ out_sel0, int *__restrict__ out_cnt0, int *__restrict__ out_selU,
                           int *__restrict__ out_cntU, int sel_stride, int *__restrict__ out_flag,
                           unsigned long long *__restrict__ eval_counter, int nbcap, int njobs, int *__restrict__ job_counter, int overlap,
                           const int *__restrict__ order, int *__restrict__ read_log, int read_log_cap)
{
    constexpr bool LAT = FORM == kFormLat;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    VisitedSet<HASHED> V{visited + (size_t)blockIdx.x * (size_t)vis_words, vis_words,
                 vis_tab ? vis_tab + (size_t)blockIdx.x * (size_t)vis_tab_cap : nullptr, (unsigned)(vis_tab_cap - 1), 0, vis_tab_cap / 4 * 3};
    ND *my_spill = spill + (size_t)blockIdx.x * spill_cap;

    TeamPort port{nullptr, 0, 0};
    if constexpr (LAT) {
        // two waves per block (see TeamMail): wave 1 serves the expansions, wave 0 is the traversal.  The mailbox follows
        // the traversal's LDS; its sequence words are zeroed before the roles part (the one barrier both waves meet at).
        TeamMail *mail = reinterpret_cast<TeamMail *>(smem + ((search_lds_bytes(k, cand_cap, dim, true, nbcap) + 15) & ~(size_t)15));
        if (threadIdx.x == 0) { mail->req_seq = 0; mail->rsp_seq = 0; mail->hint_node = -1; }
        __syncthreads();
        if (threadIdx.x >= 64) {
            const GraphView G{adj0, stride0, upper, pool, strideU};
            const SearchLds L = carve_lds(smem, k, cand_cap, dim, nbcap);
            memory_wave<METRIC, HASHED>(rows, row_sn, dim, G, V, L.qs, mail, lane);
            return;
        }
        port.m = mail;
    }
    bool v_dirty = true;
    for (;;) { // persistent, see graph_search_kernel
        if constexpr (LAT) { job_counter = HNSW_KAI(job_counter); njobs = HNSW_KAI(njobs); order = HNSW_KAI(order); }
        int job = 0;
        if (lane == 0) job = atomicAdd(job_counter, 1);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= njobs) break;
        // queue position -> batch item: the items that search several layers are taken first (they run the
        // longest; started last they would be the tail of the launch).  Results are filed by item, so the
        // order of processing changes nothing else.
        if (order) job = __builtin_amdgcn_readfirstlane(order[job]);
        insert_job<METRIC, NS, HASHED, FORM>(rows, row_sn, dim, adj0, stride0, upper, pool, strideU, jobs, k, cand_cap, my_spill, spill_cap, max_edges0, V,
                               out_sel0, out_cnt0, out_selU, out_cntU, sel_stride, out_flag, eval_counter, nbcap, smem, job, overlap, read_log, read_log_cap, &port, &v_dirty);
        if (v_dirty) V.clear(lane);
    }

    if constexpr (LAT) port.post(-1, 0, lane); // the memory wave leaves
}

} // namespace hnsw
