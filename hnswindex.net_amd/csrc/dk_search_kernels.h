// dk_search_kernels.h -- device code, part of device_kernels.h: KnnQuery / RangeQuery: search_job, graph_search_kernel (persistent, shadows, gated upload), graph_range_kernel.
#pragma once
#include "dk_sorted_top.h"
#include "dk_pool_top.h"
#include "dk_traverse_exact.h"

namespace hnsw {

// Sorted-list traversal with NS register sets (k <= 64 * NS); a wave that meets equal distances where the heap layout
// shows starts over with the exact two-heap traversal (out_flag 2, informational).  Launch flag 0x200 (beams beyond 512
// entries, HNSW_MI355X_SORTED_TOP=0): the two-heap traversal only.
// One job on this wave.  `vis` / `spill`: the wave's own scratch (vis all zero on entry; the caller
// clears it afterwards).
// Job words of a launch with SHADOW traversals (graph_search_kernel): bit 0 answered (results written), bit 1 the
// wave that owns the job met a tie, bit 2 a shadow traversal has been started for it.
constexpr int kJobAnswered = 1, kJobTied = 2, kJobShadowed = 4;

// graph_search_kernel's parameter list as a struct (same order, same types: the kernarg segment's layout) -- see kernarg_load
struct SearchKernArgs {
    const float *rows; const double *row_sn; const float *queries; const double *q_sn; int dim; const int *adj0; int stride0;
    const int64_t *upper; const int *pool; int strideU; const SearchJob *jobs; int k, cand_cap; ND *spill; int spill_cap;
    unsigned *visited; long long vis_words; int *vis_tab; int vis_tab_cap, k_out; int *out_ids; float *out_d; int *out_cnt, *out_flag;
    unsigned long long *eval_counter; int nbcap, njobs; int *job_counter; int overlap; const int *ready;
};
static_assert(offsetof(SearchKernArgs, jobs) == 80 && offsetof(SearchKernArgs, out_ids) == 144 && offsetof(SearchKernArgs, ready) == 208 && sizeof(SearchKernArgs) == 216,
              "SearchKernArgs must mirror graph_search_kernel's parameter list (the explicit arguments start the kernarg segment, each at its natural alignment)");
#define HNSW_KA(field) kernarg_load<decltype(SearchKernArgs::field)>((unsigned)offsetof(SearchKernArgs, field))

template <int METRIC, int NS, bool HASHED, int FORM = kFormPlain>
__device__ __forceinline__ void search_job(const float *__restrict__ rows, const double *__restrict__ row_sn, const float *__restrict__ queries,
                    const double *__restrict__ q_sn, int dim, const int *__restrict__ adj0, int stride0,
                    const int64_t *__restrict__ upper, const int *__restrict__ pool, int strideU,
                    const SearchJob *__restrict__ jobs, int k, int cand_cap, ND *__restrict__ spill,
                    int spill_cap, VisitedSet<HASHED> &V, int k_out, int *__restrict__ out_ids,
                    float *__restrict__ out_d, int *__restrict__ out_cnt, int *__restrict__ out_flag,
                    unsigned long long *__restrict__ eval_counter, int nbcap, unsigned char *smem, int job, int overlap,
                    int *__restrict__ job_word = nullptr, bool shadow = false, TeamPort *port = nullptr, bool *v_untouched = nullptr)
{
    constexpr bool LAT = FORM == kFormLat, LEAN = FORM == kFormLean;
    if constexpr (LEAN) { // (what only the start and the end of a job need is read where it is needed: kernarg_load)
        cand_cap = HNSW_KA(cand_cap); nbcap = HNSW_KA(nbcap); jobs = HNSW_KA(jobs); queries = HNSW_KA(queries); q_sn = HNSW_KA(q_sn);
    }
    const SearchLds L = carve_lds(smem, k, cand_cap, dim, nbcap);
    const int lane = threadIdx.x & 63;
    if (v_untouched) *v_untouched = false;
    const SearchJob jb = jobs[job];
    const GraphView G{adj0, stride0, upper, pool, strideU};

    const float *q;
    double sb = 0.0;
    if (jb.qref >= 0) {
        q = queries + (size_t)jb.qref * dim;
        if (METRIC == M_COS) sb = q_sn[jb.qref];
    } else {
        q = rows + (size_t)(~jb.qref) * dim;
        if (METRIC == M_COS) sb = row_sn[~jb.qref];
    }
    for (int i = lane; i < dim; i += 64) L.qs[i] = q[i];
    if constexpr (LAT) { if (lane == 0) port->m->sb = sb; } // (the memory wave reads both after the first request's release)
    unsigned long long evals = 0;
    int top_n = 0;
    bool repeated = shadow;
    ReadLog RL{nullptr, 0, 0};
    // With shadows, whoever sets kJobAnswered first writes the job's results (both traversals compute the same ones).
    auto claim_answer = [&]() -> bool {
        if (!job_word) return true;
        int old = 0;
        if (lane == 0) old = atomicOr(job_word, kJobAnswered);
        return (__builtin_amdgcn_readfirstlane(old) & kJobAnswered) == 0;
    };
    if constexpr (NS > 0) {
        if (jb.aux != -2 && !shadow && !(overlap & 0x200)) { // (0x200: this launch runs the exact two-heap traversal only)
        bool tie = false;
        // OrderBy + Take(k_out) reads k_out entries in order and decides between entries k_out - 1 and k_out
        bool window = false;
        bool ok1;
        if constexpr (LAT) ok1 = traverse_pool<METRIC, NS, HASHED>(rows, row_sn, dim, sb, G, jb, k, k_out + 1, V, L, lane, top_n, tie, evals, RL, nullptr, &window, port);
        else ok1 = traverse_sorted<METRIC, NS, HASHED, LEAN>(rows, row_sn, dim, sb, G, jb, k, k_out + 1, V, L, lane, top_n, tie, evals, overlap, RL, nullptr, &window);
        if (!(ok1 && tie)) {
            if (v_untouched) *v_untouched = LEAN || (!LAT && (overlap & 8) != 0); // the sorted traversal ran without a visited set: nothing to clear
            if (!claim_answer()) return;
            if constexpr (LEAN) {
                k_out = HNSW_KA(k_out); out_ids = HNSW_KA(out_ids); out_d = HNSW_KA(out_d); out_cnt = HNSW_KA(out_cnt); out_flag = HNSW_KA(out_flag);
                eval_counter = HNSW_KA(eval_counter);
            }
            // KnnQuery's tail (HNSWIndex.cs:119-123): OrderBy(Dist).Take(k) of distinct distances is the
            // head of the ascending list; missing results are padded (HNSWIndexExports.cs:144)
            for (int r = lane; r < k_out; r += 64) {
                const bool have = r < top_n;
                out_ids[(size_t)job * k_out + r] = have ? L.top[r].id : -1;
                out_d[(size_t)job * k_out + r] = have ? L.top[r].dist : __uint_as_float(0x7fc00000u);
            }
            if (lane == 0) {
                out_cnt[job] = ok1 ? top_n : 0;
                out_flag[job] = ok1 ? (window ? 4 : 0) : 1; // 4: informational (a group window of equal distances closed cleanly)
                atomicAdd(eval_counter, evals);
            }
            return;
        }
        // equal distances where the heap layout shows: the exact traversal answers this job -- the shadow that an
        // idle wave has already started for it (see graph_search_kernel), or this wave, starting over
        if (job_word) {
            int old = 0;
            if (lane == 0) old = atomicOr(job_word, kJobTied);
            if (__builtin_amdgcn_readfirstlane(old) & (kJobShadowed | kJobAnswered)) return;
        }
        if constexpr (!LEAN) V.clear(lane); // (the lean form's sorted traversal has no visited set)
        evals = 0;
        top_n = 0;
        repeated = true;
        }
    }
    bool aborted = false;
    bool ok;
    if constexpr (LEAN) {
        // the lean kernel carries neither its visited set nor its spill area from job to job: both are looked up here, for the one job
        // in several hundred that takes this path, and the set is left clean again
        cand_cap = HNSW_KA(cand_cap);
        const int spill_cap_ = HNSW_KA(spill_cap);
        const long long vis_words = HNSW_KA(vis_words);
        int *const vis_tab = HNSW_KA(vis_tab);
        const int vis_tab_cap = HNSW_KA(vis_tab_cap);
        VisitedSet<HASHED> VE{HNSW_KA(visited) + (size_t)blockIdx.x * (size_t)vis_words, vis_words,
                              vis_tab ? vis_tab + (size_t)blockIdx.x * (size_t)vis_tab_cap : nullptr, (unsigned)(vis_tab_cap - 1), 0, vis_tab_cap / 4 * 3};
        ok = traverse<METRIC, HASHED>(rows, row_sn, dim, sb, G, jb, k, cand_cap, HNSW_KA(spill) + (size_t)blockIdx.x * spill_cap_, spill_cap_, VE, L, lane, top_n, evals, RL,
                                      shadow ? job_word : nullptr, &aborted, true);
        VE.clear(lane);
        if (v_untouched) *v_untouched = true;
        if (!aborted) {
            k_out = HNSW_KA(k_out); out_ids = HNSW_KA(out_ids); out_d = HNSW_KA(out_d); out_cnt = HNSW_KA(out_cnt); out_flag = HNSW_KA(out_flag);
            eval_counter = HNSW_KA(eval_counter);
        }
    } else
    ok = traverse<METRIC, HASHED>(rows, row_sn, dim, sb, G, jb, k, cand_cap, spill, spill_cap, V, L, lane, top_n, evals, RL,
                                             shadow ? job_word : nullptr, &aborted, LAT || (overlap & 1) != 0 || repeated);
    if (aborted || !claim_answer()) return;
    if (jb.aux == -2) { // SearchLayer's own return value: topCandidates.ToArray(), the heap's array (BinaryHeap.cs:41-44)
        wave_sync();
        for (int r = lane; r < k_out; r += 64) {
            const bool have = ok && r < top_n;
            out_ids[(size_t)job * k_out + r] = have ? L.top[r].id : -1;
            out_d[(size_t)job * k_out + r] = have ? L.top[r].dist : __uint_as_float(0x7fc00000u);
        }
        if (lane == 0) {
            out_cnt[job] = ok ? top_n : 0;
            out_flag[job] = ok ? 0 : 1;
            atomicAdd(eval_counter, evals);
        }
        return;
    }
    // KnnQuery's tail (HNSWIndex.cs:119-123): OrderBy(c => c.Dist) is a STABLE sort over the heap
    // array (ToArray(), BinaryHeap.cs:41-44) and only the first k_out survive -- so select the
    // k_out smallest (float.CompareTo order: NaN first, -0 == +0) with ties broken by array index:
    // exactly the stable sort's prefix.  Key = (order-preserving bits << 32) | index, wave min.
    wave_sync();
    unsigned long long used = 0; // bit t: entry lane + 64*t already emitted
    for (int r = 0; r < k_out; ++r) {
        unsigned long long best = ~0ull;
        for (int t = 0, i = lane; i < top_n; ++t, i += 64) {
            if ((used >> t) & 1ull) continue;
            float d = L.top[i].dist;
            unsigned u;
            if (d != d) u = 0u;                      // NaN sorts first
            else {
                if (d == 0.0f) d = 0.0f;             // -0 and +0 compare equal
                u = __float_as_uint(d);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                if (u == 0u) u = 1u;                 // keep NaN's key unique (only -NaN-like bit patterns reach 0)
            }
            unsigned long long key = ((unsigned long long)u << 32) | (unsigned)i;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            unsigned long long o = __shfl_xor(best, off, 64);
            best = o < best ? o : best;
        }
        if (best == ~0ull) { // fewer than k_out results: pad (HNSWIndexExports.cs:144)
            if (lane == 0) { out_ids[(size_t)job * k_out + r] = -1; out_d[(size_t)job * k_out + r] = __uint_as_float(0x7fc00000u); }
            continue;
        }
        const int wi = (int)(best & 0xffffffffu);
        if ((wi & 63) == lane) used |= 1ull << (wi >> 6);
        if (lane == 0) { ND w = L.top[wi]; out_ids[(size_t)job * k_out + r] = w.id; out_d[(size_t)job * k_out + r] = w.dist; }
    }
    if (lane == 0) {
        out_cnt[job] = ok ? top_n : 0;
        out_flag[job] = ok ? (repeated ? 2 : 0) : 1; // 2: informational (answered by the exact traversal)
        atomicAdd(eval_counter, evals);
    }
}

// Persistent launch: one wave per block, as many blocks as stay resident; each takes jobs from a
// shared counter until none are left.  A wave owns one visited bitset and one spill area for the
// whole launch and leaves the bitset clean after every job, so the scratch is sized by the
// resident waves (not by the batch) and nothing is memset between launches.
template <int METRIC, int NS, bool HASHED, int FORM = kFormPlain>
// float rows: 168 VGPRs, three waves per SIMD; int8 records keep 16 registers of rows in flight, not 64: five waves.
// LAT (launches that do not fill the chip): no occupancy to buy -- every spilled register is a memory round trip a lone wave
// waits out in full -- so two waves per SIMD at most (256 VGPRs), one with eight register sets
__global__ void __launch_bounds__(FORM == kFormLat ? 128 : 64) __attribute__((amdgpu_waves_per_eu(HNSW_WAVES(FORM == kFormLat ? (NS <= 4 ? 2 : 1) : METRIC == M_I8 ? (NS <= 2 ? HNSW_I8_WAVES : 4) : (NS <= 4 ? 3 : 2)))))
graph_search_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, const float *__restrict__ queries,
                    const double *__restrict__ q_sn, int dim, const int *__restrict__ adj0, int stride0,
                    const int64_t *__restrict__ upper, const int *__restrict__ pool, int strideU,
                    const SearchJob *__restrict__ jobs, int k, int cand_cap, ND *__restrict__ spill,
                    int spill_cap, unsigned *__restrict__ visited, long long vis_words, int *__restrict__ vis_tab, int vis_tab_cap, int k_out,
                    int *__restrict__ out_ids, float *__restrict__ out_d, int *__restrict__ out_cnt, int *__restrict__ out_flag,
                    unsigned long long *__restrict__ eval_counter, int nbcap, int njobs, int *__restrict__ job_counter, int overlap,
                    const int *__restrict__ ready)
{
    constexpr bool LAT = FORM == kFormLat, LEAN = FORM == kFormLean;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    // (the lean form looks its visited set and spill area up when a job takes the exact traversal: search_job)
    VisitedSet<HASHED> V{LEAN ? nullptr : visited + (size_t)blockIdx.x * (size_t)vis_words, LEAN ? 0 : vis_words,
                 !LEAN && vis_tab ? vis_tab + (size_t)blockIdx.x * (size_t)vis_tab_cap : nullptr, LEAN ? 0u : (unsigned)(vis_tab_cap - 1), 0, LEAN ? 0 : vis_tab_cap / 4 * 3};
    ND *my_spill = LEAN ? nullptr : spill + (size_t)blockIdx.x * spill_cap;

    TeamPort port{nullptr, 0, 0};
    if constexpr (LAT) {
        // two waves per block (see TeamMail): wave 1 serves the expansions, wave 0 is the traversal.  The mailbox follows
        // the traversal's LDS; its sequence words are zeroed before the roles part (the one barrier both waves meet at).
        TeamMail *mail = reinterpret_cast<TeamMail *>(smem + ((search_lds_bytes(k, cand_cap, dim, false, nbcap) + 15) & ~(size_t)15));
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
    // `ready` (hnsw_knn_query on host buffers): the launch started when the first rows of the query set had landed; the
    // rest is still arriving on the copy engine, and *ready (a word in host memory the uploading thread advances) says how
    // many rows are there.  Jobs are taken in order, so a wave almost never has to wait; when it does it sleeps and
    // polls, for a bounded time -- a job whose row has not arrived by then is handed back (flag 1), never waited for.
    // SHADOW traversals (overlap bit 8; job_counter then is [next job, next shadow, -, -, one word per job ...], all zero
    // at launch).  One traversal in 700 meets equal distances where the heap layout shows and starts over in the exact
    // two-heap form, three times as long as the sorted one; whenever that happened to one of the LAST jobs of a launch,
    // the whole launch waited for it -- 7 % of a 65 536-query launch at C2, 17-35 % of the 12 500-query launches
    // (measured with the re-runs compiled out).  So a wave that finds the queue empty does not leave: it starts the
    // exact traversal of a job another wave is still working on, latest job first.  Almost always the owner answers the
    // job soon after and the shadow stops at its next expansion; when the owner meets a tie it finds the exact
    // traversal already under way and leaves it to the shadow.  Results are written by whoever finishes first -- both
    // compute the reference's answer.
    bool shadows = (overlap & 0x100) != 0 && NS > 0 && !(overlap & 0x200);
    int *job_words = job_counter + 4;
    int known_ready = 0;
    bool v_clean = false;
    for (;;) {
        if constexpr (LEAN) { // what the queue needs, read per job instead of carried through the traversal (kernarg_load)
            job_counter = HNSW_KA(job_counter); njobs = HNSW_KA(njobs); ready = HNSW_KA(ready); jobs = HNSW_KA(jobs); overlap = HNSW_KA(overlap);
            out_cnt = HNSW_KA(out_cnt); out_flag = HNSW_KA(out_flag);
            job_words = job_counter + 4;
            shadows = (overlap & 0x100) != 0 && !(overlap & 0x200);
        }
        int job = 0;
        bool shadow = false;
        if (lane == 0) job = atomicAdd(job_counter, 1);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= njobs) {
            if (!shadows) break;
            int t = 0;
            if (lane == 0) t = atomicAdd(job_counter + 1, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= njobs || t >= (int)gridDim.x) break; // only the last gridDim.x jobs can still be running
            job = njobs - 1 - t;
            if (ready) {
                // a gated launch (query rows still arriving): no shadow for a job whose row has not landed -- its owner is
                // asleep at the gate and search_job would read whatever the previous call left in that row
                const int need = __builtin_amdgcn_readfirstlane(jobs[job].qref);
                if (need >= known_ready) {
                    int r = 0;
                    if (lane == 0) r = __hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    known_ready = __builtin_amdgcn_readfirstlane(r);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    if (need >= known_ready) continue;
                }
            }
            int old = 0;
            if (lane == 0) old = atomicOr(job_words + job, kJobShadowed);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old & (kJobAnswered | kJobTied)) continue; // answered, or its owner is already starting over
            shadow = true;
        } else if (ready) {
            const int need = __builtin_amdgcn_readfirstlane(jobs[job].qref);
            if (need >= known_ready) {
                // a read of host memory per poll: few polls, far apart (thousands of waves polling back to back were
                // measured to starve the very copy they wait for) -- 512 x ~0.2 ms at most, then the job is handed back
                for (int spin = 0; spin < 512; ++spin) {
                    int r = 0;
                    if (lane == 0) r = __hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    known_ready = __builtin_amdgcn_readfirstlane(r);
                    if (need < known_ready) break;
                    for (int z = 0; z < 48; ++z) __builtin_amdgcn_s_sleep(127);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the rows read next were written by the copy engine
                if (need >= known_ready) {
                    // handed back.  With shadows the job word decides who answers, exactly as for a tie (search_job): a
                    // shadow that started because the row landed meanwhile keeps the job; otherwise this wave claims it.
                    int old = 0;
                    if (shadows && lane == 0) {
                        old = atomicOr(job_words + job, kJobTied);
                        if (!(old & (kJobShadowed | kJobAnswered))) old = atomicOr(job_words + job, kJobAnswered) & kJobAnswered;
                    }
                    old = __builtin_amdgcn_readfirstlane(old);
                    if (old == 0 && lane == 0) {
                        out_cnt[job] = 0;
                        out_flag[job] = 1;
                    }
                    continue;
                }
            }
        }
        search_job<METRIC, NS, HASHED, FORM>(rows, row_sn, queries, q_sn, dim, adj0, stride0, upper, pool, strideU, jobs, k, cand_cap, my_spill, spill_cap,
                               V, k_out, out_ids, out_d, out_cnt, out_flag, eval_counter, nbcap, smem, job, overlap, shadows ? job_words + job : nullptr,
                               shadow, &port, &v_clean);
        if (!v_clean) V.clear(lane);
    }

    if constexpr (LAT) port.post(-1, 0, lane); // the memory wave leaves
}

#ifdef HNSW_HOST_TU // few variants and launched from one place: defined only in the unit that launches it
// RangeQuery on the device: FindEntryPointQuery + GraphNavigator.SearchLayerRange (GraphNavigator.cs:262-325)
// for one query per wave.  What the reference's two heaps compute there is a closure: a neighbour enters
// `candidates` and `topCandidates` iff its distance is <= range (:302-308), nothing ever leaves topCandidates
// (its root never exceeds range, :310-311), and the stop test (:286-289) can only fire for the entry point, whose
// farthestResultDist is still MaxValue -- so every listed node and the entry point are expanded exactly once,
// whatever the pop order, and the result SET and the evaluation count do not depend on it.  The order shows only
// in RangeQuery's stable OrderBy over the heap array (HNSWIndex.cs:155) between results of EQUAL distance; the
// host sorts what comes back, and for a query that holds such a pair replays the two heaps from the entry point
// with the distances found here (no evaluation: a neighbour that is not among the results is out of range).
// `found` (per wave, found_cap entries) is both the result list and the work queue: entry `head` is the next
// node to expand.  Results are then copied to a launch-wide arena at an offset taken with one atomic.
// out_flag: 0 done; 1 hand back (more than found_cap results, or the visited table filling up); 3 arena full.
constexpr int kRangeFan = 8; // nodes expanded per step; the id / distance scratch holds kRangeFan adjacency lists
template <int METRIC, bool HASHED>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) // at most 168 VGPRs: three waves per SIMD
graph_range_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, const float *__restrict__ queries,
                   const double *__restrict__ q_sn, int dim, const int *__restrict__ adj0, int stride0,
                   const int64_t *__restrict__ upper, const int *__restrict__ pool, int strideU,
                   const SearchJob *__restrict__ jobs, float range, ND *__restrict__ found_all, int found_cap,
                   unsigned *__restrict__ visited, long long vis_words, int *__restrict__ vis_tab, int vis_tab_cap,
                   ND *__restrict__ arena, unsigned long long arena_cap, unsigned long long *__restrict__ arena_used,
                   unsigned long long *__restrict__ out_off, int *__restrict__ out_cnt, int *__restrict__ out_flag,
                   int *__restrict__ out_entry, unsigned long long *__restrict__ eval_counter, int nbcap, int njobs,
                   int *__restrict__ job_counter)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    VisitedSet<HASHED> V{visited + (size_t)blockIdx.x * (size_t)vis_words, vis_words,
                         vis_tab ? vis_tab + (size_t)blockIdx.x * (size_t)vis_tab_cap : nullptr, (unsigned)(vis_tab_cap - 1), 0, vis_tab_cap / 4 * 3};
    const SearchLds L = carve_lds(smem, 0, 0, dim, nbcap);
    const GraphView G{adj0, stride0, upper, pool, strideU};
    // the queue is read back through L2 (agent-scope loads): a line of it cached earlier may lack later entries
    unsigned long long *found = reinterpret_cast<unsigned long long *>(found_all + (size_t)blockIdx.x * (size_t)found_cap);
    int *nbuf = L.nbuf;
    float *dbuf = L.dbuf;
    for (;;) {
        int job = 0;
        if (lane == 0) job = atomicAdd(job_counter, 1);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= njobs) break;
        const SearchJob jb = jobs[job];
        const float *q = queries + (size_t)jb.qref * dim;
        double sb = 0.0;
        if (METRIC == M_COS) sb = q_sn[jb.qref];
        wave_sync();
        for (int i = lane; i < dim; i += 64) L.qs[i] = q[i];
        unsigned long long evals = 0;
        int best;
        float cur;
        ReadLog RL{nullptr, 0, 0};
        descend<METRIC>(rows, row_sn, dim, sb, G, jb, L, lane, best, cur, evals, RL); // FindEntryPointQuery; :268 reuses its distance
        if (lane == 0) (void)V.first_visit(best);                                  // :279
        V.seen += 1;
        int count = 0, head = 0;
        if (cur <= range) { // :271-275
            if (lane == 0) found[0] = ((unsigned long long)__float_as_uint(cur) << 32) | (unsigned)best;
            count = 1;
        }
        // :277 the entry point is a candidate either way; out of range it is still expanded, unless its distance
        // exceeds farthestResultDist's initial MaxValue (+inf): then :286-289 ends the search at once
        bool entry_pending = !(cur <= range) && !(cur > 3.402823466e+38f);
        bool ok = true;
        for (;;) {
            // up to kRangeFan listed nodes are expanded per step (any order gives the same set): a large result set
            // is a long dependent chain on one wave otherwise
            int W, c0 = best;
            if (entry_pending) { W = 1; entry_pending = false; }
            else {
                W = min(kRangeFan, count - head); // :283 no candidates left
                if (W == 0) break;
                unsigned long long e = 0ull;
                if (lane < W) e = __hip_atomic_load(&found[head + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // :285, :290
                c0 = (int)(unsigned)e;
                head += W;
            }
            int n[kRangeFan], nb[kRangeFan];
            const int *lw[kRangeFan];
#pragma unroll
            for (int w = 0; w < kRangeFan; ++w) {
                lw[w] = G.list(__builtin_amdgcn_readlane(c0, w < W ? w : 0), 0);
                n[w] = w < W ? __builtin_amdgcn_readfirstlane(lw[w][0]) : 0;
                nb[w] = lane < n[w] ? lw[w][1 + lane] : 0;
            }
            bool fr[kRangeFan];
#pragma unroll
            for (int w = 0; w < kRangeFan; ++w) fr[w] = lane < n[w] && V.first_visit(nb[w]); // :297 / :318 (a node two lists share is fresh once)
            int m = 0;
            wave_sync();
#pragma unroll
            for (int w = 0; w < kRangeFan; ++w) {
                const unsigned long long mask = __ballot(fr[w]);
                const int posn = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                if (fr[w]) nbuf[m + posn] = nb[w];
                m += __popcll(mask);
                for (int base = 64; base < n[w]; base += 64) { // lists beyond 64 ids (MaxEdges > 32)
                    const int i = base + lane;
                    bool fresh = false;
                    int x = 0;
                    if (i < n[w]) { x = lw[w][1 + i]; fresh = V.first_visit(x); }
                    const unsigned long long mk = __ballot(fresh);
                    const int pp = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
                    if (fresh) nbuf[m + pp] = x;
                    m += __popcll(mk);
                }
            }
            wave_sync();
            if (m == 0) continue;
            V.seen += m;
            if (V.crowded()) { ok = false; break; }
            measure_all<METRIC>(rows, row_sn, dim, L.qs, sb, nbuf, dbuf, m, lane); // :299
            wave_sync();
            evals += (unsigned long long)m;
            for (int base = 0; base < m && ok; base += 64) {
                const int i = base + lane;
                const float d = i < m ? dbuf[i] : 0.0f;
                const bool in = i < m && d <= range; // :302
                const unsigned long long mask = __ballot(in);
                const int add = __popcll(mask);
                if (count + add > found_cap) { ok = false; break; }
                const int posn = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                if (in) found[count + posn] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)nbuf[i]; // :305, :308
                count += add;
            }
            if (!ok) break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the queue entries just written are read above (through L2)
        }
        unsigned long long off = 0;
        int flag = ok ? 0 : 1;
        if (ok && count > 0) {
            if (lane == 0) off = atomicAdd(arena_used, (unsigned long long)count);
            off = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)off);
            if (off + (unsigned long long)count > arena_cap) flag = 3;
            else
                for (int i = lane; i < count; i += 64) {
                    const unsigned long long e = __hip_atomic_load(&found[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    arena[off + i] = ND{(int)(unsigned)e, __uint_as_float((unsigned)(e >> 32))};
                }
        }
        if (lane == 0) {
            out_off[job] = off;
            out_cnt[job] = flag == 0 ? count : 0;
            out_flag[job] = flag;
            out_entry[job] = best; // FindEntryPointQuery's answer: where a host replay of the heaps starts
            if (flag != 3) atomicAdd(eval_counter, evals); // (a job that found the arena full runs again)
        }
        V.clear(lane);
    }
}
#endif

} // namespace hnsw
