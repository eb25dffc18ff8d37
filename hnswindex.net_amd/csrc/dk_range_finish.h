// dk_range_finish.h -- device code, part of device_kernels.h: RangeQuery's ORDER on the device (round 5).
// graph_range_kernel leaves each query's result SET in the launch's arena, in the order the wave found it.  What the reference
// returns is topCandidates.ToArray() -- the heap's ARRAY -- stably sorted by distance (HNSWIndex.cs:155, GraphNavigator.cs:324):
//   * a list of pairwise different distances has one ascending order: range_sort_kernel ranks it by counting, in place;
//   * between results of EQUAL distance the heap array's order decides, and that is a property of the whole push history: such
//     a list is replayed -- SearchLayerRange's two heaps (GraphNavigator.cs:262-325) run again from the entry point with every
//     distance already known (a neighbour that is not in the list is out of range: marked visited and dropped, :302, :318, no
//     evaluation), range_replay_kernel -- and the heap array is then ranked stably (key, array index).
// Until round 5 both steps ran on host threads (csrc/range_replay.h, std::sort): 15 + 58 ms of a 140-ms call of 16 384 queries at
// 340 results each, beside 27 ms of traversal kernel.  The host forms remain for what these kernels hand back: lists beyond
// kRangeSortMax entries, a -0 distance (key order is not float.CompareTo order there), the lock-step mode.
#pragma once
#include "dk_heaps.h"

namespace hnsw {

constexpr int kRangeSortMax = 2048;  // entries a wave ranks in LDS (16 KB of (id, key) pairs)
// (per-job state after the finishing kernels -- kRangeFinal / kRangeTied / kRangeHostSort -- in device_backend.h: the host reads it)

#ifdef HNSW_HOST_TU // non-template kernels: only the unit that launches them defines them

// Stable ranking of m <= kRangeSortMax (id, key) pairs held in LDS `e` by (key, position): out[rank] = entry.  Returns (wave-uniform)
// whether two entries share a key.  Every lane owns the entries lane, lane + 64, ...; the inner loop reads e[j] for all lanes at
// once (an LDS broadcast).
__device__ __forceinline__ bool rank_stable(const int2 *e, int m, ND *out, const float *dist_of, int lane)
{
    bool tie = false;
    for (int i0 = 0; i0 < m; i0 += 64) {
        const int i = i0 + lane;
        const unsigned ki = i < m ? (unsigned)e[i].y : 0u;
        int below = 0;
        bool eq = false;
        for (int j = 0; j < m; ++j) {
            const unsigned kj = (unsigned)e[j].y;
            below += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
            eq = eq || (kj == ki && j != i);
        }
        if (i < m) out[below] = ND{e[i].x, dist_of ? dist_of[i] : key2f(ki)};
        tie = tie || (i < m && eq);
    }
    return __ballot(tie) != 0ull;
}

// One wave per finished range job (persistent over the launch's jobs): the job's list, in the arena at off[job], becomes ascending.
// state[job] = kRangeFinal / kRangeTied / kRangeHostSort; tied jobs are appended to tied[1 ...] (tied[0] = their number).
__global__ void __launch_bounds__(64)
range_sort_kernel(ND *__restrict__ arena, const unsigned long long *__restrict__ off, const int *__restrict__ cnt, const int *__restrict__ flag,
                  int njobs, int *__restrict__ state, int *__restrict__ tied, int *__restrict__ job_counter)
{
    __shared__ int2 e[kRangeSortMax];
    const int lane = threadIdx.x;
    for (;;) {
        int job = 0;
        if (lane == 0) job = atomicAdd(job_counter, 1);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= njobs) break;
        // (No `continue` behind an `if (lane == 0)`: the compiler may then let the other lanes run ahead into the next iteration,
        // where readfirstlane picks THEIR job word -- 0 -- and the wave never reconverges.  One store point, at the end.)
        const int m = __builtin_amdgcn_readfirstlane(cnt[job]);
        int st = kRangeFinal; // (handed-back jobs carry no list; a list of one entry is in order)
        if (__builtin_amdgcn_readfirstlane(flag[job]) == 0 && m > 1) {
            if (m > kRangeSortMax) st = kRangeHostSort;
            else {
                ND *a = arena + off[job];
                bool unsafe = false;
                wave_sync();
                for (int i = lane; i < m; i += 64) {
                    const ND v = a[i];
                    unsafe = unsafe || key_unsafe(v.dist); // -0 (NaN cannot be in range): float.CompareTo ties it with +0, the integer key does not
                    e[i] = make_int2(v.id, (int)f2key(v.dist));
                }
                wave_sync();
                if (__ballot(unsafe) != 0ull) st = kRangeHostSort;
                else st = rank_stable(e, m, a, nullptr, lane) ? kRangeTied : kRangeFinal;
            }
        }
        wave_sync();
        if (lane == 0) {
            state[job] = st;
            if (st == kRangeTied) tied[1 + atomicAdd(&tied[0], 1)] = job;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// LDS of one replaying wave: entries (id, key) in ascending order, an id -> entry hash, visited bits, the two heaps.
struct RangeReplayLds {
    int2 e[kRangeSortMax];                      // the list, ascending (as range_sort_kernel left it)
    unsigned short slot[2 * kRangeSortMax];     // open addressing: entry index + 1, 0 = empty
    unsigned visited[kRangeSortMax / 32];
    ND top[kRangeSortMax], cand[kRangeSortMax]; // heap entries: {id = entry index, dist = key bits}
};

// SearchLayerRange replayed on known distances, one wave per tied job: the wave's lanes look a popped node's neighbours up side by
// side; the heaps move under the reference's own sift rules (heap_push / heap_pop_wave, dk_heaps.h).  Afterwards the top heap's
// ARRAY, stably ranked by key, is the reference's answer and overwrites the list in the arena; state[job] = kRangeFinal.
__global__ void __launch_bounds__(64)
range_replay_kernel(ND *__restrict__ arena, const unsigned long long *__restrict__ off, const int *__restrict__ cnt, const int *__restrict__ entry,
                    const int *__restrict__ adj0, int stride0, long long n_nodes, float range, int *__restrict__ state,
                    const int *__restrict__ tied, int *__restrict__ job_counter)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    RangeReplayLds &L = *reinterpret_cast<RangeReplayLds *>(smem);
    const int lane = threadIdx.x;
    const int n_tied = tied[0];
    const LdsHeap top{L.top}, cand{L.cand};
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(job_counter, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= n_tied) break;
        const int job = __builtin_amdgcn_readfirstlane(tied[1 + t]);
        const int m = __builtin_amdgcn_readfirstlane(cnt[job]);
        ND *a = arena + off[job];
        int st = kRangeTied; // what the host finds if this wave gives up: an ascending list to replay
        if (m >= 2 && m <= kRangeSortMax) { // (always: range_sort_kernel listed it)
        wave_sync();
        for (int i = lane; i < 2 * kRangeSortMax; i += 64) L.slot[i] = 0;
        for (int i = lane; i < kRangeSortMax / 32; i += 64) L.visited[i] = 0u;
        for (int i = lane; i < m; i += 64) { const ND v = a[i]; L.e[i] = make_int2(v.id, (int)f2key(v.dist)); }
        wave_sync();
        // the table is filled by one lane at a time per slot (atomicCAS on LDS shorts is not available: 32-bit words of two slots)
        unsigned *slot32 = reinterpret_cast<unsigned *>(L.slot);
        constexpr unsigned kMask = 2 * kRangeSortMax - 1;
        for (int i = lane; i < m; i += 64) {
            unsigned h = ((unsigned)L.e[i].x * 2654435761u) & kMask;
            for (unsigned probes = 0; probes <= kMask; ++probes) {
                const unsigned w = h >> 1, sh = (h & 1u) * 16u;
                const unsigned old = slot32[w];
                if (((old >> sh) & 0xffffu) == 0u) {
                    if (atomicCAS(&slot32[w], old, old | ((unsigned)(i + 1) << sh)) == old) break;
                    continue; // the word changed under us (its other half, or this one): look again
                }
                h = (h + 1) & kMask;
            }
        }
        wave_sync();
        auto find = [&](int id) -> int { // entry index of id, or -1
            unsigned h = ((unsigned)id * 2654435761u) & kMask;
            for (unsigned probes = 0; probes <= kMask; ++probes) {
                const unsigned s = L.slot[h];
                if (s == 0u) return -1;
                if (L.e[s - 1].x == id) return (int)s - 1;
                h = (h + 1) & kMask;
            }
            return -1;
        };
        int n_top = 0, n_cand = 0;
        const int ep = entry[job];
        const int es = __builtin_amdgcn_readfirstlane(find(ep));
        const unsigned range_key = f2key(range);
        unsigned farthest = 0xffffffffu; // :269 MaxValue
        bool entry_out = es < 0;         // the entry point lies out of range: a candidate all the same (:277), never a result
        if (!entry_out) { // :271-275, :279
            const HEnt en{es, (unsigned)L.e[es].y};
            heap_push<false>(top, n_top, en);
            farthest = en.key;
            if (lane == 0) L.visited[es >> 5] |= 1u << (es & 31);
            heap_push<true>(cand, n_cand, en);
        }
        wave_sync();
        bool first = true;
        for (int step = 0; step <= m + 1; ++step) { // (every listed node and the entry point are expanded once: m + 1 steps at most)
            int node;
            if (entry_out && first) node = ep; // alone in `candidates` when popped; :286 cannot fire (farthestResultDist is still MaxValue)
            else {
                if (n_cand == 0) break;                                  // :283
                const HEnt c = cand.get(0);                              // :285
                if (c.key > farthest && c.key > range_key) break;        // :286-289 (every candidate is in range: never fires)
                const HEnt popped = heap_pop_wave<true>(cand, n_cand, lane); // :290
                wave_sync();
                node = L.e[popped.id].x;
            }
            first = false;
            if (node < 0 || (long long)node >= n_nodes) break; // (guard: ids come from the graph the list was found on)
            const int *l = adj0 + (size_t)node * stride0;
            const int n = __builtin_amdgcn_readfirstlane(l[0]);
            for (int base = 0; base < n; base += 64) {                   // :294
                const int i = base + lane;
                int idx = -1;
                if (i < n) idx = find(l[1 + i]);
                bool fresh = idx >= 0 && ((L.visited[idx >> 5] >> (idx & 31)) & 1u) == 0u; // :297; not listed = out of range (:302 fails, :318)
                wave_sync();
                if (fresh) atomicOr(&L.visited[idx >> 5], 1u << (idx & 31));             // :318 (a list holds no duplicates)
                unsigned long long mask = __ballot(fresh);
                while (mask) {                                           // in adjacency order
                    const int src = (int)__builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int ix = __builtin_amdgcn_readlane(idx, src);
                    const HEnt sel{ix, (unsigned)L.e[ix].y};
                    heap_push<true>(cand, n_cand, sel);                  // :305
                    heap_push<false>(top, n_top, sel);                   // :308 (:310-311 never pops: sel is in range)
                    farthest = top.get(0).key;                           // :313-314
                }
                wave_sync();
            }
        }
        wave_sync();
        // the top heap's array, stably ranked by key (HNSWIndex.cs:155): entries of the arena list that the replay did not reach
        // cannot exist (the kernel's closure and the heaps' closure are the same set); if they do, leave the job to the host
        if (n_top == m) {
            int2 *h2 = reinterpret_cast<int2 *>(L.cand); // the candidate heap is empty now: its LDS holds (id, key) in heap-array order
            for (int i = lane; i < m; i += 64) { const ND t2 = L.top[i]; h2[i] = make_int2(L.e[t2.id].x, L.e[t2.id].y); }
            wave_sync();
            // distances: the original bits (a key maps back to its float exactly; -0 lists never get here)
            (void)rank_stable(h2, m, a, nullptr, lane);
            st = kRangeFinal;
        }
        }
        wave_sync();
        if (lane == 0) state[job] = st; // (one store point, no `continue` behind a lane-0 branch: see range_sort_kernel)
        __builtin_amdgcn_wave_barrier();
    }
}
#endif

} // namespace hnsw
