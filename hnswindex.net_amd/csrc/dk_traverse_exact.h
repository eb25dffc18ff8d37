// dk_traverse_exact.h -- device code, part of device_kernels.h: the exact two-heap traversal (what a tie falls back to).
#pragma once
#include "dk_search_common.h"

namespace hnsw {

// Descent + beam search of one job; result = L.top[0..top_n) in heap order.  Returns false on
// candidate-heap overflow.  The query must already be staged in L.qs.
template <int METRIC, bool HASHED>
__device__ __forceinline__ bool traverse(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, double sb,
                                         const GraphView &G, const SearchJob jb, int k, int cand_cap, ND *spill, int spill_cap,
                                         VisitedSet<HASHED> &V, const SearchLds &L, int lane, int &top_n_out, unsigned long long &evals,
                                         ReadLog &RL, const int *abort_word = nullptr, bool *aborted = nullptr, bool overlapped_form = false)
{
    const LdsHeap top{L.top};
    const SpillHeap cand{L.cand, cand_cap, spill};
    const int cand_limit = cand_cap + spill_cap;
    int *nbuf = L.nbuf;
    float *dbuf = L.dbuf;
    const float *qs = L.qs;
    // ---- FindEntryPoint / FindEntryAtLayer (GraphNavigator.cs:27-82) ----
    int best = jb.entry;
    wave_sync();
    if (lane == 0) nbuf[0] = best;
    wave_sync();
    measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, 1, lane);
    wave_sync();
    float cur = dbuf[0]; // :57
    evals += 1;
    for (int layer = jb.entry_layer; layer > jb.search_layer; --layer) {
        bool changed = true;
        RL.layer(layer, lane);
        while (changed) { // :60
            changed = false;
            const int *l = G.list(best, layer);
            const int n = l[0];
            RL.put(best, lane);
            wave_sync();
            for (int i = lane; i < n; i += 64) nbuf[i] = l[1 + i]; // :65 span taken once per pass
            wave_sync();
            if (n > 0) measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, n, lane);
            wave_sync();
            evals += (unsigned long long)n;
            for (int i = 0; i < n; ++i) { // :67-78
                float d = dbuf[i];
                if (d < cur) { cur = d; best = nbuf[i]; changed = true; }
            }
        }
    }
    // ---- SearchLayer (GraphNavigator.cs:123-189) ----
    const int layer = jb.search_layer;
    RL.layer(layer, lane);
    int top_n = 0, cand_n = 0;
    bool overflow = false; // also raised for NaN / -0 distances (see f2key)
    bool hash_full = false;
    best = __builtin_amdgcn_readfirstlane(best);
    cur = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cur)));
    if (key_unsafe(cur)) overflow = true;
    // jb.aux == -2 (removal's search, GraphConnector.cs:96): the filter id != entry keeps the entry point out of the
    // results (:132-136) -- it is a candidate only, and farthestResultDist starts at MaxValue
    const bool entry_filtered = jb.aux == -2;
    {
        HEnt e{best, f2key(cur)};
        if (!entry_filtered) heap_push<false>(top, top_n, e); // :134
        heap_push<true>(cand, cand_n, e); // :138
        if (lane == 0) (void)V.first_visit(best);                       // :140
            V.seen += 1;
    }
    unsigned far_key = entry_filtered ? 0xffffffffu : f2key(cur); // farthestResultDist :135
    // Speculative prefetch of the NEXT expansion's out-edge list: while the current candidate
    // rows are in flight, lanes 0..stride fetch the list of the heap's current root.  If that
    // node is indeed popped next (it is, unless this expansion pushes something closer) its list
    // is already in registers and one dependent memory round trip disappears.
    int pre_id = -1, pre_a = 0, pre_b = 0;
    const int lstride = layer == 0 ? G.stride0 : G.strideU;
    int abort_v = 0; // a shadow traversal (graph_search_kernel): bit 0 of *abort_word = the job has been answered
    while (cand_n > 0 && !overflow) {
        if (abort_word) {
            // read now, looked at one expansion later: the load rides with this expansion's own
            if (__builtin_amdgcn_readfirstlane(abort_v) & 1) { *aborted = true; return false; }
            abort_v = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        HEnt c = heap_pop_wave<true>(cand, cand_n, lane); // :146
        if (c.key > far_key && top_n >= k) break;       // :147-150
        RL.put(c.id, lane);
        int n, nb_a = 0, nb_b = 0; // this lane's neighbour ids (list positions lane and lane + 64)
        if (c.id == pre_id) {
            n = __builtin_amdgcn_readlane(pre_a, 0);
            nb_a = __shfl(pre_a, (lane + 1) & 63, 64);            // list word lane + 1
            const int w64 = __builtin_amdgcn_readlane(pre_b, 0);  // list word 64
            if (lane == 63) nb_a = w64;
            nb_b = __shfl(pre_b, (lane + 1) & 63, 64);            // list word lane + 65
        } else {
            const int *l = G.list(c.id, layer);
            n = __builtin_amdgcn_readfirstlane(l[0]);
            if (lane < n) nb_a = l[1 + lane];
            if (lane + 64 < n) nb_b = l[65 + lane];
        }
        int m = 0;
        wave_sync();
        bool have = false; // overlapped form: this lane holds an unvisited neighbour, its distance and id
        float lane_d = 0.0f;
        int lane_id = 0;
        const bool overlapped = overlapped_form && n <= 64;
        if (overlapped) {
            // as in traverse_sorted: the rows of ALL listed neighbours requested together with the visited atomics -- one
            // dependent round trip less per expansion.  This traversal runs where a launch is draining (a re-run, a
            // shadow) or in launches that do not fill the chip; the rows of visited neighbours are bandwidth nobody misses.
            const bool in = lane < n;
            if (in) nbuf[lane] = nb_a;
            wave_sync();
            unsigned old = 0u;
            const unsigned bit = 1u << (nb_a & 31);
            unsigned hpos = 0u;
            if constexpr (HASHED) {
                hpos = ((unsigned)nb_a * 2654435761u) & V.tab_mask;
                if (in) old = (unsigned)atomicCAS(&V.tab[hpos], -1, nb_a);
            } else if (in) old = atomicOr(&V.bits[nb_a >> 5], bit); // :181
            pre_id = -1;
            if (cand_n > 0) {
                pre_id = cand.get(0).id;
                const int *pl = G.list(pre_id, layer);
                pre_a = lane < lstride ? pl[lane] : 0;
                pre_b = lane + 64 < lstride ? pl[lane + 64] : 0;
            }
            if (n > 0) measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, n, lane); // :163 (and the visited ones)
            wave_sync();
            if constexpr (HASHED) {
                have = in && (int)old == -1;
                if (in && (int)old != -1 && (int)old != nb_a) { // slot taken by another id: probe on (VisitedSet::first_visit)
                    for (unsigned probes = 0; probes <= V.tab_mask; ++probes) {
                        hpos = (hpos + 1) & V.tab_mask;
                        const int o2 = atomicCAS(&V.tab[hpos], -1, nb_a);
                        if (o2 == -1) { have = true; break; }
                        if (o2 == nb_a) break;
                    }
                }
            } else have = in && (old & bit) == 0u;
            m = (int)__popcll(__ballot(have));
            V.seen += m;
            if (V.crowded()) { hash_full = true; break; }
            lane_d = in ? dbuf[lane] : 0.0f;
            lane_id = nb_a;
            if (m == 0) continue;
            evals += (unsigned long long)m;
        } else {
        for (int base = 0; base < n; base += 64) { // :158-161 keep only unvisited, in list order
            const int i = base + lane;
            bool fresh = false;
            const int nb = base == 0 ? nb_a : nb_b;
            if (i < n) fresh = V.first_visit(nb); // :181 (lists hold no duplicates)
            const unsigned long long mask = __ballot(fresh);
            const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            if (fresh) nbuf[m + pos] = nb;
            m += __popcll(mask);
        }
        pre_id = -1;
        if (cand_n > 0) {
            pre_id = cand.get(0).id;
            const int *pl = G.list(pre_id, layer);
            pre_a = lane < lstride ? pl[lane] : 0;
            pre_b = lane + 64 < lstride ? pl[lane + 64] : 0;
        }
        wave_sync();
        if (m == 0) continue;
        V.seen += m;
        if (V.crowded()) { hash_full = true; break; } // the id table is filling up: host traversal
        measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, m, lane); // :163
        wave_sync();
        evals += (unsigned long long)m;
        }
        // Replay of the push loop (:165-178) in adjacency order.  farthest never grows once the
        // result heap is full, so a candidate that fails `d < farthest` now can never pass later:
        // only the lanes of the ballot are visited, and the exact test is repeated on each.
        const int rounds = overlapped ? 1 : (m + 63) / 64;
        for (int r = 0; r < rounds && !overflow; ++r) {
            const int i = r * 64 + lane;
            const bool valid = overlapped ? have : i < m;
            const float my_d = overlapped ? lane_d : (i < m ? dbuf[i] : 0.0f);
            const int my_id = overlapped ? lane_id : (i < m ? nbuf[i] : 0);
            const unsigned my_key = f2key(my_d);
            if (__ballot(valid && key_unsafe(my_d))) { overflow = true; break; }
            unsigned long long maybe = __ballot(valid && (top_n < k || my_key < far_key));
            while (maybe) {
                const int src = __builtin_ctzll(maybe);
                maybe &= maybe - 1;
                const unsigned dk = (unsigned)__builtin_amdgcn_readlane((int)my_key, src);
                if (top_n < k || dk < far_key) { // :165
                    HEnt sel{__builtin_amdgcn_readlane(my_id, src), dk};
                    if (cand_n >= cand_limit) { overflow = true; break; }
                    heap_push<true>(cand, cand_n, sel);               // :168
                    heap_push<false>(top, top_n, sel);                // :171
                    if (top_n > k) (void)heap_pop_wave<false>(top, top_n, lane); // :173-174
                    far_key = top.get(0).key;                         // :176-177
                }
            }
        }
    }
    // back to float distances for the callers (ToArray(): heap order, BinaryHeap.cs:41-44)
    wave_sync();
    for (int i = lane; i < top_n; i += 64) L.top[i].dist = key2f(__float_as_uint(L.top[i].dist));
    wave_sync();
    top_n_out = top_n;
    return !overflow && !hash_full;
}

} // namespace hnsw
