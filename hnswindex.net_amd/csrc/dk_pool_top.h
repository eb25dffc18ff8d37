// dk_pool_top.h -- device code, part of device_kernels.h: SearchLayer on an unsorted register pool (the latency variants' logic wave): PoolTop, traverse_pool.
#pragma once
#include "dk_team.h"

namespace hnsw {

// ---- SearchLayer on an UNSORTED pool in registers (the latency variants' logic wave) -----------------------------
// traverse_sorted keeps the beam as one ascending list, and every insertion ranks the newcomers against all of it: a few
// hundred instructions per expansion, which a full chip hides behind other waves' memory traffic and a lone wave pays in
// full (phase clocks of B = 1 inserts, two-wave form: 7 500 of an expansion's 8 500 clocks were the list's upkeep).  But
// nothing SearchLayer does needs an order: it removes the closest open candidate (:146), replaces the farthest result
// when a closer one arrives (:165-178) and asks for the farthest distance -- a minimum and a maximum.  So the logic wave
// of the latency variants keeps the k results in register slots in no particular order (slot s in lane s mod 64 of
// register set s / 64; bit 31 of the id = expanded, bit 30 = doubtful, as in SortedTop) and runs the reference's own
// loop on them: pop = wave-wide minimum over the open slots (four DPP steps inside the rows of 16 lanes, four readlanes),
// push = the slot of the farthest entry takes the newcomer, then a wave-wide maximum; the list is sorted ONCE, when the
// search is over (ranks by counting through LDS), and handed on ascending like the sorted list's.
// Equal distances: the rules of traverse_sorted, stated on keys instead of positions.  (i) the farthest result leaves
// while another entry has its distance (the maximum does not change): the survivors of that distance become doubtful
// (hard unless the one that left and all of them were expanded); (ii) the closest open candidate has an open twin: a
// group window opens (members counted by key; closes at the first pop beyond the key with all members still present);
// (a), (b), (d) inside a window and (c) at its end as there; (iii) is read off the sorted output.  Which of several
// equal entries a minimum or maximum picks differs from the sorted list (lowest slot here, first position there) -- in
// exactly the situations these rules either prove immaterial or hand to the exact two-heap traversal.
// v_writelane_b32: a uniform value into ONE lane of a register.  (No builtin reaches it.  One scalar register per VALU
// instruction on this ISA: the lane select goes through M0, as the compiler's own lowering of the intrinsic does.)
__device__ __forceinline__ int lane_write(int value, int lane_sel, int old)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(value), "s"(lane_sel) : "m0");
    return old;
}
template <int NS>
struct PoolTop {
    unsigned key[NS];  // unused slots: 0 (no distance has that key, and it never is the maximum)
    unsigned okey[NS]; // the key while the entry is open, 0xffffffff once it is expanded (and in unused slots): what pops look at
    int id[NS];        // unused slots: expanded bit set
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int t = 0; t < NS; ++t) { key[t] = 0u; okey[t] = 0xffffffffu; id[t] = (int)0x80000000; }
    }
    // slot = 64 t + lane, uniform: one v_readlane / v_writelane per register touched
    __device__ __forceinline__ int id_at(int slot) const
    {
        const int l = slot & 63;
        int v = 0;
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if ((slot >> 6) == t) v = __builtin_amdgcn_readlane(id[t], l);
        return v;
    }
    __device__ __forceinline__ void put(int slot, unsigned k0, int i0) // a new, open entry
    {
        const int l = slot & 63;
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if ((slot >> 6) == t) {
                key[t] = (unsigned)lane_write((int)k0, l, (int)key[t]);
                okey[t] = (unsigned)lane_write((int)k0, l, (int)okey[t]);
                id[t] = lane_write(i0, l, id[t]);
            }
    }
    __device__ __forceinline__ void mark_expanded(int slot, int idword) // idword: the entry's id word as it reads now
    {
        const int l = slot & 63;
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if ((slot >> 6) == t) {
                okey[t] = (unsigned)lane_write(-1, l, (int)okey[t]);
                id[t] = lane_write(idword | (int)0x80000000, l, id[t]);
            }
    }
    // where a key sits: the lowest slot holding it (-1: nowhere) and how many slots do
    template <bool OPEN>
    __device__ __forceinline__ void locate(unsigned k0, int &slot, int &count) const
    {
        slot = -1; count = 0;
#pragma unroll
        for (int t = NS - 1; t >= 0; --t) {
            const unsigned long long bm = __ballot((OPEN ? okey[t] : key[t]) == k0);
            count += (int)__popcll(bm);
            if (bm) slot = 64 * t + (int)__builtin_ctzll(bm);
        }
    }
    // the closest open entry: its key (0xffffffff: none), slot (-1), id word, and how many open entries share the key
    __device__ __forceinline__ void min_open(unsigned &mk, int &slot, int &eid, int &nsame) const
    {
        unsigned v = okey[0];
#pragma unroll
        for (int t = 1; t < NS; ++t) v = min(v, okey[t]);
        mk = wave_min_u32(v);
        slot = -1; eid = 0; nsame = 0;
        if (mk == 0xffffffffu) return;
        locate<true>(mk, slot, nsame);
        eid = id_at(slot);
    }
    __device__ __forceinline__ unsigned max_key() const // the farthest entry's key
    {
        unsigned v = key[0];
#pragma unroll
        for (int t = 1; t < NS; ++t) v = max(v, key[t]);
        return wave_max_u32(v);
    }
    __device__ __forceinline__ int count_key(unsigned k0) const // entries of that key (uniform)
    {
        int c = 0;
#pragma unroll
        for (int t = 0; t < NS; ++t) c += (int)__popcll(__ballot(key[t] == k0));
        return c;
    }
    __device__ __forceinline__ void mark_key(unsigned k0, int bit)
    {
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if (key[t] == k0) id[t] |= bit;
    }
    __device__ __forceinline__ bool any_open_key(unsigned k0) const
    {
        unsigned long long m = 0ull;
#pragma unroll
        for (int t = 0; t < NS; ++t) m |= __ballot(okey[t] == k0);
        return m != 0ull;
    }
};

// The contract of traverse_sorted (same arguments, same results: L.top[0..top_n) ascending, tie / order_tie / window,
// read log, evaluation count), for the logic wave of a latency variant: expansions are served by the memory wave
// through `port` (TeamMail).
template <int METRIC, int NS, bool HASHED>
__device__ __forceinline__ bool traverse_pool(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, double sb,
                                              const GraphView &G, const SearchJob jb, int k, int ordered_prefix, VisitedSet<HASHED> &V,
                                              const SearchLds &L, int lane, int &top_n_out, bool &tie_out, unsigned long long &evals,
                                              ReadLog &RL, bool *order_tie_out, bool *window_out, TeamPort *port)
{
    PH_DECL();
    int best;
    float cur;
    descend<METRIC, true>(rows, row_sn, dim, sb, G, jb, L, lane, best, cur, evals, RL, order_tie_out ? 0 : jb.stop_layer); // (a search job's stop_layer carries entry_block_kernel's hint)
    // ---- SearchLayer (GraphNavigator.cs:123-189) ----
    const int layer = jb.search_layer;
    RL.layer(layer, lane);
    constexpr int kDoubt = 0x40000000, kIdMask = 0x3fffffff;
    PoolTop<NS> T;
    T.init();
    int top_n = 0;
    bool unsafe = key_unsafe(cur); // NaN / -0 (see f2key)
    bool tie = false, hash_full = false;
    T.put(0, f2key(cur), best);                                         // :134, :138
    top_n = 1;
    if (lane == 0) (void)V.first_visit(best);                           // :140
    __builtin_amdgcn_s_waitcnt(0); // (the memory wave's marks follow: this one has landed)
    V.seen += 1;
    unsigned far_key = f2key(cur);                                      // farthestResultDist :135
    const bool ids_matter_everywhere = order_tie_out != nullptr; // an insert's heuristic reads the whole list; a search its first entries
    bool doubt_hard = false;
    unsigned grp_key = 0u; // the group window of (ii): its distance and its members (0: no window open)
    int grp_cnt = 0;
    int early_id = -1;     // the node whose expansion was requested before its pop (-1: none) ...
    int early_pos = 0, early_nsame = 0; // ... the slot it sits in, and how many open entries share its key
    unsigned early_key = 0u;
    PH(0);
    while (!unsafe && !tie) {
        unsigned ck;
        int pos, cid, nsame;
        if (early_id >= 0) {
            // the pop was foreseen (below): its slot, key and twins are known, its id word is re-read (a doubt may have been
            // marked since), and the memory wave has been on its expansion since before the last insertions
            pos = early_pos; ck = early_key; nsame = early_nsame;
            cid = T.id_at(pos);
            if ((cid & kIdMask) != early_id || cid < 0) { tie = true; break; } // (cannot happen: the exact traversal decides)
            early_id = -1;
        } else {
            T.min_open(ck, pos, cid, nsame);                             // :146 closest candidate; none left <=> :147-150 / empty
            if (pos < 0) break;
            port->post(cid & kIdMask, layer, lane, top_n >= k ? far_key : 0xffffffffu);
        }
        if (cid & kDoubt) { tie = true; break; } // the reference may be expanding its twin instead
        if (grp_cnt > 0 && ck > grp_key) { // the group window closes: (c) every member still listed?
            if (T.count_key(grp_key) != grp_cnt) { tie = true; break; }
            grp_cnt = 0;
            if (window_out) *window_out = true;
        }
        T.mark_expanded(pos, cid);
        RL.put(cid & kIdMask, lane, top_n >= k && grp_cnt == 0 ? far_key : 0xffffffffu);
        // what would be popped next if this expansion brought nothing closer; (ii): an open twin of the popped candidate
        unsigned nxt_key;
        int npos, nid, nn;
        T.min_open(nxt_key, npos, nid, nn);
        const int nxt_id = npos >= 0 ? (nid & kIdMask) : -1;
        if (nsame > 1) {
            if (grp_cnt == 0) { grp_key = ck; grp_cnt = T.count_key(ck); }
            else if (ck != grp_key) tie = true; // (d)
        }
        if (lane == 0) port->m->hint_node = nxt_id; // (a list for the memory wave to prefetch: stale or missing, nothing breaks)
        PH(1);
        port->wait(); // ids, keys and masks of this node's neighbours
        PH(4);
        const TeamMail *mail = port->m;
        // the answer: its header in two 16-byte reads, ids and keys one per lane -- all four requested before anything is looked at
        const int4 h0 = *reinterpret_cast<const int4 *>(&mail->rsp_seq);
        const uint4 h1 = *reinterpret_cast<const uint4 *>(&mail->fresh);
        const int my_id = mail->ids[lane];
        const unsigned my_key = __float_as_uint(mail->dist[lane]);
        const int nw = __builtin_amdgcn_readfirstlane(h0.y);
        const unsigned bk0 = (unsigned)__builtin_amdgcn_readfirstlane(h0.z);
        const int bl0 = __builtin_amdgcn_readfirstlane(h0.w);
        const unsigned long long fresh = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)h1.y) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)h1.x);
        const unsigned long long passm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)h1.w) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)h1.z);
        if ((nw & 0xffff) > 64) { hash_full = true; break; }
        const int m = (int)__popcll(fresh);
        PH_COUNT(7, 1);
        V.seen += m;
        if (V.crowded()) { hash_full = true; break; }
        if (m == 0) {
            if (grp_cnt == 0 && !tie && nxt_id >= 0) {
                early_id = nxt_id; early_pos = npos; early_key = nxt_key; early_nsame = nn;
                port->post(nxt_id, layer, lane, top_n >= k ? far_key : 0xffffffffu);
            }
            continue;
        }
        evals += (unsigned long long)m;
        if (nw & 0x10000) { unsafe = true; break; }
        if (grp_cnt > 0) { // (a), (b)
            const bool valid = ((fresh >> lane) & 1ull) != 0ull;
            if (__ballot(valid && my_key == grp_key) || (top_n >= k && __ballot(valid && my_key == far_key))) { tie = true; break; }
        }
        // the push loop (:165-178) in adjacency order.  `pass` was tested against the bound sent with the request, which the
        // farthest key has not exceeded since: every neighbour the test lets through is in it, and the test is made again,
        // against the key as it stands, when its turn comes
        unsigned long long maybe = top_n < k ? fresh : passm;
        PHX_COUNT(5, __popcll(maybe));
        // What the next pop returns is known before the insertions: the closest open entry, or a neighbour of this expansion
        // that is closer.  The memory wave is asked for it NOW, and the insertions run under its round trip.  Not foreseen
        // (the request then follows the pop): anything among equal keys -- a group window, the best neighbour tied with
        // another one or with the closest open entry.
        int want_lane = -1; // the lane of the neighbour foreseen as the next pop: its slot is noted when it goes in
        if (grp_cnt == 0 && !tie) {
            const bool cand = maybe != 0ull && (top_n < k || bk0 < far_key); // the closest neighbour passes the test as it stands (then it is the closest of those that do)
            if (cand && bk0 < nxt_key) {
                if (bl0 >= 0) { want_lane = bl0; early_id = __builtin_amdgcn_readlane(my_id, bl0); early_key = bk0; early_nsame = 1; }
            } else if (nxt_id >= 0 && (!cand || bk0 > nxt_key)) { early_id = nxt_id; early_pos = npos; early_key = nxt_key; early_nsame = nn; }
            if (early_id >= 0) port->post(early_id, layer, lane, top_n >= k ? far_key : 0xffffffffu);
        }
        PHY(9);
        while (maybe) {
            const int src = (int)__builtin_ctzll(maybe);
            maybe &= maybe - 1;
            const unsigned dk = (unsigned)__builtin_amdgcn_readlane((int)my_key, src);
            const int did = __builtin_amdgcn_readlane(my_id, src);
            if (top_n < k) {                                             // :165, :168-174
                T.put(top_n, dk, did);
                if (src == want_lane) early_pos = top_n;
                ++top_n;
                if (top_n == k) far_key = T.max_key();                   // :176-177
            } else if (dk < far_key) {
                int slot, twins;
                T.template locate<false>(far_key, slot, twins);          // the farthest result leaves (:171-174)
                if (twins == 1) {
                    T.put(slot, dk, did);
                    far_key = T.max_key();                               // :176-177: a farthest key of its own
                } else { // (i): one of several equally far results is dropped -- the key stays; (b)
                    const int evicted = T.id_at(slot);
                    T.put(slot, dk, did);
                    const bool hard = ids_matter_everywhere || evicted >= 0 || T.any_open_key(far_key);
                    doubt_hard |= hard;
                    T.mark_key(far_key, kDoubt);
                    if (grp_cnt > 0 && hard) tie = true;
                }
                if (src == want_lane) early_pos = slot;
            } else if (grp_cnt > 0 && dk == far_key) tie = true; // (b): turned away by equality
        }
        PHY(11);
        PH(5);
    }
    PH_FLUSH();
    if (port->pending()) port->wait(); // a request posted ahead of a pop that never came: let it finish (its marks die with the visited set)
    if (grp_cnt > 0 && !tie && !unsafe && !hash_full) { // (c) at the end of the search
        if (T.count_key(grp_key) != grp_cnt) tie = true;
        else if (window_out) *window_out = true;
    }
    // ToArray() for the callers, ascending: rank every entry by counting -- (key, doubtful first, slot) -- through LDS
    uint2 *raw = reinterpret_cast<uint2 *>(L.top);
    wave_sync();
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int sl = lane + 64 * t;
        if (sl < top_n) raw[sl] = make_uint2((unsigned)T.id[t], T.key[t]);
    }
    wave_sync();
    int rank[NS];
    unsigned long long mine[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        rank[t] = 0;
        mine[t] = ((unsigned long long)T.key[t] << 32) | ((T.id[t] & kDoubt) ? 0ull : 0x10000ull) | (unsigned long long)(lane + 64 * t);
    }
    for (int j = 0; j < top_n; ++j) {
        const uint2 e = raw[j];
        const unsigned long long other = ((unsigned long long)e.y << 32) | (((int)e.x & kDoubt) ? 0ull : 0x10000ull) | (unsigned long long)j;
#pragma unroll
        for (int t = 0; t < NS; ++t) rank[t] += other < mine[t] ? 1 : 0;
    }
    wave_sync();
    unsigned first_doubt = 0xffffffffu;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int sl = lane + 64 * t;
        if (sl < top_n) {
            L.top[rank[t]].id = T.id[t] & kIdMask;
            L.top[rank[t]].dist = key2f(T.key[t]);
            if (T.id[t] & kDoubt) first_doubt = min(first_doubt, (unsigned)rank[t]);
        }
    }
    wave_sync();
    top_n_out = top_n;
    first_doubt = wave_min_u32(first_doubt);
    if (first_doubt != 0xffffffffu && (doubt_hard || (int)first_doubt < min(top_n, ordered_prefix))) tie = true; // (i) left unresolved
    // (iii): equal distances next to each other in what the caller consumes in order
    bool eq = false;
    const int upto = min(top_n, ordered_prefix);
    for (int p0 = 0; p0 < upto; p0 += 64) {
        const int pp = p0 + lane;
        if (pp >= 1 && pp < upto) eq = eq || __float_as_uint(L.top[pp].dist) == __float_as_uint(L.top[pp - 1].dist);
    }
    const bool order_tie = __ballot(eq) != 0ull;
    if (order_tie_out) *order_tie_out = order_tie && !tie;
    else if (order_tie) tie = true;
    tie_out = tie;
    return !unsafe && !hash_full;
}

} // namespace hnsw
