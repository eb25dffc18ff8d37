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
// register set s / 64; bit 31 of the id = expanded; bit 30 = doubtful, set only when the list is handed on) and runs the
// reference's own loop on them: pop = wave-wide minimum over the open slots (six fused DPP steps, one readlane), push = the
// farthest entry rewrites itself under the mask of the compare that found it, then a wave-wide maximum; the list is
// sorted ONCE, when the search is over (ranks by counting through LDS), and handed on ascending like the sorted list's.
// Equal distances: the rules of traverse_sorted, stated on keys instead of positions.  (i) the farthest result leaves
// while another entry has its distance (the maximum does not change): the survivors of that distance become doubtful
// (hard unless the one that left and all of them were expanded) -- ONE bit of state: they are the entries of key far_key
// until that key changes; (ii) the popped candidate has an open twin -- read off the next lookup: the closest open key
// still equals the popped one -- a group window opens (members counted by key; closes at the first pop beyond the key
// with all members still present);
// (a), (b), (d) inside a window and (c) at its end as there; (iii) is read off the sorted output.  Which of several
// equal entries a minimum or maximum picks differs from the sorted list (lowest lane here, first position there) -- in
// exactly the situations these rules either prove immaterial or hand to the exact two-heap traversal
// (tests/test_pool_rules_model.py: the rules in Python with twins taken at random, against the oracle's SearchLayer).
// v_writelane_b32: a uniform value into ONE lane of a register.  (No builtin reaches it.  One scalar register per VALU
// instruction on this ISA: the lane select goes through M0, as the compiler's own lowering of the intrinsic does.)  The pool
// no longer writes through it -- see lanes_set below -- tools/pool_probe.hip keeps the forms that did, for comparison.
__device__ __forceinline__ int lane_write(int value, int lane_sel, int old)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(value), "s"(lane_sel) : "m0");
    return old;
}
// dst = mask ? src : dst, per lane, the mask a uniform 64-bit value.  As asm: written as a C++ select the compiler turns it into
// exec-masked moves inside s_and_saveexec / s_cbranch_execz brackets -- per register set, with a second copy of the pool's
// registers kept alive around them.
__device__ __forceinline__ void lanes_set(int &dst, int src, unsigned long long mask)
{
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(dst) : "v"(src), "s"(mask));
}
__device__ __forceinline__ void lanes_set(unsigned &dst, unsigned src, unsigned long long mask)
{
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(dst) : "v"(src), "s"(mask));
}
// a value the optimiser cannot trace back to the array element it came from (a chain of `t == sel ? id[t] : v` selects is
// otherwise folded into ONE indexed load, and the whole pool then lives in scratch memory beside its registers)
__device__ __forceinline__ int opaque_copy(int x)
{
    int v;
    asm volatile("v_mov_b32_e32 %0, %1" : "=v"(v) : "v"(x));
    return v;
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
    // slot = 64 t + lane, uniform.  Every operation is straight-line code: which register set a slot lives in is a uniform
    // MASK (the slot's lane bit under set t, zero under the others: __builtin_amdgcn_inverse_ballot_w64 hands a scalar mask to
    // v_cndmask as it is), never an `if (slot >> 6 == t)` ladder around a v_writelane -- the ladders' merges cost a dozen
    // register copies per insertion in the traversal kernels (tools/pool_probe.hip: 484 -> 352 clocks per replaced entry and
    // 641 -> 463 per pop at NS = 4, alone in a kernel; more inside one).
    __device__ __forceinline__ int id_at(int slot) const
    {
        int sel = id[0];
#pragma unroll
        for (int t = 1; t < NS; ++t) sel = (slot >> 6) == t ? opaque_copy(id[t]) : sel;
        return __builtin_amdgcn_readlane(sel, slot & 63);
    }
    __device__ __forceinline__ void put(int slot, unsigned k0, int i0) // a new, open entry
    {
        unsigned long long m[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) m[t] = (slot >> 6) == t ? 1ull << (slot & 63) : 0ull;
        replace(m, k0, i0);
    }
    __device__ __forceinline__ void mark_expanded(int slot, int idword) // idword: the entry's id word as it reads now
    {
        const unsigned long long bit = 1ull << (slot & 63);
        const int gone = -1, word = idword | (int)0x80000000;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const unsigned long long m = (slot >> 6) == t ? bit : 0ull;
            lanes_set(okey[t], (unsigned)gone, m);
            lanes_set(id[t], word, m);
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
    // the entries of key k0, as one lane mask per register set, and how many there are
    __device__ __forceinline__ void hits(unsigned k0, unsigned long long (&hb)[NS], int &count) const
    {
        count = 0;
#pragma unroll
        for (int t = 0; t < NS; ++t) { hb[t] = __ballot(key[t] == k0); count += (int)__popcll(hb[t]); }
    }
    static __device__ __forceinline__ int lowest(const unsigned long long (&hb)[NS])
    {
        int slot = -1;
#pragma unroll
        for (int t = NS - 1; t >= 0; --t)
            if (hb[t]) slot = 64 * t + (int)__builtin_ctzll(hb[t]);
        return slot;
    }
    // the ONE entry under the masks becomes (k0, i0), open: it rewrites itself, no slot is worked out
    __device__ __forceinline__ void replace(const unsigned long long (&hb)[NS], unsigned k0, int i0)
    {
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            lanes_set(key[t], k0, hb[t]);
            lanes_set(okey[t], k0, hb[t]);
            lanes_set(id[t], i0, hb[t]);
        }
    }
    // the closest open entry: its key (0xffffffff: none), slot (-1) and id word.  Of several equal ones: the lowest lane's, and
    // in that lane the lowest register set's.  (Whether the popped entry has an open twin is read off the NEXT lookup: after
    // the pop is marked expanded, the closest open key equals the popped one exactly then.)
    __device__ __forceinline__ void min_open(unsigned &mk, int &slot, int &eid) const
    {
        unsigned v = okey[0];
#pragma unroll
        for (int t = 1; t < NS; ++t) v = min(v, okey[t]);
        mk = wave_min_u32(v);
        slot = -1; eid = 0;
        if (mk == 0xffffffffu) return;
        const int l = (int)__builtin_ctzll(__ballot(v == mk));
        int tsel = NS - 1, sel = id[NS - 1];
#pragma unroll
        for (int t = NS - 2; t >= 0; --t) {
            const bool e = okey[t] == mk;
            tsel = e ? t : tsel;
            sel = e ? opaque_copy(id[t]) : sel;
        }
        slot = 64 * __builtin_amdgcn_readlane(tsel, l) + l;
        eid = __builtin_amdgcn_readlane(sel, l);
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
    __device__ __forceinline__ bool any_open_key_but(unsigned k0, int slot) const // an open entry of that key in another slot
    {
        unsigned long long m = 0ull;
#pragma unroll
        for (int t = 0; t < NS; ++t) m |= __ballot(okey[t] == k0) & ~((slot >> 6) == t ? 1ull << (slot & 63) : 0ull);
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
    descend<METRIC, true>(rows, row_sn, dim, sb, G, jb, L, lane, best, cur, evals, RL);
    // ---- SearchLayer (GraphNavigator.cs:123-189) ----
    const int layer = jb.search_layer;
    RL.layer(layer, lane);
    constexpr int kDoubt = 0x40000000, kIdMask = 0x3fffffff;
    PoolTop<NS> T;
    T.init();
    int top_n = 0;
    // the traversal's verdicts, bits of ONE scalar word (as bools each is a 64-bit lane mask in two scalar registers, and the
    // loop below ran out of those: two dozen spill moves per accepted neighbour)
    enum : unsigned { kTie = 1u, kUnsafe = 2u, kHashFull = 4u, kFarDoubt = 8u, kDoubtHard = 16u };
    unsigned st = key_unsafe(cur) ? kUnsafe : 0u; // NaN / -0 (see f2key)
    // (the hint left by the previous traversal names a node of ITS layer: the last job's layer-0 search, say, while this one
    //  starts on an upper layer where that node has no list -- the memory wave would prefetch from in front of the pool)
    if (lane == 0) port->m->hint_node = -1;
    T.put(0, f2key(cur), best);                                         // :134, :138
    top_n = 1;
    if (lane == 0) (void)V.first_visit(best);                           // :140
    __builtin_amdgcn_s_waitcnt(0); // (the memory wave's marks follow: this one has landed)
    V.seen += 1;
    unsigned far_key = f2key(cur);                                      // farthestResultDist :135
    const bool ids_matter_everywhere = order_tie_out != nullptr; // an insert's heuristic reads the whole list; a search its first entries
    // kFarDoubt -- (i): the entries at the farthest distance are doubtful (stated on the key: they are the entries of key
    // far_key for as long as any of them is listed, and the key changes when the last one leaves)
    unsigned grp_key = 0u; // the group window of (ii): its distance and its members (0: no window open)
    int grp_cnt = 0;
    int early_id = -1;     // the node whose expansion was requested before its pop (-1: none) ...
    int early_pos = 0;     // ... and the slot it sits in
    unsigned early_key = 0u;
    PH(0);
    while (!(st & (kUnsafe | kTie))) {
        unsigned ck;
        int pos, cid;
        if (early_id >= 0) {
            // the pop was foreseen (below): its slot and key are known (the entry is open and listed: it was the closest open
            // one, or a neighbour closer than that, and only the farthest entry ever leaves -- which it is not while a
            // neighbour passes the test), and the memory wave has been on its expansion since before the last insertions
            pos = early_pos; ck = early_key; cid = early_id;
            early_id = -1;
        } else {
            T.min_open(ck, pos, cid);                                    // :146 closest candidate; none left <=> :147-150 / empty
            if (pos < 0) break;
            port->post(cid & kIdMask, layer, lane, top_n >= k ? far_key : 0xffffffffu);
        }
        if ((st & kFarDoubt) && ck == far_key) { st |= kTie; break; } // a doubtful entry: the reference may be expanding its twin instead
        if (grp_cnt > 0 && ck > grp_key) { // the group window closes: (c) every member still listed?
            if (T.count_key(grp_key) != grp_cnt) { st |= kTie; break; }
            grp_cnt = 0;
            if (window_out) *window_out = true;
        }
        T.mark_expanded(pos, cid);
        RL.put(cid & kIdMask, lane, top_n >= k && grp_cnt == 0 ? far_key : 0xffffffffu);
        // what would be popped next if this expansion brought nothing closer; (ii): an open twin of the popped candidate
        unsigned nxt_key;
        int npos, nid;
        T.min_open(nxt_key, npos, nid);
        const int nxt_id = npos >= 0 ? (nid & kIdMask) : -1;
        if (nxt_key == ck) { // (nxt_key is 0xffffffff when nothing is open; no entry has that key)
            if (grp_cnt == 0) { grp_key = ck; grp_cnt = T.count_key(ck); }
            else if (ck != grp_key) st |= kTie; // (d)
        }
        if (lane == 0) port->m->hint_node = nxt_id; // (a list for the memory wave to prefetch: a node of THIS layer; late or missing, nothing breaks)
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
        // What the next pop returns is known before the insertions: the closest open entry, or a neighbour of this expansion
        // that is closer.  The memory wave is asked for it NOW -- before anything else is made of the answer: the memory wave's
        // service and this stretch are the chain an expansion's time is made of (phase clocks, B = 1 Add: service 4 700 clocks,
        // answer-to-request 1 100, and the 2 000-3 700 clocks of insertions hidden under the service) -- and the checks and the
        // insertions run under its round trip.  Not foreseen (the request then follows the pop): anything among equal keys --
        // a group window, the best neighbour tied with another one or with the closest open entry.  A request posted for a
        // traversal that one of the checks below then abandons is waited for at the end, like any early exit's.
        // `pass` was tested against the bound sent with the request, which the farthest key has not exceeded since: every
        // neighbour the push test lets through is in it, and the test is made again, against the key as it stands, when its
        // turn comes (an answer not served -- a list beyond 64 -- has empty masks)
        unsigned long long maybe = top_n < k ? fresh : passm;
        int want_lane = -1; // the lane of the neighbour foreseen as the next pop: its slot is noted when it goes in
        if (grp_cnt == 0 && !(st & kTie)) {
            const bool cand = maybe != 0ull && (top_n < k || bk0 < far_key); // the closest neighbour passes the test as it stands (then it is the closest of those that do)
            if (cand && bk0 < nxt_key) {
                if (bl0 >= 0) { want_lane = bl0; early_id = __builtin_amdgcn_readlane(my_id, bl0); early_key = bk0; }
            } else if (nxt_id >= 0 && (!cand || bk0 > nxt_key)) { early_id = nxt_id; early_pos = npos; early_key = nxt_key; }
            if (early_id >= 0) port->post(early_id, layer, lane, top_n >= k ? far_key : 0xffffffffu);
        }
        if ((nw & 0xffff) > 64) { st |= kHashFull; break; }
        const int m = (int)__popcll(fresh);
        PH_COUNT(7, 1);
        V.seen += m;
        if (V.crowded()) { st |= kHashFull; break; }
        if (m == 0) continue;
        evals += (unsigned long long)m;
        if (nw & 0x10000) { st |= kUnsafe; break; }
        if (grp_cnt > 0) { // (a), (b)
            const bool valid = ((fresh >> lane) & 1ull) != 0ull;
            if (__ballot(valid && my_key == grp_key) || (top_n >= k && __ballot(valid && my_key == far_key))) { st |= kTie; break; }
        }
        PHX_COUNT(5, __popcll(maybe)); // the push loop (:165-178) in adjacency order
        PHY(9);
        while (maybe) {
            const int src = (int)__builtin_ctzll(maybe);
            maybe &= maybe - 1;
            const unsigned dk = (unsigned)__builtin_amdgcn_readlane((int)my_key, src);
            const int did = __builtin_amdgcn_readlane(my_id, src);
            // ONE place writes the pool (a replace under lane masks, one per register set): with a write per case the compiler
            // keeps a copy of the twelve pool registers per path and moves them back and forth at every merge
            unsigned long long wm[NS];
            bool write = false, refresh = false;
            if (top_n < k) {                                             // :165, :168-174
#pragma unroll
                for (int t = 0; t < NS; ++t) wm[t] = (top_n >> 6) == t ? 1ull << (top_n & 63) : 0ull;
                if (src == want_lane) early_pos = top_n;
                ++top_n;
                write = true;
                refresh = top_n == k;
            } else if (dk < far_key) {
                int twins;
                T.hits(far_key, wm, twins);                              // the farthest result leaves (:171-174): it rewrites itself
                if (twins != 1) { // (i): one of several equally far results is dropped -- the key stays; (b)
                    const int slot = PoolTop<NS>::lowest(wm);
                    const int evicted = T.id_at(slot);
                    const bool hard = ids_matter_everywhere || evicted >= 0 || T.any_open_key_but(far_key, slot);
                    st |= kFarDoubt | (hard ? kDoubtHard : 0u); // the survivors of that distance are doubtful from here on
                    if (grp_cnt > 0 && hard) st |= kTie;
#pragma unroll
                    for (int t = 0; t < NS; ++t) wm[t] = (slot >> 6) == t ? 1ull << (slot & 63) : 0ull;
                }
                if (src == want_lane) early_pos = PoolTop<NS>::lowest(wm);
                write = refresh = true;                                  // :176-177 (with twins the maximum stays what it is)
            } else if (grp_cnt > 0 && dk == far_key) st |= kTie; // (b): turned away by equality
            if (write) {
                T.replace(wm, dk, did);
                if (refresh) {
                    const unsigned was = far_key;
                    far_key = T.max_key();
                    if (far_key != was) st &= ~kFarDoubt; // the doubtful entries were the farthest: all gone
                }
            }
        }
        PHY(11);
        PH(5);
    }
    PH_FLUSH();
    if (port->pending()) port->wait(); // a request posted ahead of a pop that never came: let it finish (its marks die with the visited set)
    if (grp_cnt > 0 && !(st & (kTie | kUnsafe | kHashFull))) { // (c) at the end of the search
        if (T.count_key(grp_key) != grp_cnt) st |= kTie;
        else if (window_out) *window_out = true;
    }
    if (st & kFarDoubt) T.mark_key(far_key, kDoubt); // what the ordering below and the callers' rules read
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
    bool tie = (st & kTie) != 0u;
    if (first_doubt != 0xffffffffu && ((st & kDoubtHard) || (int)first_doubt < min(top_n, ordered_prefix))) tie = true; // (i) left unresolved
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
    return !(st & (kUnsafe | kHashFull));
}

} // namespace hnsw
