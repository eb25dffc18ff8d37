// dk_sorted_top.h -- device code, part of device_kernels.h: SearchLayer on ONE sorted list in registers: SortedTop, traverse_sorted (tie rules: doubts, group windows; no visited set).
#pragma once
#include "dk_search_common.h"

namespace hnsw {

// ---- SearchLayer on ONE sorted list in registers ---------------------------------------------
// The reference keeps two heaps (GraphNavigator.cs:126-127): topCandidates (the k closest seen,
// farthest at the root) and candidates (everything accepted, closest at the root).  An accepted
// element is pushed to both; it leaves topCandidates only when k closer ones exist, and from
// then on its distance exceeds farthestResultDist for good, so popping it from `candidates` can
// only end the loop (:147-150).  Hence the live part of `candidates` is exactly the not yet
// expanded members of topCandidates, and when no two coexisting entries have equal distances
// the whole state is one ascending list of <= k entries with an "expanded" mark:
//   pop closest candidate  = first unmarked entry            (ballot + ctz)
//   push / trim to k       = ranked insertion, last one drops (compare + popcount + lane shift)
//   farthestResultDist     = entry k - 1
// which is straight-line wave-wide code instead of scalar sift loops in LDS (2/3 of the traversal
// time at C2, all of it scalar-issue bound).  Equal distances: a heap removes "the" extreme
// element, so as long as the extreme is unique the SETS in both heaps evolve identically whatever
// the array layout.  The layout shows only when (i) the farthest result is evicted while another
// entry has the same distance, (ii) the closest candidate is popped while another open candidate
// has the same distance, or (iii) equal distances sit next to each other in what the caller
// consumes in order (OrderBy + Take(k), Span.Sort).  (iii) raises `tie` and the caller
// repeats the job with the exact two-heap traversal below; (ii) opens a GROUP WINDOW (below) and raises
// `tie` only if the window cannot show that the order was immaterial.  After (i) the survivor (the reference
// may hold its twin instead -- same distance, other id, possibly still a candidate there) is only
// marked DOUBTFUL: the search goes on, and `tie` is raised if a doubtful entry is popped or is still
// in the list at the end; usually the next few insertions push it out and nothing depended on it.
// A search (OrderBy + Take(k_out) with k_out far below k) goes one step further with (i): when the entry that left AND
// every survivor of its distance had already been expanded, the two lists differ in ONE id of equal distance at the far
// end and in nothing that can still happen -- neither twin is a candidate any more, the farthest distance is the same --
// so such an event is only remembered as an identity doubt, which asks for the exact traversal only if a doubtful entry
// ends inside the ordered prefix the caller reads (never, with k = 128 and k_out = 10) and does not fail a group window.
// An insert reads all k entries (the heuristic's candidates): every doubt stays a doubt there.
// Equal distances elsewhere in the list are harmless.  Position p lives in lane p & 63 of register
// set p >> 6; id bit 31 = expanded, bit 30 = doubtful (node ids stay below 2^30).
//
// The group window of (ii).  Open candidates A, B, .. of one distance d, one of them popped: the reference pops them in
// an order only its heap knows, and between two of them it expands whatever closer candidates the first one's
// expansion turned up.  Whatever that order: as long as every member is still in the list, farthestResultDist >= d,
// so every node closer than d that any expansion turns up is accepted (:165) and expanded before anything farther than
// d -- the nodes expanded until the first pop beyond d are the members plus everything closer than d that is
// reachable from them through such nodes, a closure that does not depend on the order, and so are the nodes
// evaluated (their unvisited neighbours) and the list afterwards (the k closest of what there was and what was
// evaluated; a node turned away in one order is pushed out in the other).  A member can only leave the list when k
// entries rank before it, and the entries closer than d at any moment of any order are a subset of those there when
// the window closes in THIS order -- so if all members are still listed then, none was evicted in any order, and the
// state at that point (list, marks, visited set, evaluation count) is the reference's whichever way its heap went.
// The window therefore asks for the exact traversal only when (a) an evaluated neighbour has distance d itself (a
// member the other order might have turned away), (b) the list's far end meets equal distances while it is open (an
// entry turned away or evicted by equality: which twin stays depends on the order of arrival), (c) a member is missing
// when it closes, or (d) a second group opens inside it.  Of the 100 windows a 65 536-query launch at C2 opens, 86
// close cleanly (the others sit at the far end of the list, where the members themselves are evicted); with the ten
// or so unresolved cases of (i) that leaves 24 exact traversals per launch where there were 75 (15-20 with the identity
// doubts above) -- which matters because
// an exact traversal takes three times as long as a sorted one and a launch ends with its last job (17-35 % of a
// 12 500-query launch at 10M was the wait for such jobs, measured).
__device__ __forceinline__ int dpp_wave_shr1(int carry_in, int v)
{
    return __builtin_amdgcn_update_dpp(carry_in, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); // lane 0 keeps carry_in
}
template <int NS>
struct SortedTop {
    unsigned key[NS];
    int id[NS];
    __device__ __forceinline__ HEnt at(int p) const // uniform p
    {
        HEnt e{__builtin_amdgcn_readlane(id[0], p & 63), (unsigned)__builtin_amdgcn_readlane((int)key[0], p & 63)};
#pragma unroll
        for (int t = 1; t < NS; ++t) {
            const int wi = __builtin_amdgcn_readlane(id[t], p & 63);
            const unsigned wk = (unsigned)__builtin_amdgcn_readlane((int)key[t], p & 63);
            if ((p >> 6) == t) { e.id = wi; e.key = wk; }
        }
        return e;
    }
    __device__ __forceinline__ unsigned key_at(int p) const
    {
        unsigned v = (unsigned)__builtin_amdgcn_readlane((int)key[0], p & 63);
#pragma unroll
        for (int t = 1; t < NS; ++t) {
            const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)key[t], p & 63);
            if ((p >> 6) == t) v = w;
        }
        return v;
    }
    // first entry not yet expanded, or -1
    __device__ __forceinline__ int first_open(int count, int lane) const
    {
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            if (64 * t >= count) break;
            const unsigned long long m = __ballot(lane + 64 * t < count && id[t] >= 0);
            if (m) return 64 * t + (int)__builtin_ctzll(m);
        }
        return -1;
    }
    __device__ __forceinline__ void mark(int p, int lane, int bit = (int)0x80000000)
    {
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if ((p >> 6) == t && lane == (p & 63)) id[t] |= bit;
    }
    __device__ __forceinline__ void mark_key(unsigned k0, int count, int lane, int bit) // every entry of that key
    {
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if (lane + 64 * t < count && key[t] == k0) id[t] |= bit;
    }
    __device__ __forceinline__ bool any_flagged(int count, int lane, int bit) const // uniform result
    {
        bool f = false;
#pragma unroll
        for (int t = 0; t < NS; ++t) f |= lane + 64 * t < count && (id[t] & bit) != 0;
        return __ballot(f) != 0ull;
    }
    // ranked insertion of (xk, xid), before any entries of equal key; beyond k entries the last one drops
    __device__ __forceinline__ void insert(unsigned xk, int xid, int &count, int k, int lane)
    {
        int r = 0;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            if (64 * t >= count) break;
            r += (int)__popcll(__ballot(lane + 64 * t < count && key[t] < xk));
        }
#pragma unroll
        for (int t = NS - 1; t >= 0; --t) {
            if (64 * t > count || 64 * (t + 1) <= r) continue; // nothing at or after r in this set
            int ck = 0, ci = 0;
            if (t > 0) { ck = __builtin_amdgcn_readlane((int)key[t - 1], 63); ci = __builtin_amdgcn_readlane(id[t - 1], 63); }
            const int sk = dpp_wave_shr1(ck, (int)key[t]);
            const int si = dpp_wave_shr1(ci, id[t]);
            const int p = lane + 64 * t;
            key[t] = p == r ? xk : p > r ? (unsigned)sk : key[t];
            id[t] = p == r ? xid : p > r ? si : id[t];
        }
        if (count < k) ++count;
    }
    // Several insertions at once: the candidates of the lanes in `pass` (my_key, my_id; at least one).  What `insert`
    // called once per candidate in lane order leaves behind is the k smallest of the union -- a candidate that a
    // tighter farthest distance would have turned away ends beyond position k here and drops just the same -- with
    // a new entry before old entries of equal key and a later lane's before an earlier lane's.  So the final
    // position of every entry follows from counting: an old entry moves up by the new keys <= its own, a new one
    // lands at (old keys < its own) + (new ones that go before it).  The entries are scattered to `lds`
    // (k + 1 slots: ids with their mark bits, keys) at those positions and read back: one compare per register set
    // and candidate instead of insert's shift of the whole list.
    // boundary_tie: entries were dropped and the first one dropped has the key of the last one kept (the
    // reference's list may hold that twin instead: the caller marks the survivors DOUBTFUL, rule (i)).
    __device__ __forceinline__ void merge(unsigned long long pass, unsigned my_key, int my_id, int &count, int k, int lane,
                                          uint2 *lds, unsigned &last_key, bool &boundary_tie, bool &dropped_expanded)
    {
        int shift[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) shift[t] = 0;
        int rank_old = 0, rank_new = 0;
        for (unsigned long long mm = pass; mm; mm &= mm - 1) {
            const int src = __builtin_ctzll(mm);
            const unsigned xk = (unsigned)__builtin_amdgcn_readlane((int)my_key, src);
            int r = 0;
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                if (64 * t >= count) break;
                const bool have = lane + 64 * t < count;
                const bool lt = have && key[t] < xk;
                r += (int)__popcll(__ballot(lt));
                shift[t] += (have && !lt) ? 1 : 0;
            }
            if (lane == src) rank_old = r;
            rank_new += (xk < my_key || (xk == my_key && src > lane)) ? 1 : 0;
        }
        wave_lds_sync(); // (a list prefetched just before the merge stays in flight)
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int p = lane + 64 * t;
            if (p < count && p + shift[t] <= k) lds[p + shift[t]] = make_uint2((unsigned)id[t], key[t]);
        }
        if ((pass >> lane) & 1ull) {
            const int np = rank_old + rank_new;
            if (np <= k) lds[np] = make_uint2((unsigned)my_id, my_key);
        }
        wave_lds_sync();
        const int total = count + (int)__popcll(pass);
        count = min(k, total);
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int p = lane + 64 * t;
            if (p < count) { const uint2 e = lds[p]; id[t] = (int)e.x; key[t] = e.y; }
        }
        last_key = lds[count - 1].y;
        boundary_tie = total > k && lds[k].y == last_key;
        dropped_expanded = boundary_tie && (int)lds[k].x < 0; // the twin that left had been expanded (bit 31 of its id word)
        wave_lds_sync();
    }
    __device__ __forceinline__ bool contains_id(int node, int count, int lane) const // is that node listed? (uniform)
    {
        unsigned long long m = 0ull;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            if (64 * t >= count) break;
            m |= __ballot(lane + 64 * t < count && (id[t] & 0x3fffffff) == node);
        }
        return m != 0ull;
    }
    __device__ __forceinline__ bool any_open_key(unsigned k0, int count, int lane) const // an entry of that key not yet expanded? (uniform)
    {
        bool o = false;
#pragma unroll
        for (int t = 0; t < NS; ++t) o |= lane + 64 * t < count && key[t] == k0 && id[t] >= 0;
        return __ballot(o) != 0ull;
    }
    __device__ __forceinline__ int first_flagged(int count, int lane, int bit) const // position of the first entry with that bit, or count
    {
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            if (64 * t >= count) break;
            const unsigned long long m = __ballot(lane + 64 * t < count && (id[t] & bit) != 0);
            if (m) return 64 * t + (int)__builtin_ctzll(m);
        }
        return count;
    }
    __device__ __forceinline__ int count_key(unsigned k0, int count, int lane) const // entries of that key (uniform result)
    {
        int c = 0;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            if (64 * t >= count) break;
            c += (int)__popcll(__ballot(lane + 64 * t < count && key[t] == k0));
        }
        return c;
    }
    // any p in [1, upto) with key[p] == key[p - 1]?  (uniform result)
    __device__ __forceinline__ bool adjacent_equal(int upto, int lane) const
    {
        bool eq = false;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            if (64 * t >= upto) break;
            int ck = 0;
            if (t > 0) ck = __builtin_amdgcn_readlane((int)key[t - 1], 63);
            const unsigned prev = (unsigned)dpp_wave_shr1(ck, (int)key[t]);
            const int p = lane + 64 * t;
            eq |= p >= 1 && p < upto && key[t] == prev;
        }
        return __ballot(eq) != 0ull;
    }
};

// Returns false on a NaN / -0 distance (exact host re-run); `tie` asks for the exact two-heap
// traversal.  Result: L.top[0..top_n) ascending by distance.  The query must be staged in L.qs.
template <int METRIC, int NS, bool HASHED, bool LEAN = false>
__device__ __forceinline__ bool traverse_sorted(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, double sb,
                                                const GraphView &G, const SearchJob jb, int k, int ordered_prefix, VisitedSet<HASHED> &V,
                                                const SearchLds &L, int lane, int &top_n_out, bool &tie_out, unsigned long long &evals,
                                                int oflags, ReadLog &RL, bool *order_tie_out = nullptr, bool *window_out = nullptr)
{
    int *nbuf = L.nbuf;
    float *dbuf = L.dbuf;
    const float *qs = L.qs;
    PH_DECL();
    int best;
    float cur;
    descend<METRIC>(rows, row_sn, dim, sb, G, jb, L, lane, best, cur, evals, RL);
    // ---- SearchLayer (GraphNavigator.cs:123-189) ----
    const int layer = jb.search_layer;
    RL.layer(layer, lane);
    SortedTop<NS> T;
#pragma unroll
    for (int t = 0; t < NS; ++t) { T.key[t] = 0u; T.id[t] = 0; }
    int top_n = 0;
    bool unsafe = key_unsafe(cur); // NaN / -0 (see f2key)
    bool tie = false, hash_full = false;
    T.insert(f2key(cur), best, top_n, k, lane);                      // :134, :138
    // oflags bit 3 (KnnQuery launches on graphs whose visited sets are hash tables): NO visited set at all.  Such launches fetch
    // the rows of every listed neighbour anyway (overlapped form), and what the set is for follows from the list itself: a
    // neighbour seen before is either still listed -- found by its id -- or it was turned away or pushed out at a farthest key
    // that has only shrunk since, and the push test (:165) turns it away again.  One CAS per evaluation was as much HBM traffic
    // as a 128-byte int8 record, and the 64-KB table was cleared after every job.
    const bool novis = LEAN || (oflags & 8) != 0; // (LEAN: launched with flags 9 only, see kFormLean)
    if (!novis) {
        if (lane == 0) (void)V.first_visit(best);                       // :140
        V.seen += 1;
    }
    unsigned far_key = f2key(cur);                                   // farthestResultDist :135
    int pre_id = -1, pre_a = 0, pre_b = 0; // speculative prefetch of the next expansion's list (see traverse)
    const int lstride = layer == 0 ? G.stride0 : G.strideU;
    PH(0);
    constexpr int kDoubt = 0x40000000, kIdMask = 0x3fffffff;
    const bool ids_matter_everywhere = order_tie_out != nullptr; // an insert's heuristic reads the whole list; a search its first entries
    bool doubt_hard = false; // some doubt of (i) was more than one id of equal distance among expanded entries
    unsigned nxt_key = 0xffffffffu; // distance of the closest open entry once the current one is marked
    unsigned grp_key = 0u; // the group window of (ii): its distance and its members (0: no window open)
    int grp_cnt = 0;
    while (!unsafe && !tie) {
        const int pos = T.first_open(top_n, lane); // :146 closest candidate; none left <=> :147-150 / empty
        if (pos < 0) break;
        const HEnt c = T.at(pos);
        if (c.id & kDoubt) { tie = true; break; } // the reference may be expanding its twin instead
        if (grp_cnt > 0 && c.key > grp_key) { // the group window closes: (c) every member still listed?
            if (T.count_key(grp_key, top_n, lane) != grp_cnt) { tie = true; break; }
            grp_cnt = 0;
            if (window_out) *window_out = true;
        }
        T.mark(pos, lane);
        // inside a window the farthest distance at this pop depends on the order: no bound is logged (the reader's
        // validation then treats every change of the list as visible)
        RL.put(c.id & kIdMask, lane, top_n >= k && grp_cnt == 0 ? far_key : 0xffffffffu);
        PH(1);
        int n, nb_a = 0, nb_b = 0;
        if (c.id == pre_id) {
            n = __builtin_amdgcn_readlane(pre_a, 0);
            nb_a = __shfl(pre_a, (lane + 1) & 63, 64);
            const int w64 = __builtin_amdgcn_readlane(pre_b, 0);
            if (lane == 63) nb_a = w64;
            nb_b = __shfl(pre_b, (lane + 1) & 63, 64);
        } else {
            const int *l = G.list(c.id, layer);
            n = __builtin_amdgcn_readfirstlane(l[0]);
            if (lane < n) nb_a = l[1 + lane];
            if (lane + 64 < n) nb_b = l[65 + lane];
        }
        PH_COUNT(6, c.id == pre_id);
        PH_COUNT(7, 1);
        int m = 0;
        wave_sync();
        PH(2);
        // candidate distances and ids of this expansion, one per lane, in adjacency order
        bool have = false;     // this lane holds an unvisited neighbour
        float lane_d = 0.0f;
        int lane_id = 0;
        if (LEAN && n > 64) { hash_full = true; break; } // (never: the host asks for this form only where no list is longer)
        const bool overlapped = LEAN || ((oflags & 1) != 0 && n <= 64); // oflags bit 0: rows requested with the visited atomics
        if (overlapped) {
            // Latency-bound launch (fewer jobs than resident waves): the rows of ALL listed neighbours
            // are fetched together with the visited atomics instead of after them -- one dependent
            // round trip less per expansion; rows of neighbours that turn out visited are wasted
            // bandwidth, of which such a launch has plenty.  Evaluations counted: the unvisited ones.
            const bool in = lane < n;
            if (in) nbuf[lane] = nb_a;
            wave_sync();
            unsigned old = 0u;
            const unsigned bit = 1u << (nb_a & 31);
            unsigned hpos = 0u;
            if constexpr (HASHED) { // first probe of the id table; a collision is followed up after the rows
                hpos = ((unsigned)nb_a * 2654435761u) & V.tab_mask;
                if (in && !novis) old = (unsigned)atomicCAS(&V.tab[hpos], -1, nb_a);
            } else if (in && !novis) old = atomicOr(&V.bits[nb_a >> 5], bit); // :181, in flight with the row loads below
            pre_id = -1;
            {
                const int nxt = T.first_open(top_n, lane);
                nxt_key = 0xffffffffu;
                if (nxt >= 0) {
                    const HEnt e = T.at(nxt);
                    nxt_key = e.key;
                    if (e.key == c.key) { // (ii)
                        if (grp_cnt == 0) { grp_key = c.key; grp_cnt = T.count_key(c.key, top_n, lane); }
                        else if (c.key != grp_key) tie = true; // (d)
                    }
                    pre_id = e.id & kIdMask;
                    const int *pl = G.list(pre_id, layer);
                    pre_a = lane < lstride ? pl[lane] : 0;
                    pre_b = lane + 64 < lstride ? pl[lane + 64] : 0;
                }
            }
            PHX(0);
            if (n > 0) measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, n, lane); // :163 (and the visited ones)
            wave_sync();
            PHX(1);
            lane_d = in ? dbuf[lane] : 0.0f;
            lane_id = nb_a;
            if (novis) {
                {
                    // every listed neighbour counts as new -- except the ones that are in the list: only a key that could pass
                    // the push test or meet the farthest key matters to anything below, so only those are looked up
                    have = in;
                    const unsigned kq = f2key(lane_d);
                    unsigned long long look = __ballot(in && (top_n < k || kq <= far_key));
                    PHX_COUNT(3, __popcll(look));
                    unsigned long long listed = 0ull;
                    for (unsigned long long mm = look; mm; mm &= mm - 1) {
                        const int sl = (int)__builtin_ctzll(mm);
                        if (T.contains_id(__builtin_amdgcn_readlane(nb_a, sl), top_n, lane)) listed |= 1ull << sl;
                    }
                    if ((listed >> lane) & 1ull) have = false;
                }
            } else if constexpr (HASHED) {
                {
                have = in && (int)old == -1;
                if (in && (int)old != -1 && (int)old != nb_a) { // slot taken by another id: probe on (VisitedSet::first_visit)
                    for (unsigned probes = 0; probes <= V.tab_mask; ++probes) {
                        hpos = (hpos + 1) & V.tab_mask;
                        const int o2 = atomicCAS(&V.tab[hpos], -1, nb_a);
                        if (o2 == -1) { have = true; break; }
                        if (o2 == nb_a) break;
                    }
                }
                }
            } else have = in && (old & bit) == 0u;
            const unsigned long long mask = __ballot(have);
            m = __popcll(mask);
            if (!novis) {
                V.seen += m;
                if (V.crowded()) { hash_full = true; break; }
            }
            PHX(2);
            PHX_COUNT(4, m);
            if (m == 0) continue;
            evals += (unsigned long long)m;
        } else {
        if (novis) { hash_full = true; break; } // (a list of more than 64 entries: the host does not ask for this mode on such a graph)
        for (int base = 0; base < n; base += 64) { // :158-161 keep only unvisited, in list order
            const int i = base + lane;
            bool fresh = false;
            const int nb = base == 0 ? nb_a : nb_b;
            if (i < n) fresh = V.first_visit(nb); // :181 (lists hold no duplicates)
            const unsigned long long mask = __ballot(fresh);
            const int posn = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            if (fresh) nbuf[m + posn] = nb;
            m += __popcll(mask);
        }
        PH(3);
        pre_id = -1;
        {
            const int nxt = T.first_open(top_n, lane);
            nxt_key = 0xffffffffu;
            if (nxt >= 0) {
                const HEnt e = T.at(nxt);
                nxt_key = e.key;
                if (e.key == c.key) { // (ii): which of the two the reference pops first is a matter of heap layout
                    if (grp_cnt == 0) { grp_key = c.key; grp_cnt = T.count_key(c.key, top_n, lane); }
                    else if (c.key != grp_key) tie = true; // (d)
                }
                pre_id = e.id & kIdMask;
                const int *pl = G.list(pre_id, layer);
                pre_a = lane < lstride ? pl[lane] : 0;
                pre_b = lane + 64 < lstride ? pl[lane + 64] : 0;
            }
        }
        wave_sync();
        if (m == 0) continue;
        V.seen += m;
        if (V.crowded()) { hash_full = true; break; } // the id table is filling up: host traversal
        measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, m, lane); // :163
        wave_sync();
        PH(4);
        evals += (unsigned long long)m;
        }
        // the push loop (:165-178) in adjacency order; farthest never grows once the list is full,
        // so only the lanes passing the test now can pass it later: they are replayed one by one
        const int rounds = overlapped ? 1 : (m + 63) / 64;
        for (int r = 0; r < rounds && !unsafe; ++r) {
            const int i = r * 64 + lane;
            const bool valid = overlapped ? have : i < m;
            const float my_d = overlapped ? lane_d : (i < m ? dbuf[i] : 0.0f);
            const int my_id = overlapped ? lane_id : (i < m ? nbuf[i] : 0);
            const unsigned my_key = f2key(my_d);
            if (__ballot(valid && key_unsafe(my_d))) { unsafe = true; break; }
            if (grp_cnt > 0 && __ballot(valid && (my_key == grp_key || (top_n >= k && my_key == far_key)))) { tie = true; break; } // (a), (b)
            unsigned long long maybe = __ballot(valid && (top_n < k || my_key < far_key));
            PHX_COUNT(5, __popcll(maybe));
            PHY(8);
            if (rounds == 1 && maybe) {
                // What the next pop returns is known before the insertions: the closest open entry, or a neighbour of this
                // expansion that is closer.  In the second case the list prefetched above is the wrong one: request the
                // right one now, and the insertions run under its round trip.  (A guess, like every prefetch: the pop decides.)
                unsigned bk = 0xffffffffu;
                int bl = 0;
                for (unsigned long long mm = maybe; mm; mm &= mm - 1) {
                    const int sl = __builtin_ctzll(mm);
                    const unsigned kk = (unsigned)__builtin_amdgcn_readlane((int)my_key, sl);
                    if (kk < bk) { bk = kk; bl = sl; }
                }
                if (bk < nxt_key) {
                    pre_id = __builtin_amdgcn_readlane(my_id, bl);
                    const int *pl = G.list(pre_id, layer);
                    pre_a = lane < lstride ? pl[lane] : 0;
                    pre_b = lane + 64 < lstride ? pl[lane + 64] : 0;
                }
            }
            PHY(9);
#ifndef HNSW_NO_BATCH_MERGE
            if (maybe & (maybe - 1)) { // two or more: one counting merge instead of as many list shifts
                unsigned last = 0u;
                bool boundary_tie = false, dropped_expanded = false;
                T.merge(maybe, my_key, my_id, top_n, k, lane, reinterpret_cast<uint2 *>(L.top), last, boundary_tie, dropped_expanded);
                if (top_n == k) {
                    if (boundary_tie) { // (i); (b)
                        const bool hard = ids_matter_everywhere || !dropped_expanded || T.any_open_key(last, top_n, lane);
                        doubt_hard |= hard;
                        T.mark_key(last, top_n, lane, kDoubt);
                        if (grp_cnt > 0 && hard) tie = true;
                    }
                    far_key = last;                                          // :176-177
                }
                maybe = 0ull;
            }
#endif
            PHY(10);
            while (maybe) {
                const int src = __builtin_ctzll(maybe);
                maybe &= maybe - 1;
                const unsigned dk = (unsigned)__builtin_amdgcn_readlane((int)my_key, src);
                if (top_n < k || dk < far_key) { // :165
                    const bool evicts = top_n == k;
                    const bool last_expanded = evicts && T.at(k - 1).id < 0; // the entry this insertion pushes out
                    T.insert(dk, __builtin_amdgcn_readlane(my_id, src), top_n, k, lane); // :168-174
                    if (top_n == k) {
                        const unsigned nf = T.key_at(k - 1);                             // :176-177
                        if (evicts && nf == far_key) { // (i): one of several equally far results was dropped; (b)
                            const bool hard = ids_matter_everywhere || !last_expanded || T.any_open_key(nf, top_n, lane);
                            doubt_hard |= hard;
                            T.mark_key(nf, top_n, lane, kDoubt);
                            if (grp_cnt > 0 && hard) tie = true;
                        }
                        far_key = nf;
                    }
                } else if (grp_cnt > 0 && dk == far_key) tie = true; // (b): turned away by equality
            }
            PHY(11);
        }
        PH(5);
    }
    PH_FLUSH();
    if (grp_cnt > 0 && !tie && !unsafe && !hash_full) { // (c) at the end of the search
        if (T.count_key(grp_key, top_n, lane) != grp_cnt) tie = true;
        else if (window_out) *window_out = true;
    }
    // ToArray() for the callers: with distinct distances any order-insensitive consumer (OrderBy,
    // Span.Sort) sees the same thing; ascending order is also what they would produce
    wave_sync();
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int p = lane + 64 * t;
        if (p < top_n) { L.top[p].id = T.id[t] & kIdMask; L.top[p].dist = key2f(T.key[t]); }
    }
    wave_sync();
    top_n_out = top_n;
    if (T.any_flagged(top_n, lane, kDoubt) && (doubt_hard || T.first_flagged(top_n, lane, kDoubt) < min(top_n, ordered_prefix)))
        tie = true;                                                      // (i) left unresolved
    // (iii): the SET is the reference's, only its order among equal distances is open.  A caller that can tell
    // whether that order shows in what it makes of the list asks for this case separately (insert_job).
    const bool order_tie = T.adjacent_equal(min(top_n, ordered_prefix), lane);
    if (order_tie_out) *order_tie_out = order_tie && !tie;
    else if (order_tie) tie = true;
    tie_out = tie;
    return !unsafe && !hash_full;
}

} // namespace hnsw
