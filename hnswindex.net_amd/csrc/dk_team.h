// dk_team.h -- device code, part of device_kernels.h: the latency variants' second wave: wave-wide min / max, the LDS mailbox, memory_wave.
#pragma once
#include "dk_search_common.h"

namespace hnsw {

// wave-wide minimum / maximum: four DPP steps inside the rows of 16 lanes, then the four rows' results
// (v_min / v_max with the DPP operand fused, written out: the compiler keeps a v_mov_dpp and the hazard nops apart from the
// operation.  A DPP operand needs two wait states after the VALU write of its register: s_nop 1.  Rows are the wave's
// groups of 16 lanes; row_bcast:15 / :31 carry a row's result into the next row / the upper half, so lane 63 ends up with
// the whole wave's.)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) // uniform result
{
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 0"
                 : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 0"
                 : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long lds_uniform_u64(const unsigned long long *p) // a word every lane reads alike, as two scalars
{
    const unsigned long long v = *p;
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}
// ---- the latency variants' second wave ------------------------------------------------------------------
// One wavefront issues at most one instruction every four clocks, and a traversal is a chain of expansions: in a launch
// that does not fill the chip (B = 1 Add, a round of the exact window, a small query call) the chain's length IS the
// launch, and the phase clocks of such a launch show an expansion of 11 000-12 000 clocks of which the memory round trip
// is 1 800 (tools/latency_probe.hip) -- the rest is one wave's instruction stream: list, visited atomics, 64 loads and
// 128 multiply-adds, then the ranked insertions.  The chip has SIMDs to spare in such a launch, so the latency variants
// run a job on TWO waves of one block with roles of their own:
//   * the LOGIC wave (wave 0) is the traversal as everywhere else: the sorted list, the pops, the tie rules, the
//     insertions, the read log, the heuristic;
//   * the MEMORY wave (wave 1) serves requests "expand node v on layer l": out-edge list, visited atomics, the rows of
//     all listed neighbours (overlapped form), their distances (two lanes per row) -- and answers with ids, distances
//     and the mask of first visits in the block's LDS mailbox.
// What the NEXT pop returns is known before the insertions -- the closest open entry, or a neighbour of this expansion
// that is closer (see the guess below) -- so the logic wave posts the next request BEFORE it merges, and the merge runs
// under the memory wave's round trip.  The prediction is checked when the pop actually happens; a mismatch (never
// observed: equal keys are not predicted) or any early exit abandons the traversal's state as a tie would, which clears
// the visited set the early request has touched.  The waves meet only through LDS words (release / acquire at
// workgroup scope, in-order LDS): never at a barrier.
struct TeamMail {
    int req_seq, req_node, req_layer;       // written by the logic wave; node < 0: the launch is over
    unsigned req_far;                       // ... and an upper bound of the farthest result's key while this request is served (0xffffffff: none)
    // the answer's header, two 16-byte reads for the logic wave:
    int rsp_seq;                            // written last by the memory wave
    int n;                                  // length of the list (> 64: not served); bit 16: a first-visited neighbour's distance is NaN / -0
    unsigned best_key;                      // the smallest key among the neighbours in `pass` (0xffffffff: none) ...
    int best_lane;                          // ... the first lane holding it, bit 31 set if another one holds it too
    unsigned long long fresh;               // bit i: neighbour i had not been visited
    unsigned long long pass;                // ... and its key is below req_far (a superset of what the push test lets through: the bound only shrinks)
    double sb;                              // cosine: sqrt-norm of the job's vector
    int hint_node, pad0;                    // the logic wave's guess at the NEXT node (its closest open entry; -1: none): a list to prefetch, no more
    int ids[64];                            // the listed neighbours, in list order
    float dist[64];                         // distances to the job's vector (staged in L.qs by the logic wave); on answer: their KEYS (f2key), as bits
};
static_assert(offsetof(TeamMail, rsp_seq) == 16 && offsetof(TeamMail, fresh) == 32 && offsetof(TeamMail, ids) % 16 == 0, "TeamMail layout");
struct TeamPort { // the logic wave's end
    TeamMail *m;
    int sent, got;

    __device__ __forceinline__ void post(int node, int layer, int lane, unsigned far = 0xffffffffu)
    {
        ++sent;
        if (lane == 0) {
            m->req_node = node;
            m->req_layer = layer;
            m->req_far = far;
            __hip_atomic_store(&m->req_seq, sent, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __device__ __forceinline__ bool pending() const { return sent != got; }
    __device__ __forceinline__ void wait()
    {
        while (__hip_atomic_load(&m->rsp_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != sent) __builtin_amdgcn_s_sleep(1);
        got = sent;
    }
};

// The memory wave's loop (until a request names node -1).  V: the block's visited set (the logic wave clears it between
// jobs and counts its entries; this wave only marks).
template <int METRIC, bool HASHED>
__device__ __forceinline__ void memory_wave(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, const GraphView &G,
                                            VisitedSet<HASHED> &V, const float *qs, TeamMail *mail, int lane)
{
#ifdef EXP_PHASE_CLOCKS
    long long mw_t = __builtin_readcyclecounter(), mw_acc[4] = {0, 0, 0, 0};
#define MW_PH(i) do { const long long mw_n = __builtin_readcyclecounter(); mw_acc[i] += mw_n - mw_t; mw_t = mw_n; } while (0)
#define MW_FLUSH() do { if (lane == 0) for (int mw_i = 0; mw_i < 4; ++mw_i) atomicAdd(&g_phase_x[12 + mw_i], (unsigned long long)mw_acc[mw_i]); } while (0)
#else
#define MW_PH(i) do {} while (0)
#define MW_FLUSH() do {} while (0)
#endif
    // Two lists requested ahead of their node's expansion (a list is a dependent HBM round trip of its own, 1 900 clocks in
    // front of the rows'): the logic wave's hint -- its closest open entry, the next pop unless this expansion finds something
    // closer -- while the rows are in flight, and the closest neighbour passing the push test as soon as the distances are
    // known -- the next pop in the other case.  [count, e0 .. e63] in one register per lane plus the 64th entry.
    int ha_node = -1, ha_layer = 0, ha_v = 0, ha_w = 0; // the hint's list
    int hc_node = -1, hc_layer = 0, hc_v = 0, hc_w = 0; // the closest neighbour's
    for (int seq = 1;; ++seq) {
        while (__hip_atomic_load(&mail->req_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) __builtin_amdgcn_s_sleep(1);
        MW_PH(0);
        const int node = __builtin_amdgcn_readfirstlane(mail->req_node);
        if (node < 0) { MW_FLUSH(); return; }
        const int layer = __builtin_amdgcn_readfirstlane(mail->req_layer);
        int n, nb = 0;
        if ((node == ha_node && layer == ha_layer) || (node == hc_node && layer == hc_layer)) {
            const bool a = node == ha_node && layer == ha_layer;
            const int v = a ? ha_v : hc_v, w = a ? ha_w : hc_w;
            n = __builtin_amdgcn_readlane(v, 0);
            nb = __shfl(v, (lane + 1) & 63, 64);
            if (lane == 63) nb = __builtin_amdgcn_readlane(w, 0);
        } else {
            const int *l = G.list(node, layer);
            n = __builtin_amdgcn_readfirstlane(l[0]);
            if (lane < n && lane < 64) nb = l[1 + lane];
        }
        if (n > 64) { // (the host never launches this variant on such a graph)
            if (lane == 0) { mail->n = n; mail->fresh = 0ull; mail->pass = 0ull; __hip_atomic_store(&mail->rsp_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
            continue;
        }
        const bool in = lane < n;
        if (in) mail->ids[lane] = nb;
        wave_lds_sync();
        MW_PH(1);
        {   // the hint's list, in flight with the marks and the rows
            const int hint = __builtin_amdgcn_readfirstlane(mail->hint_node);
            if (hint >= 0 && !(hint == ha_node && layer == ha_layer)) {
                const int *pl = G.list(hint, layer);
                const int lstride = layer == 0 ? G.stride0 : G.strideU;
                ha_node = hint; ha_layer = layer;
                ha_v = lane < lstride ? pl[lane] : 0;
                ha_w = 64 < lstride ? pl[64] : 0;
            }
        }
        // visited marks (GraphNavigator.cs:181), in flight with the row loads
        unsigned old = 0u;
        const unsigned bit = 1u << (nb & 31);
        unsigned hpos = 0u;
        if constexpr (HASHED) {
            hpos = ((unsigned)nb * 2654435761u) & V.tab_mask;
            if (in) old = (unsigned)atomicCAS(&V.tab[hpos], -1, nb);
        } else if (in) old = atomicOr(&V.bits[nb >> 5], bit);
        if (n > 0) measure_all<METRIC, true>(rows, row_sn, dim, qs, mail->sb, mail->ids, mail->dist, n, lane); // :163 (and the visited ones)
        bool have;
        if constexpr (HASHED) {
            have = in && (int)old == -1;
            if (in && (int)old != -1 && (int)old != nb) { // slot taken by another id: probe on (VisitedSet::first_visit)
                for (unsigned probes = 0; probes <= V.tab_mask; ++probes) {
                    hpos = (hpos + 1) & V.tab_mask;
                    const int o2 = atomicCAS(&V.tab[hpos], -1, nb);
                    if (o2 == -1) { have = true; break; }
                    if (o2 == nb) break;
                }
            }
        } else have = in && (old & bit) == 0u;
        const unsigned long long mask = __ballot(have);
        MW_PH(2);
        // what the logic wave would compute first of all, done here (this wave has the slack): keys, the push test against
        // the bound that came with the request, the closest neighbour passing it
        wave_lds_sync();
        const float d = in ? mail->dist[lane] : 0.0f;
        const unsigned key = f2key(d);
        const unsigned far = (unsigned)__builtin_amdgcn_readfirstlane((int)mail->req_far);
        const unsigned long long odd = __ballot(have && key_unsafe(d));
        const unsigned long long pass = __ballot(key < far) & mask;
        const unsigned bk = wave_min_u32(((pass >> lane) & 1ull) != 0ull ? key : 0xffffffffu);
        const unsigned long long bm = __ballot(key == bk) & pass;
        wave_lds_sync();
        if (in) mail->dist[lane] = __uint_as_float(key);
        if (lane == 0) {
            mail->n = n | (odd != 0ull ? 0x10000 : 0); mail->fresh = mask; mail->pass = pass;
            mail->best_key = bk;
            mail->best_lane = bm ? ((int)__builtin_ctzll(bm) | ((bm & (bm - 1)) ? (int)0x80000000 : 0)) : 0;
        }
        wave_lds_sync();
        if (lane == 0) __hip_atomic_store(&mail->rsp_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (bm) { // the closest neighbour's list: under way while the logic wave reads the answer and decides
            const int cn = __builtin_amdgcn_readlane(nb, (int)__builtin_ctzll(bm));
            const int *pl = G.list(cn, layer);
            const int lstride = layer == 0 ? G.stride0 : G.strideU;
            hc_node = cn; hc_layer = layer;
            hc_v = lane < lstride ? pl[lane] : 0;
            hc_w = 64 < lstride ? pl[64] : 0;
        }
        MW_PH(3);
    }
}

} // namespace hnsw
