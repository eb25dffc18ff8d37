// dk_heuristic.h -- device code, part of device_kernels.h: RelativeNeighborPruning: the restated Span.Sort, the MFMA Gram-block prefilter, grouped and one-by-one forms.
#pragma once
#include "dk_measure.h"
#include "dk_search_common.h"

namespace hnsw {

// ---- MemoryExtensions.Sort(Span<NodeDistance>, DistanceComparer) on an LDS array: the BCL
// introsort restated (insertion sort <= 16, median of three, heapsort at depth limit
// 2*(log2 n + 1)); wave-uniform scalar code, recursion replaced by a work stack in LDS.
// Same algorithm as csrc/host_structs.h::dotnet_sort, so tie order is identical. ----
__device__ __forceinline__ void sw_swap(ND *k, int i, int j) { ND t = k[i]; k[i] = k[j]; k[j] = t; }
__device__ __forceinline__ void sw_swap_if_greater(ND *k, int i, int j) { if (nd_cmp<false>(k[i], k[j]) > 0) sw_swap(k, i, j); }
__device__ inline void sw_insertion(ND *k, int n)
{
    for (int i = 0; i < n - 1; i++) {
        ND t = k[i + 1];
        int j = i;
        while (j >= 0 && nd_cmp<false>(t, k[j]) < 0) { k[j + 1] = k[j]; j--; }
        k[j + 1] = t;
    }
}
__device__ inline void sw_down_heap(ND *k, int i, int n)
{
    ND d = k[i - 1];
    while (i <= (n >> 1)) {
        int child = 2 * i;
        if (child < n && nd_cmp<false>(k[child - 1], k[child]) < 0) child++;
        if (!(nd_cmp<false>(d, k[child - 1]) < 0)) break;
        k[i - 1] = k[child - 1];
        i = child;
    }
    k[i - 1] = d;
}
__device__ inline void sw_heap_sort(ND *k, int n)
{
    for (int i = n >> 1; i >= 1; i--) sw_down_heap(k, i, n);
    for (int i = n; i > 1; i--) { sw_swap(k, 0, i - 1); sw_down_heap(k, 1, i - 1); }
}
__device__ inline int sw_partition(ND *k, int n)
{
    int hi = n - 1, mid = hi >> 1;
    sw_swap_if_greater(k, 0, mid);
    sw_swap_if_greater(k, 0, hi);
    sw_swap_if_greater(k, mid, hi);
    ND pivot = k[mid];
    sw_swap(k, mid, hi - 1);
    int left = 0, right = hi - 1;
    while (left < right) {
        while (nd_cmp<false>(k[++left], pivot) < 0) {}
        while (nd_cmp<false>(pivot, k[--right]) < 0) {}
        if (left >= right) break;
        sw_swap(k, left, right);
    }
    if (left != hi - 1) sw_swap(k, left, hi - 1);
    return left;
}
__device__ inline void dev_dotnet_sort(ND *arr, int n, int *stk)
{
    if (n <= 1) return;
    int sp = 0;
    stk[0] = 0; stk[1] = n; stk[2] = 2 * ((31 - __clz(n)) + 1);
    sp = 1;
    while (sp > 0) {
        --sp;
        ND *k = arr + stk[3 * sp];
        int ps = stk[3 * sp + 1];
        int depth = stk[3 * sp + 2];
        while (ps > 1) {
            if (ps <= 16) {
                if (ps == 2) { sw_swap_if_greater(k, 0, 1); break; }
                if (ps == 3) { sw_swap_if_greater(k, 0, 1); sw_swap_if_greater(k, 0, 2); sw_swap_if_greater(k, 1, 2); break; }
                sw_insertion(k, ps);
                break;
            }
            if (depth == 0) { sw_heap_sort(k, ps); break; }
            depth--;
            int p = sw_partition(k, ps);
            // right part [p+1, ps) is an independent sub-problem: queue it (the BCL recurses into it)
            if (sp < 39) {
                stk[3 * sp] = (int)(k - arr) + p + 1; stk[3 * sp + 1] = ps - (p + 1); stk[3 * sp + 2] = depth;
                ++sp;
            }
            ps = p;
        }
    }
}

// ---- MFMA Gram block: the dense contraction inside RelativeNeighborPruning ----------------------
// Heuristic.cs:23-40 tests every candidate c against every id s accepted so far: dist(s, c) < c.Dist.  Over
// a block of candidates that is a dense C x C (and accepted x C) block of pair distances -- dot products of
// stored rows -- the one place on this path where a matrix core applies.  v_mfma_f32_32x32x2_f32 (f32 in,
// f32 accumulate: a chain of K fused multiply-adds per output element) gives a 32 x 32 tile of dots per pass
// over the rows; it CANNOT reproduce the lane-ordered sums bit for bit, so it never stands in for a distance:
// it only PREFILTERS the comparison.  Both sums round at most once per step, each step by at most
// u |partial sum| <= u S with S = sum |a_k b_k| <= |a| |b| (u = 2^-24): the MFMA chain has K steps, the lane
// order K/8 adds per lane plus a product rounding per term (u S in total) plus a three-level tree, so
// |mfma - lane-ordered| <= (K + K/8 + 5) u S.  With E = (1.125 K + 32) u -- for rows of length <= 1 (ucosine;
// checked per block on the Gram diagonal, a longer row sends its block to the exact path) or after the
// division by the norms (cosine) -- a pair whose approximate distance is further than E from the threshold
// has the same outcome as the exact test, and a pair within E is evaluated again with the exact kernels
// (measure_all).  Measured (tools/mfma_probe.hip, K = 768): the two sums differ by 9.5e-7 at most; E = 5.3e-5.
// Ids are therefore decided by exact fp32 distances or by a margin no rounding can cross; the graph hashes of
// the parity tests (oracle: scalar CPU code) hold this at every size.
typedef float floatx16 __attribute__((ext_vector_type(16)));
// D[i][j] = dot(row idA of lane (i = lane % 32 as A operand), row idB (j = lane % 32 as B operand)); result layout,
// measured (tools/mfma_probe.hip): lane l, register v hold j = l % 32, i = 8 (v / 4) + 4 (l / 32) + v % 4.
// Lane (r, h) streams floats [8 t + 4 h, 8 t + 4 h + 4) of its row: which k meets which MFMA step is free as long as
// both operands agree.  dim % 8 == 0.
__device__ __forceinline__ floatx16 gram_tile(const float *__restrict__ rows, int dim, int idA, int idB, int lane)
{
    const int h = lane >> 5;
    const float4 *pa = reinterpret_cast<const float4 *>(rows + (size_t)idA * dim) + h;
    const float4 *pb = reinterpret_cast<const float4 *>(rows + (size_t)idB * dim) + h;
    floatx16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
    const int nt = dim >> 3;
    constexpr int U = 8;
    int t = 0;
    for (; t + U <= nt; t += U) {
        float4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = pa[2 * (t + u)]; b[u] = pb[2 * (t + u)]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
        }
    }
    for (; t < nt; ++t) {
        const float4 a = pa[2 * t], b = pb[2 * t];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    return acc;
}

__device__ __forceinline__ int nbcap_of(int max_edges) { return (max_edges + 1 + 7) & ~7; } // row stride of the grouped heuristic's distance table
// Heuristic.RelativeNeighborPruning (Heuristic.cs:11-46) on cands[0..n) (LDS): writes the
// selected ids to L.acc, returns their count.  The candidate under test is staged in L.qs2
// and measured against ALL accepted rows at once (the reference's early break only skips
// evaluations).
template <int METRIC, bool MFMA = false>
__device__ __forceinline__ int relative_neighbor_pruning(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                                                         ND *cands, int n, int max_edges, const SearchLds &L, int lane,
                                                         unsigned long long &evals, bool presorted = false,
                                                         float *gscratch = nullptr, size_t gscratch_bytes = 0, bool mfma_ok = false)
{
    int *acc = L.acc;
    wave_sync();
    if (n < max_edges) { // :13-18 input (heap) order, unsorted
        for (int i = lane; i < n; i += 64) acc[i] = cands[i].id;
        wave_sync();
        return n;
    }
    if (!presorted) { // :22 (a sorted-list traversal hands them over in order)
        bool ranked = false;
        if (n <= 64) {
            // distinct ordinary distances have one ascending order whatever the sort: rank by counting
            // (the link kernel's 2M+1 candidates; the scalar introsort below was 9 % of a PruneOverflow)
            const ND mine = lane < n ? cands[lane] : ND{0, 0.0f};
            const unsigned my_key = f2key(mine.dist);
            bool odd = lane < n && key_unsafe(mine.dist);
            int rank = 0;
            for (int t = 0; t < n; ++t) {
                const unsigned kt = (unsigned)__builtin_amdgcn_readlane((int)my_key, t);
                rank += kt < my_key ? 1 : 0;
                odd |= lane < n && t != lane && kt == my_key;
            }
            if (__ballot(odd) == 0ull) {
                wave_sync();
                if (lane < n) cands[rank] = mine;
                ranked = true;
            }
        }
        if (!ranked) dev_dotnet_sort(cands, n, L.stk); // equal / NaN / -0 distances: the BCL introsort decides
    }
    wave_sync();
    int rc = 0;
    constexpr int kPre = 4; // floats per lane: rows up to 256 floats; longer rows (bandwidth-bound anyway) are staged on demand
    const bool prefetch = dim <= 64 * kPre;
    const int dimp = (dim + 3) & ~3;
    constexpr int kPreG = 4; // the grouped form prefetches four rows at once: rows up to 256 floats
    if constexpr (MFMA && (METRIC == M_UCOS || METRIC == M_COS || METRIC == M_SQ)) {
        // MFMA-prefiltered form (rows of a multiple of 8 floats, at least 256 of them: measured at C2's 128-float rows
        // the tiles cost more than the grouped form below -- insert kernel 0.94 s against 0.83 s -- at C3's 768 they
        // save 13 % of it; instantiated for the 8-register-set kernels only, i.e. beams above 256 candidates, which
        // have the registers -- in the 168-VGPR variants the extra code spilled): candidates in blocks of 32.
        // Per block: one tile per 32 accepted ids (accepted x block) and one block x block tile give the
        // approximate distance of every pair the greedy pass can ask for; the pass then walks the 32 in order
        // on those numbers, and only a pair within E of its threshold is measured exactly.
        // sq_euclid: |a - b|^2 = na + nb - 2 dot with the three terms off the same tiles (na, nb: the Gram diagonal);
        // each is a K-step chain, so |approx - lane-ordered| <= u (K (na + nb + 2 S) + (K/8 + 5) D) with S <= (na + nb) / 2
        // and D = |a - b|^2 <= 2 (na + nb): E_pair = (2.25 K + 32) u (na + nb), norms taken 1 % up for their own error.
        const size_t need_sn = METRIC == M_COS ? 8u * (size_t)nbcap_of(max_edges) : METRIC == M_SQ ? 4u * (size_t)nbcap_of(max_edges) : 0u;
        if ((dim & 7) == 0 && dim >= 256 && mfma_ok && (METRIC == M_UCOS || (gscratch && gscratch_bytes >= need_sn))) {
            const float E = (1.125f * (float)dim + 32.0f) * 5.9604645e-8f;
            const float Esq = (2.25f * (float)dim + 32.0f) * 5.9604645e-8f * 1.01f;
            double *snacc = reinterpret_cast<double *>(gscratch); // cosine: sqrt-norms of the accepted rows, by position
            float *nacc = reinterpret_cast<float *>(gscratch);    // sq_euclid: their squared norms (Gram diagonal)
            const int r = lane & 31, h = lane >> 5;
            float *qbuf = L.qs2;
            // the exact test of one candidate against everything accepted so far (Heuristic.cs:31-35)
            auto exact_rejects = [&](const ND c) -> bool {
                const float *crow = rows + (size_t)c.id * dim;
                wave_sync();
                for (int e = lane; e < dim; e += 64) qbuf[e] = crow[e];
                double sbc = 0.0;
                if (METRIC == M_COS) sbc = row_sn[c.id];
                wave_sync();
                bool ok = true;
                const int chunk = dim >= 512 ? 16 : 32;
                for (int a0 = 0; a0 < rc && ok; a0 += chunk) {
                    const int an = min(chunk, rc - a0);
                    measure_all<METRIC>(rows, row_sn, dim, qbuf, sbc, acc + a0, L.dbuf, an, lane);
                    wave_sync();
                    evals += (unsigned long long)an;
                    const float dj = lane < an ? L.dbuf[lane] : 0.0f;
                    ok = __ballot(lane < an && dj < c.dist) == 0ull;
                    wave_sync();
                }
                return !ok;
            };
            for (int b0 = 0; b0 < n && rc < max_edges; b0 += 32) { // :23, thirty-two at a time
                const int bsz = min(32, n - b0);
                const ND mine = cands[b0 + (r < bsz ? r : 0)]; // column j = r of this block
                const float thr = mine.dist;
                double sn_j = 0.0;
                if (METRIC == M_COS) sn_j = row_sn[mine.id];
                const int rc0 = rc;
                const floatx16 S = gram_tile(rows, dim, mine.id, mine.id, lane); // block x block
                float sd[16], se[16]; // block x block: approximate distance and (sq_euclid) its error bound
                bool long_row = false; // ucosine: the bound assumes |row| <= 1
                float n_j = 0.0f;      // sq_euclid: |row j|^2 off the diagonal (one of the lanes r, r + 32 holds it)
                if (METRIC == M_SQ) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) n_j += (8 * (v >> 2) + 4 * h + (v & 3)) == r ? S[v] : 0.0f;
                    n_j += __shfl_xor(n_j, 32, 64);
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int i = 8 * (v >> 2) + 4 * h + (v & 3);
                    se[v] = E;
                    if (METRIC == M_UCOS) {
                        sd[v] = 1.0f - S[v];
                        long_row = long_row || (i == r && !(S[v] <= 1.0001f));
                    } else if (METRIC == M_SQ) {
                        const float n_i = __shfl(n_j, i, 64);
                        sd[v] = (n_i + n_j) - 2.0f * S[v];
                        se[v] = Esq * (n_i + n_j);
                    } else {
                        const double sn_i = __shfl(sn_j, i, 64); // row i of the block = column i's own norm
                        const float denom = (float)(sn_i * sn_j);
                        sd[v] = denom < 1e-30f ? 1.0f : 1.0f - S[v] / denom;
                    }
                }
                if (__ballot(long_row) != 0ull) { // not unit rows: this block on the exact kernels alone
                    for (int j = 0; j < bsz && rc < max_edges; ++j) {
                        const ND c = cands[b0 + j];
                        if (rc == 0 || !exact_rejects(c)) { if (lane == 0) acc[rc] = c.id; rc++; }
                        wave_sync();
                    }
                    continue;
                }
                bool def_r = false, unc_r = false; // column j against the ids accepted before the block
                for (int a0 = 0; a0 < rc0; a0 += 32) {
                    const int na = min(32, rc0 - a0);
                    const floatx16 D = gram_tile(rows, dim, acc[a0 + (r < na ? r : 0)], mine.id, lane);
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int i = 8 * (v >> 2) + 4 * h + (v & 3);
                        float d, e = E;
                        if (METRIC == M_UCOS) d = 1.0f - D[v];
                        else if (METRIC == M_SQ) {
                            const float n_i = nacc[a0 + (i < na ? i : 0)];
                            d = (n_i + n_j) - 2.0f * D[v];
                            e = Esq * (n_i + n_j);
                        } else {
                            const float denom = (float)(snacc[a0 + (i < na ? i : 0)] * sn_j);
                            d = denom < 1e-30f ? 1.0f : 1.0f - D[v] / denom;
                        }
                        const bool valid = i < na && r < bsz;
                        def_r = def_r || (valid && d < thr - e);
                        unc_r = unc_r || (valid && !(d < thr - e) && !(d > thr + e)); // also catches NaN
                    }
                }
                unsigned in_block = 0u; // bit u: member u of the block accepted (uniform)
                for (int j = 0; j < bsz && rc < max_edges; ++j) {
                    bool def = r == j && def_r, unc = r == j && unc_r;
                    if (r == j) {
#pragma unroll
                        for (int v = 0; v < 16; ++v) {
                            const int i = 8 * (v >> 2) + 4 * h + (v & 3);
                            const bool live = ((in_block >> i) & 1u) != 0u; // accepted members all precede j
                            def = def || (live && sd[v] < thr - se[v]);
                            unc = unc || (live && !(sd[v] < thr - se[v]) && !(sd[v] > thr + se[v]));
                        }
                    }
                    const bool any_def = __ballot(def) != 0ull, any_unc = __ballot(unc) != 0ull;
                    const ND c = cands[b0 + j];
                    bool rejected = any_def;
                    if (!any_def && any_unc) rejected = exact_rejects(c); // too close to call
                    if (!rejected) {
                        if (lane == 0) { acc[rc] = c.id; if (METRIC == M_COS) snacc[rc] = row_sn[c.id]; }
                        if (METRIC == M_SQ) { const float nj = __shfl(n_j, j, 64); if (lane == 0) nacc[rc] = nj; }
                        rc++;
                        in_block |= 1u << j;
                    }
                }
                evals += (unsigned long long)(rc0 + bsz); // rows streamed by the tiles of this block (each once per tile)
                wave_sync(); // acc / snacc written by lane 0 are read by the next block's tiles
            }
            return rc;
        }
    }
    if constexpr (METRIC != M_I8) {
        // Grouped form (rows up to 256 floats, when the caller lends scratch): FOUR candidates are tested per
        // step.  Their rows sit in LDS; every accepted row is fetched once and measured against all four
        // (measure_multi), the six pairs inside the group are measured from LDS alone, and the greedy pass
        // :23-40 then runs over the four in order on those numbers -- a candidate is rejected by an id accepted
        // before the group (D) or by an earlier member of the group that was accepted (P).  Same distances,
        // same decisions, a quarter of the dependent round trips and of the row reads.
        const size_t need = 2u * 4u * (size_t)dimp + 4u * 4u * (size_t)nbcap_of(max_edges) + 64u + 64u;
        if (dim <= 64 * kPreG && gscratch && gscratch_bytes >= need) {
            // (no indexed local arrays below: they would live in scratch memory)
            auto gq = [&](int t) -> float * { return t < 2 ? L.qs2 + t * dimp : gscratch + (t - 2) * dimp; };
            float *D = gscratch + 2 * dimp;
            const int ds = nbcap_of(max_edges);
            float *P = D + 4 * ds;                                  // P[u * 4 + t], u < t
            double *sbq = reinterpret_cast<double *>(P + 16);       // cosine: sqrt-norms of the group's rows [0..4), of the next group's [4..8)
            {   // stage the first group
                const int gsz = min(4, n);
                for (int t = 0; t < gsz; ++t) {
                    const float *crow = rows + (size_t)cands[t].id * dim;
                    float *dst = gq(t);
                    for (int e = lane; e < dim; e += 64) dst[e] = crow[e];
                    if (METRIC == M_COS && lane == 0) sbq[t] = row_sn[cands[t].id];
                }
                wave_sync();
            }
            for (int g0 = 0; g0 < n && rc < max_edges; g0 += 4) { // :23, four at a time
                const int gsz = min(4, n - g0);
                // the next group's rows: loads in flight while this group is tested
                float pre0[kPreG], pre1[kPreG], pre2[kPreG], pre3[kPreG];
                const int nsz = min(4, max(0, n - (g0 + 4)));
#define HNSW_PRE_LOAD(T, PRE)                                                                          \
                if (T < nsz) {                                                                         \
                    const int nid = cands[g0 + 4 + T].id;                                              \
                    const float *nrow = rows + (size_t)nid * dim;                                      \
                    _Pragma("unroll") for (int e = 0; e < kPreG; ++e)                                  \
                        if (64 * e < dim) PRE[e] = lane + 64 * e < dim ? nrow[lane + 64 * e] : 0.0f;   \
                    if (METRIC == M_COS && lane == 0) sbq[4 + T] = row_sn[nid];                        \
                }
                HNSW_PRE_LOAD(0, pre0) HNSW_PRE_LOAD(1, pre1) HNSW_PRE_LOAD(2, pre2) HNSW_PRE_LOAD(3, pre3)
#undef HNSW_PRE_LOAD
                const int rc0 = rc;
                if (rc0 > 0) { // distanceFnc(s.Id, candidateId) :34 for every accepted s and the four candidates
                    if (gsz == 4) measure_multi<METRIC, 4>(rows, row_sn, dim, gq(0), gq(1), gq(2), gq(3), sbq, acc, rc0, D, ds, lane);
                    else if (gsz == 3) measure_multi<METRIC, 3>(rows, row_sn, dim, gq(0), gq(1), gq(2), gq(3), sbq, acc, rc0, D, ds, lane);
                    else if (gsz == 2) measure_multi<METRIC, 2>(rows, row_sn, dim, gq(0), gq(1), gq(2), gq(3), sbq, acc, rc0, D, ds, lane);
                    else measure_multi<METRIC, 1>(rows, row_sn, dim, gq(0), gq(1), gq(2), gq(3), sbq, acc, rc0, D, ds, lane);
                    evals += (unsigned long long)rc0; // rows fetched
                }
                // pairs inside the group, from LDS: lane group p <-> pair (u, t), u < t
                {
                    const int pg = lane >> 3, j = lane & 7;
                    const int pu = pg == 0 ? 0 : pg == 1 ? 0 : pg == 2 ? 1 : pg == 3 ? 0 : pg == 4 ? 1 : 2;
                    const int pt = pg == 0 ? 1 : pg <= 2 ? 2 : 3;
                    const bool live = pg < 6 && pt < gsz;
                    double sa = 0.0, sb = 0.0;
                    if (METRIC == M_COS) { sa = sbq[live ? pu : 0]; sb = sbq[live ? pt : 0]; }
                    const float v = group_metric<METRIC>(gq(live ? pu : 0), gq(live ? pt : 0), dim, j, sa, sb);
                    if (live && j == 0) P[pu * 4 + pt] = v;
                }
                wave_sync();
                unsigned in_group = 0u; // bit u: member u accepted
                for (int t = 0; t < gsz && rc < max_edges; ++t) {
                    const ND c = cands[g0 + t];
                    bool rej = false;
                    for (int r0 = 0; r0 < rc0; r0 += 64) {
                        const int r = r0 + lane;
                        const float dj = r < rc0 ? D[t * ds + r] : 0.0f;
                        rej = rej || __ballot(r < rc0 && dj < c.dist) != 0ull;
                    }
                    for (int u = 0; u < t; ++u)
                        if ((in_group >> u) & 1u) rej = rej || P[u * 4 + t] < c.dist;
                    if (!rej) { if (lane == 0) acc[rc] = c.id; rc++; in_group |= 1u << t; }
                }
                wave_sync();
#define HNSW_PRE_STORE(T, PRE)                                                                         \
                if (T < nsz) {                                                                         \
                    float *dst = gq(T);                                                                \
                    _Pragma("unroll") for (int e = 0; e < kPreG; ++e)                                  \
                        if (64 * e < dim && lane + 64 * e < dim) dst[lane + 64 * e] = PRE[e];          \
                    if (METRIC == M_COS && lane == 0) sbq[T] = sbq[4 + T];                             \
                }
                HNSW_PRE_STORE(0, pre0) HNSW_PRE_STORE(1, pre1) HNSW_PRE_STORE(2, pre2) HNSW_PRE_STORE(3, pre3)
#undef HNSW_PRE_STORE
                wave_sync();
            }
            return rc;
        }
    }
    // One candidate per step.  The row of candidate i + 1 is fetched while candidate i is being tested
    // (registers, then the other of two LDS buffers): one dependent memory round trip per candidate instead of two.
    float *buf[2] = {L.qs2, L.qs3};
    int cur = 0;
    double sbc = 0.0, sbn = 0.0;
    for (int i = 0; i < n && rc < max_edges; ++i) { // :23
        const ND c = cands[i];
        float pre[kPre];
        const bool have_next = prefetch && i + 1 < n;
        if (have_next) {
            const int nid = cands[i + 1].id;
            const float *nrow = rows + (size_t)nid * dim;
#pragma unroll
            for (int t = 0; t < kPre; ++t)
                if (64 * t < dim) pre[t] = lane + 64 * t < dim ? nrow[lane + 64 * t] : 0.0f;
            if (METRIC == M_COS) sbn = row_sn[nid];
        }
        bool ok = true;
        if (rc > 0) {
            if (!prefetch) { // candidate i on demand
                const float *crow = rows + (size_t)c.id * dim;
                for (int t = lane; t < dim; t += 64) buf[cur][t] = crow[t];
                if (METRIC == M_COS) sbc = row_sn[c.id];
                wave_sync();
            }
            // accepted ids are measured in chunks, in acceptance order, stopping at the first chunk
            // that rejects (the reference breaks at the first hit, :34; later pairs cannot change the
            // outcome) -- with long rows this saves most of the traffic of rejected candidates
            const int chunk = dim >= 512 ? 16 : 32;
            for (int a0 = 0; a0 < rc && ok; a0 += chunk) {
                const int an = min(chunk, rc - a0);
                measure_all<METRIC>(rows, row_sn, dim, buf[cur], sbc, acc + a0, L.dbuf, an, lane); // distanceFnc(s.Id, candidateId) :34
                wave_sync();
                evals += (unsigned long long)an;
                const float dj = lane < an ? L.dbuf[lane] : 0.0f;
                ok = __ballot(lane < an && dj < c.dist) == 0ull;
                wave_sync();
            }
        }
        if (ok) { if (lane == 0) acc[rc] = c.id; rc++; }
        if (have_next) {
#pragma unroll
            for (int t = 0; t < kPre; ++t)
                if (64 * t < dim && lane + 64 * t < dim) buf[cur ^ 1][lane + 64 * t] = pre[t];
            cur ^= 1;
            sbc = sbn;
        }
        wave_sync();
    }
    return rc;
}

} // namespace hnsw
