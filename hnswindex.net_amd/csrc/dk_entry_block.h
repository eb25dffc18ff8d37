// dk_entry_block.h -- device code, part of device_kernels.h: the MFMA dense block on the KnnQuery path.
#pragma once
#include "dk_heuristic.h"

namespace hnsw {

// ---- the shared first hop of a batch of queries as ONE dense block (MFMA prefilter) --------------------------------
// Every KnnQuery of a batch starts FindEntryAtLayer (GraphNavigator.cs:51-82) at the same node on the same layer, so the
// first pass of every query measures the SAME rows -- the entry point and its out-edges on the top layer, at most
// MaxEdges + 1 of them: Q x (MaxEdges + 1) x dim, the one dense query x candidate contraction of the search path
// (BASELINE.json, north_star).  This kernel computes it on the matrix cores: one wave per tile of 32 queries x 32 rows
// (v_mfma_f32_32x32x2_f32, f32 in, f32 accumulate), queries and rows streamed straight into the operand layout.  As in
// RelativeNeighborPruning (dk_heuristic.h) the tile PREFILTERS and never stands in for a distance: the matrix core sums
// K products in one chain, the reference in eight interleaved chains with a tree, so the two differ in the last bits;
// with E as derived there ((1.125 K + 32) u for dots of rows no longer than 1 and for cosine's normalised dots,
// (2.25 K + 32) u (|q|^2 + |r|^2) for squared distances from |q|^2 + |r|^2 - 2 q.r) the pass's outcome -- the EARLIEST row
// of minimal distance among the entry point and its neighbours in list order (:67-78: strict improvements only) -- is
// known whenever one row beats every other by more than both error bounds; the search kernel then skips the pass, measures
// that ONE row in the reference's lane order (its distance is what the next pass compares against) and goes on.  Anything
// closer than the bounds, a NaN, a row or query longer than 1 under ucosine: no hint, the wave runs the pass as always.
// The answer travels in the job itself: SearchJob::stop_layer (unused by search jobs) = winner + 1, 0 = no hint.
// What it is worth is measured, not assumed: the pass is 17 of a query's 4 400 evaluations.
template <bool WITH_NORMS>
__device__ __forceinline__ floatx16 dot_tile(const float *__restrict__ arow, const float *__restrict__ brow, int dim, int lane, float &na, float &nb)
{
    const int h = lane >> 5;
    const float4 *pa = reinterpret_cast<const float4 *>(arow) + h;
    const float4 *pb = reinterpret_cast<const float4 *>(brow) + h;
    floatx16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
    na = 0.0f; nb = 0.0f;
    const int nt = dim >> 3;
    for (int t = 0; t < nt; ++t) {
        const float4 a = pa[2 * t], b = pb[2 * t];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        if (WITH_NORMS) {
            na = __builtin_fmaf(a.x, a.x, na); na = __builtin_fmaf(a.y, a.y, na); na = __builtin_fmaf(a.z, a.z, na); na = __builtin_fmaf(a.w, a.w, na);
            nb = __builtin_fmaf(b.x, b.x, nb); nb = __builtin_fmaf(b.y, b.y, nb); nb = __builtin_fmaf(b.z, b.z, nb); nb = __builtin_fmaf(b.w, b.w, nb);
        }
    }
    if (WITH_NORMS) { na += __shfl_xor(na, 32, 64); nb += __shfl_xor(nb, 32, 64); } // the two halves of the row
    return acc;
}

#ifdef HNSW_HOST_TU // launched from one place: defined only in the unit that launches it
template <int METRIC>
__global__ void __launch_bounds__(64)
entry_block_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, const float *__restrict__ queries, const double *__restrict__ q_sn,
                   int dim, const int64_t *__restrict__ upper, const int *__restrict__ pool, int strideU, int ep, int ep_layer,
                   SearchJob *__restrict__ jobs, int nq, unsigned long long *__restrict__ hinted)
{
    const int *ep_list = pool + upper[ep] + (size_t)(ep_layer - 1) * strideU; // GraphView::list for an upper layer
    __shared__ float D[32][33], Eb[32][33];
    __shared__ int ids[32];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int n_e = min(ep_list[0], 31);
    if (lane < 32) ids[lane] = lane == 0 ? ep : (lane - 1 < n_e ? ep_list[lane] : ep); // row 0: the entry point itself; pads repeat it and are never looked at
    __syncthreads();
    const int q = blockIdx.x * 32 + r;
    const int qc = min(q, nq - 1);
    const int rid = ids[r];
    float nqv, nrv;
    const floatx16 S = dot_tile<true>(queries + (size_t)qc * dim, rows + (size_t)rid * dim, dim, lane, nqv, nrv);
    const float u = 5.9604645e-8f, K = (float)dim;
    const float E = (1.125f * K + 32.0f) * u, Esq = (2.25f * K + 32.0f) * u * 1.01f;
    double sa_r = 0.0, sb_r = 0.0;
    if (METRIC == M_COS) { sa_r = q_sn[qc]; sb_r = row_sn[rid]; }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int i = 8 * (v >> 2) + 4 * h + (v & 3); // the query of this element; its row is j = r
        const float nq_i = __shfl(nqv, i, 64);
        float d, e;
        if (METRIC == M_SQ) { d = (nq_i + nrv) - 2.0f * S[v]; e = Esq * (nq_i + nrv); }
        else if (METRIC == M_UCOS) { d = 1.0f - S[v]; e = (nq_i <= 1.0001f && nrv <= 1.0001f) ? E : __uint_as_float(0x7f800000u); } // longer than 1: no bound, no hint
        else {
            const double sa_i = __shfl(sa_r, i, 64);
            const float denom = (float)(sa_i * sb_r);
            d = denom < 1e-30f ? 1.0f : 1.0f - S[v] / denom;
            e = denom < 1e-30f ? __uint_as_float(0x7f800000u) : E;
        }
        D[i][r] = d;
        Eb[i][r] = e;
    }
    __syncthreads();
    if (lane < 32 && q < nq) {
        // the pass's outcome: the earliest row of minimal distance -- known if one row beats all others beyond both bounds
        int w = 0;
        for (int j = 1; j <= n_e; ++j) if (D[lane][j] < D[lane][w]) w = j;
        bool sure = D[lane][w] == D[lane][w]; // (not NaN)
        const float top = D[lane][w] + Eb[lane][w];
        for (int j = 0; j <= n_e; ++j) sure = sure && (j == w || D[lane][j] - Eb[lane][j] > top);
        jobs[q].stop_layer = sure ? ids[w] + 1 : 0;
        if (sure && hinted) atomicAdd(hinted, 1ull);
    }
}
#endif

} // namespace hnsw
