// dk_metric.h -- device code, part of device_kernels.h: the metric arithmetic: lane partials, collapse trees, int8 records, slot_distance_kernel.
#pragma once
#include "dk_base.h"

namespace hnsw {

// ---- int8 rows (BASELINE config 5; no reference counterpart: the reference is generic over TDistance,
// src/HNSWIndex/HNSWIndex.cs:6, and ships float metrics only) ----------------------------------------
// A stored row (and a resident query) is one RECORD of `pitch` 32-bit words, pitch a multiple of 16
// (64 bytes: whole fetch sectors; 128 B for dim 96):
//     words [0, pitch-2)   the quantised elements, four int8 per word, zero padded
//     word  pitch-2        scale  (float)   = max|x| / 127
//     word  pitch-1        sumsq  (int32)   = sum of q_i^2
// with q_i = clamp(rint(x_i / scale), -127, 127) (IEEE float division, round-half-even; q = 0 when the
// scale is not positive).  The kernels address records exactly like float rows of `pitch` floats, so the
// traversals, the heuristic and the link kernel are the float code; only the measure passes differ.
// Distance of records a, b -- the squared Euclidean distance of the DEQUANTISED vectors, from exact
// integers and one fixed sequence of IEEE double operations (never contracted: -ffp-contract=off):
//     dot = sum q_a q_b (int32, v_dot4_i32_i8: exact, any order)
//     A = (sa*sa)*na,  B = (sb*sb)*nb,  C = (sa*sb)*dot      (doubles; the scale products are exact)
//     d = (float)((A + B) - 2*C)
// The test-side CPU restatement of this definition does the same, so ids are bit-exact.
__device__ __forceinline__ float i8_epilogue(float sa, int na, float sb, int nb, int dot)
{
    const double A = ((double)sa * (double)sa) * (double)na;
    const double B = ((double)sb * (double)sb) * (double)nb;
    const double C = ((double)sa * (double)sb) * (double)dot;
    return (float)((A + B) - 2.0 * C);
}
__device__ __forceinline__ int dot4_i8(int a, int b, int acc) { return __builtin_amdgcn_sdot4(a, b, acc, false); }
// sum of an int over the 8 lanes of a group (every lane gets it)
// (DPP, no LDS crossbar: lane ^ 1 and lane ^ 2 are quad permutations; after them the four lanes of a quad hold the same sum, so
// the mirror inside the half row -- lane i <-> 7 - i, a lane of the group's OTHER quad -- completes it.  Exact integers: any
// pairing gives the same sum.  Until round 5 these were three ds_bpermute with their address arithmetic: the int8 search
// kernel issued 1 276 of them, and that kernel is bound by instruction issue)
__device__ __forceinline__ int group_sum_i32(int v)
{
    v += __builtin_amdgcn_mov_dpp(v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true);
    v += __builtin_amdgcn_mov_dpp(v, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, true);
    v += __builtin_amdgcn_mov_dpp(v, 0x141 /* row_half_mirror */, 0xf, 0xf, true);
    return v;
}
// what lane 8g + k of a group holds, brought to the group's lane 0 (the other lanes get whatever lies k lanes up their row of 16:
// only lane 0 of a group uses the result)
template <int K>
__device__ __forceinline__ int group_lane_to_first(int v)
{
    return __builtin_amdgcn_mov_dpp(v, 0x100 + K /* row_shl:K */, 0xf, 0xf, true);
}

// v + (the same register of lane ^ MASK), MASK in {1, 2, 4}, as DPP moves inside the row of 16 lanes -- no LDS crossbar (until round 5
// these were __shfl_xor = ds_bpermute with its address arithmetic and an LDS round trip in every collapse; the pairings, and with them
// every rounding, are the same: a + b is commutative, the TREE is what the reference fixes).  lane ^ 1 / ^ 2: quad permutations.
// lane ^ 4: lane + 4 for the lower half of an 8-lane group (row_shl:4, written to banks 0 and 2 of the row = lanes 0-3, 8-11),
// lane - 4 for the upper half (row_shr:4, banks 1 and 3).
template <int MASK>
__device__ __forceinline__ float lane_xor_add(float v)
{
    const int x = __float_as_int(v);
    int o;
    if constexpr (MASK == 1) o = __builtin_amdgcn_mov_dpp(x, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true);
    else if constexpr (MASK == 2) o = __builtin_amdgcn_mov_dpp(x, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, true);
    else {
        static_assert(MASK == 4, "lane_xor_add: 1, 2 or 4");
        o = __builtin_amdgcn_update_dpp(0, x, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
        o = __builtin_amdgcn_update_dpp(o, x, 0x114 /* row_shr:4 */, 0xf, 0xa, false);
    }
    return v + __int_as_float(o);
}

// Collapse of the eight lane partials, L2 order: EuclideanMetric.cs:45-50.
__device__ __forceinline__ float collapse_l2(float p)
{
    float t = lane_xor_add<4>(p); // p_j + p_{j+4}
    t = lane_xor_add<1>(t);       // (t0+t1), (t2+t3)
    t = lane_xor_add<2>(t);       // (t0+t1)+(t2+t3)
    return t;
}
// Collapse, cosine-family order: CosineMetric.cs:145-171.
__device__ __forceinline__ float collapse_cos(float p)
{
    float u = lane_xor_add<4>(p); // p_j + p_{j+4}
    u = lane_xor_add<2>(u);       // (u0+u2), (u1+u3)
    u = lane_xor_add<1>(u);       // (u0+u2)+(u1+u3)
    return u;
}

// Lane j (0..7) of an 8-lane group walks elements j, j+8, j+16, ... of rows a and b.
template <int METRIC>
__device__ __forceinline__ float lane_chain(const float *__restrict__ a, const float *__restrict__ b, int dim, int j)
{
    const int nblk = dim >> 3;
    float acc = 0.0f;
#pragma unroll 8
    for (int k = 0; k < nblk; ++k) {
        float x = a[8 * k + j], y = b[8 * k + j];
        if (METRIC == M_SQ) {
            float d = x - y;
            acc = __builtin_fmaf(d, d, acc); // Fma.MultiplyAdd, EuclideanMetric.cs:30
        } else {
            float p = x * y;                 // Avx.Multiply, CosineMetric.cs:114
            acc = acc + p;                   // Avx.Add      :115
        }
    }
    return acc;
}

// Scalar tail for dim % 8 != 0 (every lane redundantly; mul then add, no fma).
template <int METRIC>
__device__ __forceinline__ float scalar_tail(float s, const float *__restrict__ a, const float *__restrict__ b, int dim)
{
    for (int i = dim & ~7; i < dim; ++i) {
        float x = a[i], y = b[i];
        if (METRIC == M_SQ) {
            float d = x - y;
            float m = d * d;
            s = s + m; // EuclideanMetric.cs:53-57
        } else {
            float p = x * y;
            s = s + p; // CosineMetric.cs:135-138 / :78-85
        }
    }
    return s;
}

// Correctly rounded double sqrt from the device's sqrt plus an exact one-ulp repair
// (residual via fma; see DESIGN.md "cosine epilogue").  Math.Sqrt at CosineMetric.cs:88 is
// IEEE correctly rounded; this must be too.
__device__ inline double sqrt_rn(double x)
{
    if (!(x > 0.0) || x == __builtin_inf()) return x == 0.0 ? x : sqrt(x);
    double scale = 1.0;
    if (x < 0x1p-900) { x *= 0x1p200; scale = 0x1p-100; } // keep the residual test clear of underflow
    double y = sqrt(x);
    for (int it = 0; it < 2; ++it) {
        double r = __builtin_fma(-y, y, x);
        double yu = __longlong_as_double(__double_as_longlong(y) + 1);
        double yd = __longlong_as_double(__double_as_longlong(y) - 1);
        if (r > y * (yu - y)) y = yu;
        else if (r <= -(y * (y - yd))) y = yd;
        else break;
    }
    return y * scale;
}

// Full metric for one (row a, vector b) pair evaluated by an 8-lane group; every lane of the
// group returns the same value.  sa/sb: precomputed sqrt((double)|.|^2) for cosine.
template <int METRIC>
__device__ __forceinline__ float group_metric(const float *__restrict__ a, const float *__restrict__ b, int dim, int j,
                                              double sa, double sb)
{
    if constexpr (METRIC == M_I8) { // dim = record pitch in words; the last block's lanes 6 / 7 hold scale / sumsq
        const int *ia = reinterpret_cast<const int *>(a), *ib = reinterpret_cast<const int *>(b);
        const int nblk = dim >> 3, lane = threadIdx.x & 63;
        int acc = 0, ta = 0, tb = 0;
        for (int k = 0; k < nblk; ++k) {
            const int wa = ia[8 * k + j], wb = ib[8 * k + j];
            if (k == nblk - 1 && j >= 6) { ta = wa; tb = wb; }
            else acc = dot4_i8(wa, wb, acc);
        }
        const int dot = group_sum_i32(acc);
        const int g6 = (lane & ~7) | 6, g7 = (lane & ~7) | 7;
        return i8_epilogue(__int_as_float(__shfl(ta, g6, 64)), __shfl(ta, g7, 64), __int_as_float(__shfl(tb, g6, 64)), __shfl(tb, g7, 64), dot);
    }
    else {
    float p = lane_chain<METRIC>(a, b, dim, j);
    float s = (METRIC == M_SQ) ? collapse_l2(p) : collapse_cos(p);
    if (dim & 7) s = scalar_tail<METRIC>(s, a, b, dim);
    if (METRIC == M_SQ) return s;
    if (METRIC == M_UCOS) return 1.0f - s; // CosineMetric.cs:141
    float denom = (float)(sa * sb);        // :88  (float)(Math.Sqrt(nA) * Math.Sqrt(nB))
    if (denom < 1e-30f) return 1.0f;       // :89-90
    return 1.0f - s / denom;               // :91
    }
}

// One wave per search slot; inputs are the packed per-slot records (device_backend.h).
// Guards: a record that names a row / query outside what was uploaded, or more ids than the slot
// holds, is never dereferenced -- its distances come back NaN and `guard` is raised, which
// wait_step() turns into an error return (the records may come from a foreign host through
// hnswdev_step_submit; a bad id must not become a GPU fault).
template <int METRIC>
__global__ void __launch_bounds__(256)
slot_distance_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn,
                     const float *__restrict__ queries, const double *__restrict__ q_sn, int dim,
                     const int *__restrict__ rec, float *__restrict__ out, int stride, int rec_stride, int nslots,
                     long long n_rows, long long n_queries, int *__restrict__ guard)
{
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nslots) return;
    const int *r = rec + (size_t)s * rec_stride;
    int cnt = r[0];
    if (cnt <= 0) return;
    const int qraw = r[1];
    const int *sid = r + 2;
    const bool q_ok = qraw >= 0 ? qraw < n_queries : (long long)(~qraw) < n_rows;
    if (cnt > stride || !q_ok) {
        if (lane == 0) atomicOr(guard, 1);
        cnt = min(cnt, stride);
        for (int c = lane; c < cnt; c += 64) out[(size_t)s * stride + c] = __uint_as_float(0x7fc00000u);
        return;
    }
    const float *q;
    double sb = 0.0;
    if (qraw >= 0) {
        q = queries + (size_t)qraw * dim;
        if (METRIC == M_COS) sb = q_sn[qraw];
    } else {
        q = rows + (size_t)(~qraw) * dim;
        if (METRIC == M_COS) sb = row_sn[~qraw];
    }
    const int grp = lane >> 3, j = lane & 7;
    float *so = out + (size_t)s * stride;
    for (int c0 = 0; c0 < cnt; c0 += 8) {
        const int c = c0 + grp;
        const bool act = c < cnt;
        int id = sid[act ? c : c0]; // idle groups shadow a valid row and discard
        const bool bad = (unsigned long long)(long long)id >= (unsigned long long)n_rows;
        if (bad) id = 0;
        double sa = 0.0;
        if (METRIC == M_COS) sa = row_sn[id];
        float v = group_metric<METRIC>(rows + (size_t)id * dim, q, dim, j, sa, sb);
        if (act && j == 0) {
            so[c] = bad ? __uint_as_float(0x7fc00000u) : v;
            if (bad) atomicOr(guard, 1);
        }
    }
}

} // namespace hnsw
