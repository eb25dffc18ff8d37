// dk_misc_kernels.h -- device code, part of device_kernels.h: pair distances, per-row norms, int8 quantisation, row download, the sqrt test kernel.
#pragma once
#include "dk_metric.h"

namespace hnsw {

// Flat id<->id pairs: 8 lanes per pair (hnswdev_dist_pair_batch).  Ids outside the uploaded rows
// give NaN and raise `guard` (see slot_distance_kernel).
template <int METRIC>
__global__ void __launch_bounds__(256)
pair_distance_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                     const int *__restrict__ a_ids, const int *__restrict__ b_ids, float *__restrict__ out, int n,
                     long long n_rows, int *__restrict__ guard)
{
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 3;
    const int j = threadIdx.x & 7;
    const bool act = g < n;
    int a = a_ids[act ? g : 0], b = b_ids[act ? g : 0];
    const bool bad = (unsigned long long)(long long)a >= (unsigned long long)n_rows || (unsigned long long)(long long)b >= (unsigned long long)n_rows;
    if (bad) { a = 0; b = 0; }
    double sa = 0.0, sb = 0.0;
    if (METRIC == M_COS) { sa = row_sn[a]; sb = row_sn[b]; }
    float r = group_metric<METRIC>(rows + (size_t)a * dim, rows + (size_t)b * dim, dim, j, sa, sb);
    if (act && j == 0) {
        out[g] = bad ? __uint_as_float(0x7fc00000u) : r;
        if (bad) atomicOr(guard, 1);
    }
}

// sqrt((double)|row|^2) with |row|^2 summed in f32 in the reference's lane order
// (CosineMetric.cs:40-41,47 / :43-44,48 and the tail :83-84): 8 lanes per row.
#ifdef HNSW_HOST_TU // non-template kernels: only the unit that launches them defines them
__global__ void __launch_bounds__(256)
row_sqrtnorm_kernel(const float *__restrict__ rows, int dim, long long first, int n, double *__restrict__ out)
{
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 3;
    const int j = threadIdx.x & 7;
    const bool act = g < n;
    const float *a = rows + (size_t)(first + (act ? g : 0)) * dim;
    float p = lane_chain<M_COS>(a, a, dim, j);
    float s = collapse_cos(p);
    if (dim & 7) s = scalar_tail<M_COS>(s, a, a, dim);
    if (act && j == 0) out[first + g] = sqrt_rn((double)s);
}
#endif

// exposed for tests: sqrt_rn over an array
#ifdef HNSW_HOST_TU
// float rows -> int8 records (see the layout above): one wave per row; lane l owns elements 4l .. 4l+3 of
// each 256-element stretch.  max and the integer sum are exact in any order.
__global__ void __launch_bounds__(256)
quantize_rows_kernel(const float *__restrict__ src, int dim, int n, float *__restrict__ dst, long long first, int pitch)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float *x = src + (size_t)r * dim;
    int *rec = reinterpret_cast<int *>(dst + (size_t)(first + r) * pitch);
    float m = 0.0f;
    for (int i = lane; i < dim; i += 64) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float scale = m / 127.0f;
    int sumsq = 0;
    const int nwords = pitch - 2;
    for (int w = lane; w < nwords; w += 64) {
        unsigned packed = 0u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = 4 * w + t;
            int q = 0;
            if (i < dim && scale > 0.0f) {
                const float v = __builtin_rintf(x[i] / scale);
                q = (int)fminf(fmaxf(v, -127.0f), 127.0f);
            }
            sumsq += q * q;
            packed |= (unsigned)(q & 0xff) << (8 * t);
        }
        rec[w] = (int)packed;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sumsq += __shfl_xor(sumsq, o, 64);
    if (lane == 0) { rec[pitch - 2] = __float_as_int(scale); rec[pitch - 1] = sumsq; }
}
// records -> the dequantised float rows q_i * scale (hnswdev_download_rows on an int8 context)
__global__ void __launch_bounds__(256)
dequantize_rows_kernel(const float *__restrict__ recs, int pitch, long long first, int n, int dim, float *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n * dim) return;
    const int r = (int)(t / dim), i = (int)(t % dim);
    const int *rec = reinterpret_cast<const int *>(recs + (size_t)(first + r) * pitch);
    const int q = (int)(signed char)((rec[i >> 2] >> (8 * (i & 3))) & 0xff);
    out[t] = (float)q * __int_as_float(rec[pitch - 2]);
}
#endif
#ifdef HNSW_HOST_TU // non-template kernels: only the unit that launches them defines them
__global__ void sqrt_rn_kernel(const double *in, double *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sqrt_rn(in[i]);
}
#endif

} // namespace hnsw
