// dk_measure.h -- device code, part of device_kernels.h: measure passes: rows of up to 32 candidates per memory round trip (8 lanes per row; two lanes per row; int8; multi-vector).
#pragma once
#include "dk_heaps.h"

namespace hnsw {

// Distances of nbuf[0..m) to the query staged in LDS (qs), written to dbuf[0..m).
// 8 lanes per candidate, NP candidates per lane group in flight (row loads of all NP passes
// are independent, so one HBM round trip serves up to 8*NP rows).
template <int METRIC, int NP>
__device__ __forceinline__ void measure_pass(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                                             const float *qs, double sb, const int *nbuf, float *dbuf, int p0, int m, int lane)
{
    const int grp = lane >> 3, j = lane & 7;
    const float *a[NP];
    int cidx[NP];
    float acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int c = p0 + grp + 8 * p;
        cidx[p] = c;
        int id = nbuf[c < m ? c : p0]; // idle groups shadow a valid row
        a[p] = rows + (size_t)id * dim;
        acc[p] = 0.0f;
    }
    const int nblk = dim >> 3;
    int k = 0;
    // All row loads of a 16-block (128-float) chunk are issued before any arithmetic, so a chunk
    // costs ONE memory round trip for its 8 * NP rows: the lane partials must be summed in k
    // order, the loads need not be issued in it.  (A plain unrolled loop waits per unroll group --
    // four dependent round trips per 512-B row pass, most of an expansion's latency.)
    for (; k + 16 <= nblk; k += 16) {
        float x[NP][16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int p = 0; p < NP; ++p) x[p][kk] = a[p][8 * (k + kk) + j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float y = qs[8 * (k + kk) + j];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (METRIC == M_SQ) {
                    const float d = x[p][kk] - y;
                    acc[p] = __builtin_fmaf(d, d, acc[p]);
                } else {
                    const float pr = x[p][kk] * y;
                    acc[p] = acc[p] + pr;
                }
            }
        }
    }
#pragma unroll 4
    for (; k < nblk; ++k) {
        float y = qs[8 * k + j];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float x = a[p][8 * k + j];
            if (METRIC == M_SQ) {
                float d = x - y;
                acc[p] = __builtin_fmaf(d, d, acc[p]);
            } else {
                float pr = x * y;
                acc[p] = acc[p] + pr;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        float s = (METRIC == M_SQ) ? collapse_l2(acc[p]) : collapse_cos(acc[p]);
        if (dim & 7) s = scalar_tail<METRIC>(s, a[p], qs, dim);
        float r;
        if (METRIC == M_SQ) r = s;
        else if (METRIC == M_UCOS) r = 1.0f - s;
        else {
            int id = nbuf[cidx[p] < m ? cidx[p] : p0];
            float denom = (float)(row_sn[id] * sb);
            r = (denom < 1e-30f) ? 1.0f : 1.0f - s / denom;
        }
        if (j == 0 && cidx[p] < m) dbuf[cidx[p]] = r;
    }
}

// lane P of each 8-lane group takes candidate P's dot sum, scale and sumsq (see the end of measure_pass_i8)
template <int P, int NP>
__device__ __forceinline__ void i8_gather_lane(const int (&acc)[NP], const int (&tr)[NP], int j, int &dj, int &saj, int &naj)
{
    if constexpr (P < NP) {
        const int dot = group_sum_i32(acc[P]);
        const int sa = __builtin_amdgcn_mov_dpp(tr[P], 0x100 + (6 - P) /* row_shl: lane P reads the group's lane 6 */, 0xf, 0xf, true);
        const int na = __builtin_amdgcn_mov_dpp(tr[P], 0x100 + (7 - P) /* lane 7 */, 0xf, 0xf, true);
        const bool mine = (j == P);
        dj = mine ? dot : dj;
        saj = mine ? sa : saj;
        naj = mine ? na : naj;
        i8_gather_lane<P + 1, NP>(acc, tr, j, dj, saj, naj);
    }
}

// int8 records: NP candidates per lane group, every load of the pass issued before any arithmetic (one
// memory round trip for up to 8 * NP records); qs = the query's record staged in LDS.
template <int NP, int NB>
__device__ __forceinline__ void measure_pass_i8(const float *__restrict__ rows, int pitch, const float *qs, const int *nbuf, float *dbuf,
                                                int p0, int m, int lane)
{
    const int grp = lane >> 3, j = lane & 7;
    const int *a[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int c = p0 + grp + 8 * p;
        const int id = nbuf[c < m ? c : p0]; // idle groups shadow a valid record
        a[p] = reinterpret_cast<const int *>(rows + (size_t)id * pitch);
    }
    const int *iq = reinterpret_cast<const int *>(qs);
    const int nblk = NB > 0 ? NB : (pitch >> 3);
    int acc[NP], tr[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) { acc[p] = 0; tr[p] = 0; }
    if constexpr (NB > 0) {
        int w[NP][NB];
#pragma unroll
        for (int k = 0; k < NB; ++k)
#pragma unroll
            for (int p = 0; p < NP; ++p) w[p][k] = a[p][8 * k + j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int y = iq[8 * k + j];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (k == NB - 1) { if (j >= 6) tr[p] = w[p][k]; else acc[p] = dot4_i8(w[p][k], y, acc[p]); }
                else acc[p] = dot4_i8(w[p][k], y, acc[p]);
            }
        }
    } else {
        for (int k = 0; k < nblk; ++k) {
            const int y = iq[8 * k + j];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int wv = a[p][8 * k + j];
                if (k == nblk - 1 && j >= 6) tr[p] = wv;
                else acc[p] = dot4_i8(wv, y, acc[p]);
            }
        }
    }
    const float sq = __int_as_float(iq[pitch - 2]);
    const int nq = iq[pitch - 1];
    // ONE epilogue per pass: lane p of a group finishes the group's candidate p (the double arithmetic of i8_epilogue is ~20
    // instructions; run once per candidate register it was half of the pass's instruction count, and the int8 traversal is bound
    // by instruction issue).  Every lane of a group holds the group's dot sums; the scale and sumsq of candidate p
    // sit in lanes 6 / 7 of the group and reach lane p by row_shl:(6-p) / (7-p).
    int dj = group_sum_i32(acc[0]);
    int saj = group_lane_to_first<6>(tr[0]);
    int naj = group_lane_to_first<7>(tr[0]);
    i8_gather_lane<1, NP>(acc, tr, j, dj, saj, naj);
    const float r = i8_epilogue(__int_as_float(saj), naj, sq, nq, dj);
    const int c = p0 + grp + 8 * j;
    if (j < NP && c < m) dbuf[c] = r;
}
template <int NP>
__device__ __forceinline__ void measure_pass_i8_any(const float *rows, int pitch, const float *qs, const int *nbuf, float *dbuf, int p0, int m, int lane)
{
    // the common record sizes keep their words in registers: 128 B (dim <= 120), 192 B, 256 B
    if (pitch == 32) measure_pass_i8<NP, 4>(rows, pitch, qs, nbuf, dbuf, p0, m, lane);
    else if (pitch == 48) measure_pass_i8<NP, 6>(rows, pitch, qs, nbuf, dbuf, p0, m, lane);
    else if (pitch == 16) measure_pass_i8<NP, 2>(rows, pitch, qs, nbuf, dbuf, p0, m, lane);
    else measure_pass_i8<NP, 0>(rows, pitch, qs, nbuf, dbuf, p0, m, lane);
}

// ---- the same distances with TWO lanes per row and 16-byte loads (latency form) -------------------------
// A launch that does not fill the chip is bound by how long ONE wave takes over an expansion, and measure_pass
// above issues 64 dword loads per lane for 32 rows of 128 floats: the wave's memory instructions alone (16+ cycles
// of address processing each, eight 32-byte pieces per instruction) outlast the HBM round trip several times over.
// Here lane 2r holds the AVX lanes 0-3 of row r and lane 2r + 1 the lanes 4-7: one dwordx4 load per eight elements
// and lane, 16 loads for a 128-float row, all 32 rows of an expansion in one pass; lane partial j still walks
// elements j, j + 8, ... in order with the same operations (two-wide packed where the ISA has them: v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 round each half like the scalar instruction), p_j + p_{j+4} is one exchange inside
// the lane pair (DPP quad_perm, no LDS), and the rest of the collapse tree is in-lane: EuclideanMetric.cs:45-50
// (t0 + t1) + (t2 + t3), CosineMetric.cs:145-171 (u0 + u2) + (u1 + u3).  Bit for bit the value of measure_pass.
// Rows of a multiple of 8 floats (16-byte aligned pieces); float metrics.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float dpp_pair_swap(float v) // the other lane of the pair (lane ^ 1)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true));
}
template <int METRIC>
__device__ __forceinline__ void measure_pass2(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                                              const float *qs, double sb, const int *nbuf, float *dbuf, int p0, int m, int lane)
{
    const int r = lane >> 1, h = lane & 1;
    const int c = p0 + r;
    const int id = nbuf[c < m ? c : p0]; // idle pairs shadow a valid row
    const float *a = rows + (size_t)id * dim + 4 * h;
    const float *q = qs + 4 * h;
    f32x2 acc01 = {0.0f, 0.0f}, acc23 = {0.0f, 0.0f}; // lane partials 4h + 0, 1 and 4h + 2, 3
    const int nblk = dim >> 3;
    int k = 0;
    auto step = [&](const f32x4 x, const f32x4 y) {
        const f32x2 x01 = {x.x, x.y}, x23 = {x.z, x.w}, y01 = {y.x, y.y}, y23 = {y.z, y.w};
        if (METRIC == M_SQ) {
            const f32x2 d01 = x01 - y01, d23 = x23 - y23;
            acc01 = __builtin_elementwise_fma(d01, d01, acc01);
            acc23 = __builtin_elementwise_fma(d23, d23, acc23);
        } else {
            const f32x2 p01 = x01 * y01, p23 = x23 * y23;
            acc01 = acc01 + p01;
            acc23 = acc23 + p23;
        }
    };
    for (; k + 16 <= nblk; k += 16) { // one memory round trip per 128-float chunk (see measure_pass)
        f32x4 x[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) x[kk] = *reinterpret_cast<const f32x4 *>(a + 8 * (k + kk));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) step(x[kk], *reinterpret_cast<const f32x4 *>(q + 8 * (k + kk)));
    }
    if (k + 8 <= nblk) {
        f32x4 x[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) x[kk] = *reinterpret_cast<const f32x4 *>(a + 8 * (k + kk));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) step(x[kk], *reinterpret_cast<const f32x4 *>(q + 8 * (k + kk)));
        k += 8;
    }
#pragma unroll 4
    for (; k < nblk; ++k) step(*reinterpret_cast<const f32x4 *>(a + 8 * k), *reinterpret_cast<const f32x4 *>(q + 8 * k));
    // p_j + p_{j+4}: the two lanes of the pair exchange their partials (the sum is commutative: both get t_j)
    const float t0 = acc01.x + dpp_pair_swap(acc01.x), t1 = acc01.y + dpp_pair_swap(acc01.y);
    const float t2 = acc23.x + dpp_pair_swap(acc23.x), t3 = acc23.y + dpp_pair_swap(acc23.y);
    float s;
    if (METRIC == M_SQ) { const float u = t0 + t1, v = t2 + t3; s = u + v; }
    else { const float u = t0 + t2, v = t1 + t3; s = u + v; }
    float res;
    if (METRIC == M_SQ) res = s;
    else if (METRIC == M_UCOS) res = 1.0f - s;
    else {
        const float denom = (float)(row_sn[id] * sb);
        res = (denom < 1e-30f) ? 1.0f : 1.0f - s / denom;
    }
    if (h == 0 && c < m) dbuf[c] = res;
}

template <int METRIC, bool TWO = false>
__device__ __forceinline__ void measure_all(const float *rows, const double *row_sn, int dim, const float *qs, double sb,
                                            const int *nbuf, float *dbuf, int m, int lane)
{
    if constexpr (METRIC != M_I8 && TWO) {
        if (m > 8 && (dim & 7) == 0) { // latency form: two lanes per row (up to 8 rows the eight-lane pass issues as few loads)
            for (int p0 = 0; p0 < m; p0 += 32) measure_pass2<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, p0, m, lane);
            return;
        }
    }
    if constexpr (METRIC == M_I8) {
        for (int p0 = 0; p0 < m; p0 += 32) {
            const int left = m - p0;
            if (left > 24) measure_pass_i8_any<4>(rows, dim, qs, nbuf, dbuf, p0, m, lane);
            else if (left > 16) measure_pass_i8_any<3>(rows, dim, qs, nbuf, dbuf, p0, m, lane);
            else if (left > 8) measure_pass_i8_any<2>(rows, dim, qs, nbuf, dbuf, p0, m, lane);
            else measure_pass_i8_any<1>(rows, dim, qs, nbuf, dbuf, p0, m, lane);
        }
    } else {
    for (int p0 = 0; p0 < m; p0 += 32) {
        int left = m - p0;
        if (left > 24) measure_pass<METRIC, 4>(rows, row_sn, dim, qs, sb, nbuf, dbuf, p0, m, lane);
        else if (left > 16) measure_pass<METRIC, 3>(rows, row_sn, dim, qs, sb, nbuf, dbuf, p0, m, lane);
        else if (left > 8) measure_pass<METRIC, 2>(rows, row_sn, dim, qs, sb, nbuf, dbuf, p0, m, lane);
        else measure_pass<METRIC, 1>(rows, row_sn, dim, qs, sb, nbuf, dbuf, p0, m, lane);
    }
    }
}

// The same pass against NQ vectors staged in LDS at once: every row is fetched ONCE and measured against
// all of them (D[q * ds + c] = metric(row[ids[c]], qs_q)), each (row, vector) pair in exactly the lane order of
// measure_pass -- so the bits are those of NQ separate passes, for a quarter of the row traffic and of the
// dependent round trips.  Float metrics only.
template <int METRIC, int NP, int NQ>
__device__ __forceinline__ void measure_pass_multi(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                                                   const float *q0, const float *q1, const float *q2, const float *q3, const double *sbq,
                                                   const int *ids, float *D, int ds, int p0, int m, int lane)
{
    const int grp = lane >> 3, j = lane & 7;
    const float *qs[4] = {q0, q1, q2, q3};
    const float *a[NP];
    int cidx[NP];
    float acc[NP][NQ];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int c = p0 + grp + 8 * p;
        cidx[p] = c;
        const int id = ids[c < m ? c : p0];
        a[p] = rows + (size_t)id * dim;
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[p][q] = 0.0f;
    }
    const int nblk = dim >> 3;
    int k = 0;
    for (; k + 16 <= nblk; k += 16) {
        float x[NP][16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int p = 0; p < NP; ++p) x[p][kk] = a[p][8 * (k + kk) + j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float y = qs[q][8 * (k + kk) + j];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (METRIC == M_SQ) {
                        const float d = x[p][kk] - y;
                        acc[p][q] = __builtin_fmaf(d, d, acc[p][q]);
                    } else {
                        const float pr = x[p][kk] * y;
                        acc[p][q] = acc[p][q] + pr;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0); // keeps the NQ LDS reads of one step from being hoisted over the others (registers)
        }
    }
    for (; k < nblk; ++k) {
        float xr[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) xr[p] = a[p][8 * k + j];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float y = qs[q][8 * k + j];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (METRIC == M_SQ) {
                    const float d = xr[p] - y;
                    acc[p][q] = __builtin_fmaf(d, d, acc[p][q]);
                } else {
                    const float pr = xr[p] * y;
                    acc[p][q] = acc[p][q] + pr;
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        double sa = 0.0;
        if (METRIC == M_COS) sa = row_sn[ids[cidx[p] < m ? cidx[p] : p0]];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float s = (METRIC == M_SQ) ? collapse_l2(acc[p][q]) : collapse_cos(acc[p][q]);
            if (dim & 7) s = scalar_tail<METRIC>(s, a[p], qs[q], dim);
            float r;
            if (METRIC == M_SQ) r = s;
            else if (METRIC == M_UCOS) r = 1.0f - s;
            else {
                const float denom = (float)(sa * sbq[q]);
                r = (denom < 1e-30f) ? 1.0f : 1.0f - s / denom;
            }
            if (j == 0 && cidx[p] < m) D[q * ds + cidx[p]] = r;
        }
    }
}
template <int METRIC, int NQ>
__device__ __forceinline__ void measure_multi(const float *rows, const double *row_sn, int dim, const float *q0, const float *q1,
                                              const float *q2, const float *q3, const double *sbq, const int *ids, int m, float *D, int ds, int lane)
{
#pragma nounroll
    for (int p0 = 0; p0 < m; p0 += 16) { // 16 rows x NQ vectors per pass: more rows in flight would spill (168 VGPRs)
        const int left = m - p0;
        if (left > 8) measure_pass_multi<METRIC, 2, NQ>(rows, row_sn, dim, q0, q1, q2, q3, sbq, ids, D, ds, p0, m, lane);
        else measure_pass_multi<METRIC, 1, NQ>(rows, row_sn, dim, q0, q1, q2, q3, sbq, ids, D, ds, p0, m, lane);
    }
}

} // namespace hnsw
