// device_kernels.h -- all gfx950 device code of the backend (included by device_backend.hip, which
// holds the host side, and by the traverse_*.hip units, which only instantiate the two big
// traversal kernel templates -- one metric each -- so that the build compiles them in parallel).
// See device_backend.hip's header comment for what the kernels replace and the numerical contract.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "device_backend.h"

#ifdef EXP_LAT_REGS // experiment: no occupancy target for the traversal kernels (every register the wave can have: no spills)
#define HNSW_WAVES(x) 1
#else
#define HNSW_WAVES(x) (x)
#endif

namespace hnsw {

// Every block of the kernels below that stage data through LDS is ONE wavefront working on its own job (the latency
// variants add a second wave with a role of its own, which never meets the first at a barrier): what the phases of
// such a wave need between a write and the reads of other lanes is that its own memory operations have completed and
// that the compiler keeps the order -- what __syncthreads() does in front of its s_barrier, without the barrier.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

// LDS ordering inside ONE wave (every traversal block is one wave): the wave's LDS instructions execute in order, so
// all a write-then-read by other lanes needs is that the compiler keeps them in order -- not wave_sync(), whose
// s_waitcnt also drains the vector-memory counter and with it every load still in flight.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// ------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------
enum { M_SQ = HNSWDEV_SQ_EUCLID, M_COS = HNSWDEV_COSINE, M_UCOS = HNSWDEV_UCOSINE, M_I8 = HNSWDEV_SQ_EUCLID_I8 };

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
__device__ __forceinline__ int group_sum_i32(int v)
{
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

__device__ __forceinline__ float lane_xor_add(float v, int mask) { return v + __shfl_xor(v, mask, 64); }

// Collapse of the eight lane partials, L2 order: EuclideanMetric.cs:45-50.
__device__ __forceinline__ float collapse_l2(float p)
{
    float t = lane_xor_add(p, 4); // p_j + p_{j+4}
    t = lane_xor_add(t, 1);       // (t0+t1), (t2+t3)
    t = lane_xor_add(t, 2);       // (t0+t1)+(t2+t3)
    return t;
}
// Collapse, cosine-family order: CosineMetric.cs:145-171.
__device__ __forceinline__ float collapse_cos(float p)
{
    float u = lane_xor_add(p, 4); // p_j + p_{j+4}
    u = lane_xor_add(u, 2);       // (u0+u2), (u1+u3)
    u = lane_xor_add(u, 1);       // (u0+u2)+(u1+u3)
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


// ------------------------------------------------------------------------------------
// Graph-resident search: the whole traversal of one query on one wavefront.
//
// SearchLayer / SearchLayerQuery (GraphNavigator.cs:123-256) and FindEntryAtLayer (:51-82)
// restated for a wave64.  Two variants share everything but the search state: traverse_sorted
// (further down) keeps one sorted list in registers and is what normally runs; the variant
// below keeps the two BinaryHeaps (BinaryHeap.cs:30-107) in LDS, manipulated by wave-uniform
// scalar code with the reference's exact sift rules (so the heap ARRAY, not just the heap SET,
// matches -- tie order decides ids), and is what a wave falls back to when equal distances
// make the heap layout observable; the visited set
// (VisitedListPool.cs:10-67) is a private bitset in HBM; the out-edge lists come from the HBM
// mirror of the host graph; candidate rows are measured 8 lanes per row exactly as in
// slot_distance_kernel.  Unvisited neighbours keep their adjacency order (ballot + prefix
// count), so pushes happen in the reference's order.
// ------------------------------------------------------------------------------------
struct ND {
    int id;
    float dist;
};

__device__ __forceinline__ int dev_float_compare_to(float x, float y)
{
    if (x < y) return -1;
    if (x > y) return 1;
    if (x == y) return 0;
    if (x != x) return (y != y) ? 0 : -1;
    return 1;
}
// DistanceComparer (farther first) / ReverseDistanceComparer (closer first), DistanceComparer.cs:9-25
template <bool CLOSER>
__device__ __forceinline__ int nd_cmp(ND x, ND y)
{
    if (CLOSER) {
        if (x.dist > y.dist) return -1;
        if (x.dist < y.dist) return 1;
        return dev_float_compare_to(y.dist, x.dist);
    }
    if (x.dist < y.dist) return -1;
    if (x.dist > y.dist) return 1;
    return dev_float_compare_to(x.dist, y.dist);
}
// Heap entries on the device are {id, key}: key = the distance's float bits mapped to an
// unsigned integer with the same order (sign flip).  For every float except NaN and -0 the
// integer order IS the float.CompareTo order the reference's comparers use
// (DistanceComparer.cs:9-25), equal keys <=> equal distances, so every sift decision -- ties
// included -- is unchanged; a traversal that meets a NaN or -0 distance is flagged and re-run on
// the host path, where the comparers are restated literally.  Why keys: every value below is
// wave-uniform; with integer keys pulled through readfirstlane the whole heap logic compiles to
// SCALAR compares and branches (no exec-mask juggling), ~5x fewer instructions per sift level
// than float compares on "divergent" VGPRs -- and this serial code, not memory, was the
// bottleneck of the traversal kernels.
__device__ __forceinline__ unsigned f2key(float d)
{
    unsigned u = __float_as_uint(d);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__device__ __forceinline__ bool key_unsafe(float d) { return d != d || __float_as_uint(d) == 0x80000000u; } // NaN or -0

struct HEnt {
    int id;
    unsigned key;
};
__device__ __forceinline__ HEnt uniform_ent(int2 v) // two 32-bit scalars (keeps the key compares on s_cmp_*_u32)
{
    HEnt e;
    e.id = __builtin_amdgcn_readfirstlane(v.x);
    e.key = (unsigned)__builtin_amdgcn_readfirstlane(v.y);
    return e;
}
__device__ __forceinline__ int2 pack_ent(HEnt e) { return make_int2(e.id, (int)e.key); }

// `top` lives entirely in LDS; `cand` keeps its first `cap` entries in LDS and spills the
// (rarely reached) deep leaves to a private HBM area, so the LDS footprint -- and with it the
// number of resident waves -- is set by the common case, not the worst one.
struct LdsHeap {
    ND *b;
    __device__ __forceinline__ int2 lane_get(int i) const { return *reinterpret_cast<const int2 *>(b + i); }
    __device__ __forceinline__ void lane_set(int i, int2 v) const { *reinterpret_cast<int2 *>(b + i) = v; }
    __device__ __forceinline__ HEnt get(int i) const { return uniform_ent(*reinterpret_cast<const int2 *>(b + i)); }
    __device__ __forceinline__ void set(int i, HEnt v) const { *reinterpret_cast<int2 *>(b + i) = pack_ent(v); }
    // both children in one LDS round trip (entry i + 1 may be one past the heap: never used then)
    __device__ __forceinline__ void get2(int i, HEnt &x, HEnt &y) const
    {
        const int2 *p = reinterpret_cast<const int2 *>(b + i);
        const int2 vx = p[0], vy = p[1];
        x = uniform_ent(vx);
        y = uniform_ent(vy);
    }
};
struct SpillHeap {
    ND *b;
    int cap;
    ND *g;
    // per-lane (divergent) access for the wave-parallel pop
    __device__ __forceinline__ int2 lane_get(int i) const { return i < cap ? *reinterpret_cast<const int2 *>(b + i) : *reinterpret_cast<const int2 *>(g + (i - cap)); }
    __device__ __forceinline__ void lane_set(int i, int2 v) const
    {
        if (i < cap) *reinterpret_cast<int2 *>(b + i) = v;
        else *reinterpret_cast<int2 *>(g + (i - cap)) = v;
    }
    __device__ __forceinline__ HEnt get(int i) const
    {
        return uniform_ent(i < cap ? *reinterpret_cast<const int2 *>(b + i) : *reinterpret_cast<const int2 *>(g + (i - cap)));
    }
    __device__ __forceinline__ void set(int i, HEnt v) const
    {
        if (i < cap) *reinterpret_cast<int2 *>(b + i) = pack_ent(v);
        else *reinterpret_cast<int2 *>(g + (i - cap)) = pack_ent(v);
    }
    __device__ __forceinline__ void get2(int i, HEnt &x, HEnt &y) const
    {
        if (i + 1 < cap) {
            const int2 *p = reinterpret_cast<const int2 *>(b + i);
            const int2 vx = p[0], vy = p[1];
            x = uniform_ent(vx);
            y = uniform_ent(vy);
        } else {
            x = get(i);
            y = get(i + 1); // i + 1 <= count <= cap + spill_cap - 1: inside the spill area
        }
    }
};
// comparer outcomes on keys: FartherFirst cmp(x,y) = sign(kx - ky); CloserFirst the reverse
template <bool CLOSER> __device__ __forceinline__ bool cmp_le0(HEnt x, HEnt y) { return CLOSER ? x.key >= y.key : x.key <= y.key; }
template <bool CLOSER> __device__ __forceinline__ bool cmp_lt0(HEnt x, HEnt y) { return CLOSER ? x.key > y.key : x.key < y.key; }

template <bool CLOSER, class H>
__device__ __forceinline__ void heap_push(const H &h, int &count, HEnt item) // BinaryHeap.cs:30-34, :89-107
{
    int i = count++;
    while (i > 0) {
        int p = (i - 1) >> 1;
        HEnt parent = h.get(p);
        if (cmp_le0<CLOSER>(item, parent)) break;
        h.set(i, parent);
        i = p;
    }
    h.set(i, item);
}
template <bool CLOSER, class H>
__device__ __forceinline__ HEnt heap_pop(const H &h, int &count) // BinaryHeap.cs:53-87
{
    HEnt result = h.get(0);
    int n = --count;
    HEnt item = h.get(n);
    if (n != 0) {
        int i = 0, half = n >> 1;
        while (i < half) {
            int left = (i << 1) + 1, right = left + 1;
            HEnt mv, rv;
            h.get2(left, mv, rv);
            int mc = left;
            if (right < n && cmp_lt0<CLOSER>(mv, rv)) { mc = right; mv = rv; }
            if (cmp_le0<CLOSER>(mv, item)) break;
            h.set(i, mv);
            i = mc;
        }
        h.set(i, item);
    }
    return result;
}

// heap_pop with the wave's lanes side by side -- the same array afterwards, entry for entry.  The scalar loop above
// pays one LDS round trip per level (children, compare, branch), nine levels deep in a candidate heap; but WHICH child
// a node hands up (:76-77: the right one only if the left compares below it) does not depend on the item that sinks,
// so the whole root-to-leaf chain of those choices can be read off in parallel: 63 lanes load the child pairs of a
// six-level subtree, one ballot holds their choices, six scalar steps follow them, and the next subtree starts where
// they end.  Then one lane per level of that chain loads its entry, a ballot finds where the item stops (:79), and the
// entries above move up one level together.  Three to four round trips instead of seven to ten: the exact traversal
// of a 1M-node graph took 1.3 ms on an idle chip against the sorted one's 0.45, nearly all of it in these loops --
// and the exact traversal is what a launch's last jobs wait for (graph_search_kernel, shadows).
template <bool CLOSER, class H>
__device__ __forceinline__ HEnt heap_pop_wave(const H &h, int &count, int lane) // BinaryHeap.cs:53-87
{
    const HEnt result = h.get(0);
    const int n = --count;
    if (n == 0) return result;
    const HEnt item = h.get(n);
    const int half = n >> 1; // nodes below `half` have a left child (:70)
    // the chain of chosen children from the root: lane d keeps the node of depth d + 1
    int v_path = 0, depth = 0;
    {
        int cur = 0;                                                     // root of the subtree looked at
        const int l = 31 - __builtin_clz(lane + 1), o = lane + 1 - (1 << l); // this lane's place in it: level, offset
        while (cur < half) {
            const int node = ((cur + 1) << l) - 1 + o;
            const bool inner = lane < 63 && node < half;
            bool right = false;
            if (inner) {
                const int2 lv = h.lane_get(2 * node + 1);
                if (2 * node + 2 < n) {
                    const int2 rv = h.lane_get(2 * node + 2);
                    right = cmp_lt0<CLOSER>(HEnt{lv.x, (unsigned)lv.y}, HEnt{rv.x, (unsigned)rv.y}); // :76-77
                }
            }
            const unsigned long long rm = __ballot(right), im = __ballot(inner);
            int j = 0, nd = cur;
#pragma unroll
            for (int lev = 0; lev < 6; ++lev) {
                if (!((im >> j) & 1ull)) break;
                const int bit = (int)((rm >> j) & 1ull);
                nd = 2 * nd + 1 + bit;
                j = 2 * j + 1 + bit;
                if (lane == depth) v_path = nd;
                ++depth;
            }
            if (nd == cur) break;
            cur = nd;
            if (j < 63) break; // the chain ended inside this subtree (a node without children)
        }
    }
    // where does the item stop?  (:79: at the first chosen child that does not compare above it)
    bool stops = false;
    int2 mine = make_int2(0, 0);
    if (lane < depth) {
        mine = h.lane_get(v_path);
        stops = cmp_le0<CLOSER>(HEnt{mine.x, (unsigned)mine.y}, item);
    }
    const unsigned long long sm = __ballot(stops);
    const int s = sm ? (int)__builtin_ctzll(sm) : depth; // levels the item sinks
    if (lane < s) h.lane_set((v_path - 1) >> 1, mine);   // :80-81, all levels at once
    const int at = s > 0 ? __builtin_amdgcn_readlane(v_path, s - 1) : 0;
    if (lane == 0) h.lane_set(at, pack_ent(item));       // :84
    return result;
}

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

// int8 records: NP candidates per lane group, every load of the pass issued before any arithmetic (one
// memory round trip for up to 8 * NP records); qs = the query's record staged in LDS.
template <int NP, int NB>
__device__ __forceinline__ void measure_pass_i8(const float *__restrict__ rows, int pitch, const float *qs, const int *nbuf, float *dbuf,
                                                int p0, int m, int lane)
{
    const int grp = lane >> 3, j = lane & 7;
    const int *a[NP];
    int cidx[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int c = p0 + grp + 8 * p;
        cidx[p] = c;
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
    const int g6 = (lane & ~7) | 6, g7 = (lane & ~7) | 7;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int dot = group_sum_i32(acc[p]);
        const float sa = __int_as_float(__shfl(tr[p], g6, 64));
        const int na = __shfl(tr[p], g7, 64);
        const float r = i8_epilogue(sa, na, sq, nq, dot);
        if (j == 0 && cidx[p] < m) dbuf[cidx[p]] = r;
    }
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

constexpr int kNewMax = 4;     // link kernel shortcut: new entries of an overflowing list measured against all others
constexpr int kSpillCap = 8192; // candidate-heap entries per traversal that may spill to HBM

// LDS carve-up shared by the traversal kernels
struct SearchLds {
    ND *top;    // k + 1
    ND *cand;   // cand_cap
    float *qs;  // dim (padded to 4)
    float *qs2; // dim (padded to 4): second vector (heuristic / prune)
    float *qs3; // dim (padded to 4): the heuristic's next candidate, staged while the current one is tested
    int *nbuf;  // nbcap
    float *dbuf; // nbcap
    int *acc;   // nbcap: accepted ids of the heuristic
    int *stk;   // 3 * 40: introsort work stack
};
// heur: also room for the heuristic (second vector, accepted ids, introsort stack)
// nbcap: capacity of the id / distance scratch = longest adjacency list, rounded up to 8
__host__ __device__ inline size_t search_lds_bytes(int k, int cand_cap, int dim, bool heur, int nbcap)
{
    size_t b = ((sizeof(ND) * (size_t)(k + 1 + cand_cap) + 15u) & ~(size_t)15u) + sizeof(float) * (size_t)((dim + 3) & ~3) + 2u * 4u * (size_t)nbcap;
    if (heur) b += 2u * sizeof(float) * (size_t)((dim + 3) & ~3) + 4u * (size_t)nbcap + 4u * 3u * 40u;
    return b;
}
__device__ __forceinline__ SearchLds carve_lds(unsigned char *smem, int k, int cand_cap, int dim, int nbcap)
{
    SearchLds L;
    L.top = reinterpret_cast<ND *>(smem);
    L.cand = L.top + (k + 1);
    L.qs = reinterpret_cast<float *>(smem + ((sizeof(ND) * (size_t)(k + 1 + cand_cap) + 15u) & ~(size_t)15u)); // 16-byte aligned: read in 16-byte pieces (measure_pass2)
    L.nbuf = reinterpret_cast<int *>(L.qs + ((dim + 3) & ~3));
    L.dbuf = reinterpret_cast<float *>(L.nbuf + nbcap);
    // heuristic-only regions (present when the launch sized LDS with heur = true)
    L.qs2 = L.dbuf + nbcap;
    L.qs3 = L.qs2 + ((dim + 3) & ~3);
    L.acc = reinterpret_cast<int *>(L.qs3 + ((dim + 3) & ~3));
    L.stk = L.acc + nbcap;
    return L;
}

struct GraphView {
    const int *adj0;
    int stride0;
    const int64_t *upper;
    const int *pool;
    int strideU;
    __device__ __forceinline__ const int *list(int id, int layer) const
    {
        return layer == 0 ? adj0 + (size_t)id * stride0 : pool + upper[id] + (size_t)(layer - 1) * strideU;
    }
};

// A wave's visited set (VisitedListPool.cs:10-67 restated for one in-flight traversal), empty
// between jobs.  Up to 4M nodes: a bitset over node ids in HBM, cleared by streaming over it.
// Above: an open-addressing hash table of the visited ids (tab != nullptr, entries -1 when empty),
// 64 KB per wave whatever the graph size -- at 10M nodes the bitsets of all resident waves span
// gigabytes, and streaming a 1.25-MB clear per traversal cost as much as the row reads (measured:
// 0.98 M queries/s streaming, 1.28 M clearing through a log of the ids, 1.48 M with the table; at 1M
// nodes the bitset wins, 2.5 M against 1.9 M).  `seen` counts insertions; beyond `limit` the
// traversal is handed back to the host, so the table never fills.
template <bool HASHED> // compile-time choice: the bitset kernels carry none of the table's code or registers
struct VisitedSet {
    unsigned *bits;
    long long words; // multiple of 4; the arena is 16-byte aligned
    int *tab;
    unsigned tab_mask;
    int seen, limit;
    // true: id was not in the set (and now is).  Per lane; lists hold no duplicates.
    __device__ __forceinline__ bool first_visit(int id)
    {
        if constexpr (!HASHED) {
            const unsigned bit = 1u << (id & 31);
            return (atomicOr(&bits[id >> 5], bit) & bit) == 0u;
        }
        unsigned h = ((unsigned)id * 2654435761u) & tab_mask;
        for (unsigned probes = 0; probes <= tab_mask; ++probes) {
            const int old = atomicCAS(&tab[h], -1, id);
            if (old == -1) return true;
            if (old == id) return false;
            h = (h + 1) & tab_mask;
        }
        return true; // table full (the host sizes it so that crowded() fires long before): the job is handed back, never stuck
    }
    __device__ __forceinline__ bool crowded() const { return HASHED && seen > limit; }
    __device__ __forceinline__ void clear(int lane)
    {
        wave_sync();
        if constexpr (HASHED) {
            uint4 *t4 = reinterpret_cast<uint4 *>(tab);
            const uint4 e = make_uint4(~0u, ~0u, ~0u, ~0u);
            for (unsigned w = lane; w < ((tab_mask + 1u) >> 2); w += 64) t4[w] = e;
        } else {
            uint4 *v4 = reinterpret_cast<uint4 *>(bits);
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            for (long long w = lane; w < (words >> 2); w += 64) v4[w] = z;
        }
        seen = 0;
        wave_sync();
    }
};

#ifdef EXP_PHASE_CLOCKS // experiment build: shader-clock cycles per traversal phase, summed over waves
// The counters live in ONE device buffer owned by the host unit; every translation unit keeps a pointer to it in a
// device global of its own, bound by that unit's hnsw_phase_bind_<unit>() (device_backend.hip calls them all).
static __device__ unsigned long long *g_phase_ptr;
#define g_phase (g_phase_ptr)             // [12]
#define g_phase_link (g_phase_ptr + 12)   // [12]
#define g_phase_x (g_phase_ptr + 24)      // [16] finer split of an expansion (traverse_sorted)
constexpr int kPhaseWords = 40;
static inline hipError_t hnsw_phase_bind_tu(unsigned long long *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ptr), &p, sizeof p); }
#define HNSW_PHASE_BIND(UNIT) extern "C" hipError_t hnsw_phase_bind_##UNIT(unsigned long long *p) { return hnsw::hnsw_phase_bind_tu(p); }
#define PH_FLUSH_LINK() do { if (lane == 0) for (int ph_i = 0; ph_i < 8; ++ph_i) atomicAdd(&g_phase_link[ph_i], (unsigned long long)ph_acc[ph_i]); } while (0)
#define PH_DECL() long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long ph_x[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long ph_t = __builtin_readcyclecounter()
#define PH(i) do { long long ph_n = __builtin_readcyclecounter(); ph_acc[i] += ph_n - ph_t; ph_t = ph_n; } while (0)
#define PHX(i) do { long long ph_n = __builtin_readcyclecounter(); ph_x[i] += ph_n - ph_t; ph_acc[4] += ph_n - ph_t; ph_t = ph_n; } while (0)
#define PHX_COUNT(i, v) ph_x[i] += (v)
#define PHY(i) do { long long ph_n = __builtin_readcyclecounter(); ph_x[i] += ph_n - ph_t; ph_acc[5] += ph_n - ph_t; ph_t = ph_n; } while (0)
#define PH_COUNT(i, v) ph_acc[i] += (v)
#define PH_FLUSH() do { if (lane == 0) { for (int ph_i = 0; ph_i < 8; ++ph_i) atomicAdd(&g_phase[ph_i], (unsigned long long)ph_acc[ph_i]); for (int ph_i = 0; ph_i < 16; ++ph_i) atomicAdd(&g_phase_x[ph_i], (unsigned long long)ph_x[ph_i]); } } while (0)
#else
#define PH_DECL() do {} while (0)
#define PHX(i) do {} while (0)
#define PHX_COUNT(i, v) do {} while (0)
#define PHY(i) do {} while (0)
#define HNSW_PHASE_BIND(UNIT)
#define PH(i) do {} while (0)
#define PH_COUNT(i, v) do {} while (0)
#define PH_FLUSH() do {} while (0)
#define PH_FLUSH_LINK() do {} while (0)
#endif

// Read log of the reference-exact windowed Add (hnsw_index.cpp "exact window"): the adjacency lists one insert's
// searches READ -- the node whose out-edges a descent pass scans (GraphNavigator.cs:65) and every candidate a
// beam search expands (:152-156) -- in order, a marker -(layer + 1) in front of each layer's entries.  These
// lists (and the stored rows, which never change) are all a search depends on, so a result computed on an older
// snapshot of the graph is still the sequential one while none of them has been written since.  p == nullptr
// (every other caller): nothing is recorded and the code folds away.  n keeps counting beyond cap: the host
// sees the overflow.
// Every entry is a pair: the node (or marker) and, for a beam-search expansion, the key of the farthest result at that
// moment if the result list was full (0xffffffff otherwise, and for descent passes and markers): a neighbour whose distance
// key is not below it would not have been pushed by that expansion -- which lets the host tell that a list which did change
// since the snapshot changed in a way this reader would not have noticed (hnsw_index.cpp, "a change the reader does not see").
struct ReadLog {
    int *p;
    int n, cap; // in entries (pairs)
    __device__ __forceinline__ void put(int v, int lane, unsigned far = 0xffffffffu)
    {
        if (p) {
            if (lane == 0 && n < cap) { p[2 * n] = v; p[2 * n + 1] = (int)far; }
            n++;
        }
    }
    __device__ __forceinline__ void layer(int l, int lane) { put(-(l + 1), lane); }
};

// FindEntryPoint / FindEntryAtLayer (GraphNavigator.cs:27-82): greedy descent from jb.entry at
// jb.entry_layer down to (not including) jb.search_layer.  Leaves the entry of the search layer
// in `best` and its distance in `cur` (both wave-uniform).
template <int METRIC, bool TWO = false>
__device__ __forceinline__ void descend(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, double sb,
                                        const GraphView &G, const SearchJob jb, const SearchLds &L, int lane, int &best, float &cur,
                                        unsigned long long &evals, ReadLog &RL)
{
    int *nbuf = L.nbuf;
    float *dbuf = L.dbuf;
    const float *qs = L.qs;
    best = jb.entry;
    wave_sync();
    if (lane == 0) nbuf[0] = best;
    wave_sync();
    measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, 1, lane);
    wave_sync();
    cur = dbuf[0]; // :57
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
            if (n > 0) measure_all<METRIC, TWO>(rows, row_sn, dim, qs, sb, nbuf, dbuf, n, lane);
            wave_sync();
            evals += (unsigned long long)n;
            for (int i = 0; i < n; ++i) { // :67-78
                float d = dbuf[i];
                if (d < cur) { cur = d; best = nbuf[i]; changed = true; }
            }
        }
    }
    best = __builtin_amdgcn_readfirstlane(best);
    cur = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cur)));
}

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

template <int METRIC, int NS, bool HASHED>
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
    const bool novis = (oflags & 8) != 0;
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
        const bool overlapped = (oflags & 1) != 0 && n <= 64; // oflags bit 0: rows requested with the visited atomics
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
            if (n > 0) measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, n, lane); // :163 (and the visited ones)
            wave_sync();
            lane_d = in ? dbuf[lane] : 0.0f;
            lane_id = nb_a;
            if (novis) {
                {
                    // every listed neighbour counts as new -- except the ones that are in the list: only a key that could pass
                    // the push test or meet the farthest key matters to anything below, so only those are looked up
                    have = in;
                    const unsigned kq = f2key(lane_d);
                    unsigned long long look = __ballot(in && (top_n < k || kq <= far_key));
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
            PH(4);
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
    descend<METRIC, true>(rows, row_sn, dim, sb, G, jb, L, lane, best, cur, evals, RL);
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

// NS > 0: sorted-list traversal with NS register sets (k <= 64 * NS); a wave that meets equal
// distances where the heap layout shows starts over with the exact two-heap traversal (out_flag 2,
// informational).  NS = 0: two-heap traversal only.
// One job on this wave.  `vis` / `spill`: the wave's own scratch (vis all zero on entry; the caller
// clears it afterwards).
// Job words of a launch with SHADOW traversals (graph_search_kernel): bit 0 answered (results written), bit 1 the
// wave that owns the job met a tie, bit 2 a shadow traversal has been started for it.
constexpr int kJobAnswered = 1, kJobTied = 2, kJobShadowed = 4;

template <int METRIC, int NS, bool HASHED, bool LAT = false>
__device__ __forceinline__ void search_job(const float *__restrict__ rows, const double *__restrict__ row_sn, const float *__restrict__ queries,
                    const double *__restrict__ q_sn, int dim, const int *__restrict__ adj0, int stride0,
                    const int64_t *__restrict__ upper, const int *__restrict__ pool, int strideU,
                    const SearchJob *__restrict__ jobs, int k, int cand_cap, ND *__restrict__ spill,
                    int spill_cap, VisitedSet<HASHED> &V, int k_out, int *__restrict__ out_ids,
                    float *__restrict__ out_d, int *__restrict__ out_cnt, int *__restrict__ out_flag,
                    unsigned long long *__restrict__ eval_counter, int nbcap, unsigned char *smem, int job, int overlap,
                    int *__restrict__ job_word = nullptr, bool shadow = false, TeamPort *port = nullptr, bool *v_untouched = nullptr)
{
    const SearchLds L = carve_lds(smem, k, cand_cap, dim, nbcap);
    const int lane = threadIdx.x & 63;
    if (v_untouched) *v_untouched = false;
    const SearchJob jb = jobs[job];
    const GraphView G{adj0, stride0, upper, pool, strideU};

    const float *q;
    double sb = 0.0;
    if (jb.qref >= 0) {
        q = queries + (size_t)jb.qref * dim;
        if (METRIC == M_COS) sb = q_sn[jb.qref];
    } else {
        q = rows + (size_t)(~jb.qref) * dim;
        if (METRIC == M_COS) sb = row_sn[~jb.qref];
    }
    for (int i = lane; i < dim; i += 64) L.qs[i] = q[i];
    if constexpr (LAT) { if (lane == 0) port->m->sb = sb; } // (the memory wave reads both after the first request's release)
    unsigned long long evals = 0;
    int top_n = 0;
    bool repeated = shadow;
    ReadLog RL{nullptr, 0, 0};
    // With shadows, whoever sets kJobAnswered first writes the job's results (both traversals compute the same ones).
    auto claim_answer = [&]() -> bool {
        if (!job_word) return true;
        int old = 0;
        if (lane == 0) old = atomicOr(job_word, kJobAnswered);
        return (__builtin_amdgcn_readfirstlane(old) & kJobAnswered) == 0;
    };
    if constexpr (NS > 0) {
        if (jb.aux != -2 && !shadow) {
        bool tie = false;
        // OrderBy + Take(k_out) reads k_out entries in order and decides between entries k_out - 1 and k_out
        bool window = false;
        bool ok1;
        if constexpr (LAT) ok1 = traverse_pool<METRIC, NS, HASHED>(rows, row_sn, dim, sb, G, jb, k, k_out + 1, V, L, lane, top_n, tie, evals, RL, nullptr, &window, port);
        else ok1 = traverse_sorted<METRIC, NS, HASHED>(rows, row_sn, dim, sb, G, jb, k, k_out + 1, V, L, lane, top_n, tie, evals, overlap, RL, nullptr, &window);
        if (!(ok1 && tie)) {
            if (v_untouched) *v_untouched = !LAT && (overlap & 8) != 0; // the sorted traversal ran without a visited set: nothing to clear
            if (!claim_answer()) return;
            // KnnQuery's tail (HNSWIndex.cs:119-123): OrderBy(Dist).Take(k) of distinct distances is the
            // head of the ascending list; missing results are padded (HNSWIndexExports.cs:144)
            for (int r = lane; r < k_out; r += 64) {
                const bool have = r < top_n;
                out_ids[(size_t)job * k_out + r] = have ? L.top[r].id : -1;
                out_d[(size_t)job * k_out + r] = have ? L.top[r].dist : __uint_as_float(0x7fc00000u);
            }
            if (lane == 0) {
                out_cnt[job] = ok1 ? top_n : 0;
                out_flag[job] = ok1 ? (window ? 4 : 0) : 1; // 4: informational (a group window of equal distances closed cleanly)
                atomicAdd(eval_counter, evals);
            }
            return;
        }
        // equal distances where the heap layout shows: the exact traversal answers this job -- the shadow that an
        // idle wave has already started for it (see graph_search_kernel), or this wave, starting over
        if (job_word) {
            int old = 0;
            if (lane == 0) old = atomicOr(job_word, kJobTied);
            if (__builtin_amdgcn_readfirstlane(old) & (kJobShadowed | kJobAnswered)) return;
        }
        V.clear(lane);
        evals = 0;
        top_n = 0;
        repeated = true;
        }
    }
    bool aborted = false;
    const bool ok = traverse<METRIC, HASHED>(rows, row_sn, dim, sb, G, jb, k, cand_cap, spill, spill_cap, V, L, lane, top_n, evals, RL,
                                             shadow ? job_word : nullptr, &aborted, LAT || (overlap & 1) != 0 || repeated);
    if (aborted || !claim_answer()) return;
    if (jb.aux == -2) { // SearchLayer's own return value: topCandidates.ToArray(), the heap's array (BinaryHeap.cs:41-44)
        wave_sync();
        for (int r = lane; r < k_out; r += 64) {
            const bool have = ok && r < top_n;
            out_ids[(size_t)job * k_out + r] = have ? L.top[r].id : -1;
            out_d[(size_t)job * k_out + r] = have ? L.top[r].dist : __uint_as_float(0x7fc00000u);
        }
        if (lane == 0) {
            out_cnt[job] = ok ? top_n : 0;
            out_flag[job] = ok ? 0 : 1;
            atomicAdd(eval_counter, evals);
        }
        return;
    }
    // KnnQuery's tail (HNSWIndex.cs:119-123): OrderBy(c => c.Dist) is a STABLE sort over the heap
    // array (ToArray(), BinaryHeap.cs:41-44) and only the first k_out survive -- so select the
    // k_out smallest (float.CompareTo order: NaN first, -0 == +0) with ties broken by array index:
    // exactly the stable sort's prefix.  Key = (order-preserving bits << 32) | index, wave min.
    wave_sync();
    unsigned long long used = 0; // bit t: entry lane + 64*t already emitted
    for (int r = 0; r < k_out; ++r) {
        unsigned long long best = ~0ull;
        for (int t = 0, i = lane; i < top_n; ++t, i += 64) {
            if ((used >> t) & 1ull) continue;
            float d = L.top[i].dist;
            unsigned u;
            if (d != d) u = 0u;                      // NaN sorts first
            else {
                if (d == 0.0f) d = 0.0f;             // -0 and +0 compare equal
                u = __float_as_uint(d);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                if (u == 0u) u = 1u;                 // keep NaN's key unique (only -NaN-like bit patterns reach 0)
            }
            unsigned long long key = ((unsigned long long)u << 32) | (unsigned)i;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            unsigned long long o = __shfl_xor(best, off, 64);
            best = o < best ? o : best;
        }
        if (best == ~0ull) { // fewer than k_out results: pad (HNSWIndexExports.cs:144)
            if (lane == 0) { out_ids[(size_t)job * k_out + r] = -1; out_d[(size_t)job * k_out + r] = __uint_as_float(0x7fc00000u); }
            continue;
        }
        const int wi = (int)(best & 0xffffffffu);
        if ((wi & 63) == lane) used |= 1ull << (wi >> 6);
        if (lane == 0) { ND w = L.top[wi]; out_ids[(size_t)job * k_out + r] = w.id; out_d[(size_t)job * k_out + r] = w.dist; }
    }
    if (lane == 0) {
        out_cnt[job] = ok ? top_n : 0;
        out_flag[job] = ok ? (repeated ? 2 : 0) : 1; // 2: informational (answered by the exact traversal)
        atomicAdd(eval_counter, evals);
    }
}

// Persistent launch: one wave per block, as many blocks as stay resident; each takes jobs from a
// shared counter until none are left.  A wave owns one visited bitset and one spill area for the
// whole launch and leaves the bitset clean after every job, so the scratch is sized by the
// resident waves (not by the batch) and nothing is memset between launches.
template <int METRIC, int NS, bool HASHED, bool LAT = false>
// float rows: 168 VGPRs, three waves per SIMD; int8 records keep 16 registers of rows in flight, not 64: five waves.
// LAT (launches that do not fill the chip): no occupancy to buy -- every spilled register is a memory round trip a lone wave
// waits out in full -- so two waves per SIMD at most (256 VGPRs), one with eight register sets
__global__ void __launch_bounds__(LAT ? 128 : 64) __attribute__((amdgpu_waves_per_eu(HNSW_WAVES(LAT ? (NS <= 4 ? 2 : 1) : METRIC == M_I8 ? (NS <= 2 ? 5 : 4) : (NS <= 4 ? 3 : 2)))))
graph_search_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, const float *__restrict__ queries,
                    const double *__restrict__ q_sn, int dim, const int *__restrict__ adj0, int stride0,
                    const int64_t *__restrict__ upper, const int *__restrict__ pool, int strideU,
                    const SearchJob *__restrict__ jobs, int k, int cand_cap, ND *__restrict__ spill,
                    int spill_cap, unsigned *__restrict__ visited, long long vis_words, int *__restrict__ vis_tab, int vis_tab_cap, int k_out,
                    int *__restrict__ out_ids, float *__restrict__ out_d, int *__restrict__ out_cnt, int *__restrict__ out_flag,
                    unsigned long long *__restrict__ eval_counter, int nbcap, int njobs, int *__restrict__ job_counter, int overlap,
                    const int *__restrict__ ready)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    VisitedSet<HASHED> V{visited + (size_t)blockIdx.x * (size_t)vis_words, vis_words,
                 vis_tab ? vis_tab + (size_t)blockIdx.x * (size_t)vis_tab_cap : nullptr, (unsigned)(vis_tab_cap - 1), 0, vis_tab_cap / 4 * 3};
    ND *my_spill = spill + (size_t)blockIdx.x * spill_cap;

    TeamPort port{nullptr, 0, 0};
    if constexpr (LAT) {
        // two waves per block (see TeamMail): wave 1 serves the expansions, wave 0 is the traversal.  The mailbox follows
        // the traversal's LDS; its sequence words are zeroed before the roles part (the one barrier both waves meet at).
        TeamMail *mail = reinterpret_cast<TeamMail *>(smem + ((search_lds_bytes(k, cand_cap, dim, false, nbcap) + 15) & ~(size_t)15));
        if (threadIdx.x == 0) { mail->req_seq = 0; mail->rsp_seq = 0; mail->hint_node = -1; }
        __syncthreads();
        if (threadIdx.x >= 64) {
            const GraphView G{adj0, stride0, upper, pool, strideU};
            const SearchLds L = carve_lds(smem, k, cand_cap, dim, nbcap);
            memory_wave<METRIC, HASHED>(rows, row_sn, dim, G, V, L.qs, mail, lane);
            return;
        }
        port.m = mail;
    }
    // `ready` (hnsw_knn_query on host buffers): the launch started when the first rows of the query set had landed; the
    // rest is still arriving on the copy engine, and *ready (a word in host memory the uploading thread advances) says how
    // many rows are there.  Jobs are taken in order, so a wave almost never has to wait; when it does it sleeps and
    // polls, for a bounded time -- a job whose row has not arrived by then is handed back (flag 1), never waited for.
    // SHADOW traversals (overlap bit 8; job_counter then is [next job, next shadow, -, -, one word per job ...], all zero
    // at launch).  One traversal in 700 meets equal distances where the heap layout shows and starts over in the exact
    // two-heap form, three times as long as the sorted one; whenever that happened to one of the LAST jobs of a launch,
    // the whole launch waited for it -- 7 % of a 65 536-query launch at C2, 17-35 % of the 12 500-query launches
    // (measured with the re-runs compiled out).  So a wave that finds the queue empty does not leave: it starts the
    // exact traversal of a job another wave is still working on, latest job first.  Almost always the owner answers the
    // job soon after and the shadow stops at its next expansion; when the owner meets a tie it finds the exact
    // traversal already under way and leaves it to the shadow.  Results are written by whoever finishes first -- both
    // compute the reference's answer.
    const bool shadows = (overlap & 0x100) != 0 && NS > 0;
    int *job_words = job_counter + 4;
    int known_ready = 0;
    bool v_clean = false;
    for (;;) {
        int job = 0;
        bool shadow = false;
        if (lane == 0) job = atomicAdd(job_counter, 1);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= njobs) {
            if (!shadows) break;
            int t = 0;
            if (lane == 0) t = atomicAdd(job_counter + 1, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= njobs || t >= (int)gridDim.x) break; // only the last gridDim.x jobs can still be running
            job = njobs - 1 - t;
            if (ready) {
                // a gated launch (query rows still arriving): no shadow for a job whose row has not landed -- its owner is
                // asleep at the gate and search_job would read whatever the previous call left in that row
                const int need = __builtin_amdgcn_readfirstlane(jobs[job].qref);
                if (need >= known_ready) {
                    int r = 0;
                    if (lane == 0) r = __hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    known_ready = __builtin_amdgcn_readfirstlane(r);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    if (need >= known_ready) continue;
                }
            }
            int old = 0;
            if (lane == 0) old = atomicOr(job_words + job, kJobShadowed);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old & (kJobAnswered | kJobTied)) continue; // answered, or its owner is already starting over
            shadow = true;
        } else if (ready) {
            const int need = __builtin_amdgcn_readfirstlane(jobs[job].qref);
            if (need >= known_ready) {
                // a read of host memory per poll: few polls, far apart (thousands of waves polling back to back were
                // measured to starve the very copy they wait for) -- 512 x ~0.2 ms at most, then the job is handed back
                for (int spin = 0; spin < 512; ++spin) {
                    int r = 0;
                    if (lane == 0) r = __hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    known_ready = __builtin_amdgcn_readfirstlane(r);
                    if (need < known_ready) break;
                    for (int z = 0; z < 48; ++z) __builtin_amdgcn_s_sleep(127);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the rows read next were written by the copy engine
                if (need >= known_ready) {
                    // handed back.  With shadows the job word decides who answers, exactly as for a tie (search_job): a
                    // shadow that started because the row landed meanwhile keeps the job; otherwise this wave claims it.
                    int old = 0;
                    if (shadows && lane == 0) {
                        old = atomicOr(job_words + job, kJobTied);
                        if (!(old & (kJobShadowed | kJobAnswered))) old = atomicOr(job_words + job, kJobAnswered) & kJobAnswered;
                    }
                    old = __builtin_amdgcn_readfirstlane(old);
                    if (old == 0 && lane == 0) {
                        out_cnt[job] = 0;
                        out_flag[job] = 1;
                    }
                    continue;
                }
            }
        }
        search_job<METRIC, NS, HASHED, LAT>(rows, row_sn, queries, q_sn, dim, adj0, stride0, upper, pool, strideU, jobs, k, cand_cap, my_spill, spill_cap,
                               V, k_out, out_ids, out_d, out_cnt, out_flag, eval_counter, nbcap, smem, job, overlap, shadows ? job_words + job : nullptr,
                               shadow, &port, &v_clean);
        if (!v_clean) V.clear(lane);
    }

    if constexpr (LAT) port.post(-1, 0, lane); // the memory wave leaves
}

#ifdef HNSW_HOST_TU // few variants and launched from one place: defined only in the unit that launches it
// RangeQuery on the device: FindEntryPointQuery + GraphNavigator.SearchLayerRange (GraphNavigator.cs:262-325)
// for one query per wave.  What the reference's two heaps compute there is a closure: a neighbour enters
// `candidates` and `topCandidates` iff its distance is <= range (:302-308), nothing ever leaves topCandidates
// (its root never exceeds range, :310-311), and the stop test (:286-289) can only fire for the entry point, whose
// farthestResultDist is still MaxValue -- so every listed node and the entry point are expanded exactly once,
// whatever the pop order, and the result SET and the evaluation count do not depend on it.  The order shows only
// in RangeQuery's stable OrderBy over the heap array (HNSWIndex.cs:155) between results of EQUAL distance; the
// host sorts what comes back, and for a query that holds such a pair replays the two heaps from the entry point
// with the distances found here (no evaluation: a neighbour that is not among the results is out of range).
// `found` (per wave, found_cap entries) is both the result list and the work queue: entry `head` is the next
// node to expand.  Results are then copied to a launch-wide arena at an offset taken with one atomic.
// out_flag: 0 done; 1 hand back (more than found_cap results, or the visited table filling up); 3 arena full.
constexpr int kRangeFan = 8; // nodes expanded per step; the id / distance scratch holds kRangeFan adjacency lists
template <int METRIC, bool HASHED>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) // at most 168 VGPRs: three waves per SIMD
graph_range_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, const float *__restrict__ queries,
                   const double *__restrict__ q_sn, int dim, const int *__restrict__ adj0, int stride0,
                   const int64_t *__restrict__ upper, const int *__restrict__ pool, int strideU,
                   const SearchJob *__restrict__ jobs, float range, ND *__restrict__ found_all, int found_cap,
                   unsigned *__restrict__ visited, long long vis_words, int *__restrict__ vis_tab, int vis_tab_cap,
                   ND *__restrict__ arena, unsigned long long arena_cap, unsigned long long *__restrict__ arena_used,
                   unsigned long long *__restrict__ out_off, int *__restrict__ out_cnt, int *__restrict__ out_flag,
                   int *__restrict__ out_entry, unsigned long long *__restrict__ eval_counter, int nbcap, int njobs,
                   int *__restrict__ job_counter)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    VisitedSet<HASHED> V{visited + (size_t)blockIdx.x * (size_t)vis_words, vis_words,
                         vis_tab ? vis_tab + (size_t)blockIdx.x * (size_t)vis_tab_cap : nullptr, (unsigned)(vis_tab_cap - 1), 0, vis_tab_cap / 4 * 3};
    const SearchLds L = carve_lds(smem, 0, 0, dim, nbcap);
    const GraphView G{adj0, stride0, upper, pool, strideU};
    // the queue is read back through L2 (agent-scope loads): a line of it cached earlier may lack later entries
    unsigned long long *found = reinterpret_cast<unsigned long long *>(found_all + (size_t)blockIdx.x * (size_t)found_cap);
    int *nbuf = L.nbuf;
    float *dbuf = L.dbuf;
    for (;;) {
        int job = 0;
        if (lane == 0) job = atomicAdd(job_counter, 1);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= njobs) break;
        const SearchJob jb = jobs[job];
        const float *q = queries + (size_t)jb.qref * dim;
        double sb = 0.0;
        if (METRIC == M_COS) sb = q_sn[jb.qref];
        wave_sync();
        for (int i = lane; i < dim; i += 64) L.qs[i] = q[i];
        unsigned long long evals = 0;
        int best;
        float cur;
        ReadLog RL{nullptr, 0, 0};
        descend<METRIC>(rows, row_sn, dim, sb, G, jb, L, lane, best, cur, evals, RL); // FindEntryPointQuery; :268 reuses its distance
        if (lane == 0) (void)V.first_visit(best);                                  // :279
        V.seen += 1;
        int count = 0, head = 0;
        if (cur <= range) { // :271-275
            if (lane == 0) found[0] = ((unsigned long long)__float_as_uint(cur) << 32) | (unsigned)best;
            count = 1;
        }
        // :277 the entry point is a candidate either way; out of range it is still expanded, unless its distance
        // exceeds farthestResultDist's initial MaxValue (+inf): then :286-289 ends the search at once
        bool entry_pending = !(cur <= range) && !(cur > 3.402823466e+38f);
        bool ok = true;
        for (;;) {
            // up to kRangeFan listed nodes are expanded per step (any order gives the same set): a large result set
            // is a long dependent chain on one wave otherwise
            int W, c0 = best;
            if (entry_pending) { W = 1; entry_pending = false; }
            else {
                W = min(kRangeFan, count - head); // :283 no candidates left
                if (W == 0) break;
                unsigned long long e = 0ull;
                if (lane < W) e = __hip_atomic_load(&found[head + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // :285, :290
                c0 = (int)(unsigned)e;
                head += W;
            }
            int n[kRangeFan], nb[kRangeFan];
            const int *lw[kRangeFan];
#pragma unroll
            for (int w = 0; w < kRangeFan; ++w) {
                lw[w] = G.list(__builtin_amdgcn_readlane(c0, w < W ? w : 0), 0);
                n[w] = w < W ? __builtin_amdgcn_readfirstlane(lw[w][0]) : 0;
                nb[w] = lane < n[w] ? lw[w][1 + lane] : 0;
            }
            bool fr[kRangeFan];
#pragma unroll
            for (int w = 0; w < kRangeFan; ++w) fr[w] = lane < n[w] && V.first_visit(nb[w]); // :297 / :318 (a node two lists share is fresh once)
            int m = 0;
            wave_sync();
#pragma unroll
            for (int w = 0; w < kRangeFan; ++w) {
                const unsigned long long mask = __ballot(fr[w]);
                const int posn = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                if (fr[w]) nbuf[m + posn] = nb[w];
                m += __popcll(mask);
                for (int base = 64; base < n[w]; base += 64) { // lists beyond 64 ids (MaxEdges > 32)
                    const int i = base + lane;
                    bool fresh = false;
                    int x = 0;
                    if (i < n[w]) { x = lw[w][1 + i]; fresh = V.first_visit(x); }
                    const unsigned long long mk = __ballot(fresh);
                    const int pp = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
                    if (fresh) nbuf[m + pp] = x;
                    m += __popcll(mk);
                }
            }
            wave_sync();
            if (m == 0) continue;
            V.seen += m;
            if (V.crowded()) { ok = false; break; }
            measure_all<METRIC>(rows, row_sn, dim, L.qs, sb, nbuf, dbuf, m, lane); // :299
            wave_sync();
            evals += (unsigned long long)m;
            for (int base = 0; base < m && ok; base += 64) {
                const int i = base + lane;
                const float d = i < m ? dbuf[i] : 0.0f;
                const bool in = i < m && d <= range; // :302
                const unsigned long long mask = __ballot(in);
                const int add = __popcll(mask);
                if (count + add > found_cap) { ok = false; break; }
                const int posn = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                if (in) found[count + posn] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)nbuf[i]; // :305, :308
                count += add;
            }
            if (!ok) break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the queue entries just written are read above (through L2)
        }
        unsigned long long off = 0;
        int flag = ok ? 0 : 1;
        if (ok && count > 0) {
            if (lane == 0) off = atomicAdd(arena_used, (unsigned long long)count);
            off = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)off);
            if (off + (unsigned long long)count > arena_cap) flag = 3;
            else
                for (int i = lane; i < count; i += 64) {
                    const unsigned long long e = __hip_atomic_load(&found[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    arena[off + i] = ND{(int)(unsigned)e, __uint_as_float((unsigned)(e >> 32))};
                }
        }
        if (lane == 0) {
            out_off[job] = off;
            out_cnt[job] = flag == 0 ? count : 0;
            out_flag[job] = flag;
            out_entry[job] = best; // FindEntryPointQuery's answer: where a host replay of the heaps starts
            if (flag != 3) atomicAdd(eval_counter, evals); // (a job that found the arena full runs again)
        }
        V.clear(lane);
    }
}
#endif

// Insert, search half, fused: for one new item, GraphConnector.AddNewConnections' whole loop
// (GraphConnector.cs:172-181): FindEntryPoint, then for every layer of the item ConnectAtLayer's
// SearchLayer + RelativeNeighborPruning (:189-190) with the next layer's entry = selected[0]
// (:216).  One launch serves every layer of every item of a batch (the few multi-layer items
// clear their visited bitset between layers).  Output per (job, layer): the selected ids in
// selection order (layer 0 -> slot `job`; layer L >= 1 -> upper slot jobs[].aux + L - 1).
// jobs[].search_layer = the item's first layer min(level, top).
template <int METRIC, int NS, bool HASHED, bool LAT = false>
__device__ __forceinline__ void insert_job(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                           const int *__restrict__ adj0, int stride0, const int64_t *__restrict__ upper,
                           const int *__restrict__ pool, int strideU, const SearchJob *__restrict__ jobs, int k,
                           int cand_cap, ND *__restrict__ spill, int spill_cap, int max_edges0, VisitedSet<HASHED> &V,
                           int *__restrict__ out_sel0, int *__restrict__ out_cnt0, int *__restrict__ out_selU,
                           int *__restrict__ out_cntU, int sel_stride, int *__restrict__ out_flag,
                           unsigned long long *__restrict__ eval_counter, int nbcap, unsigned char *smem, int job, int overlap_and_flags,
                           int *__restrict__ read_log, int read_log_cap, TeamPort *port = nullptr, bool *v_dirty_out = nullptr)
{
    bool v_dirty = false; // the visited set has marks in it (the sorted traversal without a visited set -- oflags bit 3 -- leaves none)
    const bool novis = !LAT && (overlap_and_flags & 8) != 0;
    const SearchLds L = carve_lds(smem, k, cand_cap, dim, nbcap);
    const int lane = threadIdx.x & 63;
    const int overlap = overlap_and_flags & 1; // bit 0: overlapped form (bit 1: the MFMA-prefiltered heuristic is allowed)
    SearchJob jb = jobs[job];
    const GraphView G{adj0, stride0, upper, pool, strideU};
    const int item = ~jb.qref;
    const float *q = rows + (size_t)item * dim;
    double sb = 0.0;
    if (METRIC == M_COS) sb = row_sn[item];
    for (int i = lane; i < dim; i += 64) L.qs[i] = q[i];
    if constexpr (LAT) { if (lane == 0) port->m->sb = sb; }
    unsigned long long evals = 0;
    bool ok = true, repeat = false;
    // exact-window Add: record [n, entries...] of this job's read log (n beyond the capacity = overflow)
    ReadLog RL{read_log ? read_log + (size_t)job * read_log_cap + 2 : nullptr, 0, (read_log_cap - 2) / 2};
#ifdef EXP_PHASE_CLOCKS
    const long long ph_j0 = __builtin_readcyclecounter();
#endif
    const int first_layer = jb.search_layer, last_layer = jb.stop_layer;
    for (int layer = first_layer; layer >= last_layer && ok; --layer) {
        if (layer != first_layer && v_dirty) { V.clear(lane); v_dirty = false; } // a fresh SearchLayer: new visited list (VisitedListPool.cs:74-106)
        int top_n = 0;
        const int rl_n0 = RL.n;
        const int max_edges = layer == 0 ? max_edges0 : (max_edges0 >> 1); // GraphData.MaxEdges :247-250
        bool exact = NS == 0, order_tie = false;
        const unsigned long long ev0 = evals;
        if constexpr (NS > 0) {
            bool tie = false;
            if constexpr (LAT) ok = traverse_pool<METRIC, NS, HASHED>(rows, row_sn, dim, sb, G, jb, k, k, V, L, lane, top_n, tie, evals, RL, &order_tie, nullptr, port);
            else ok = traverse_sorted<METRIC, NS, HASHED>(rows, row_sn, dim, sb, G, jb, k, k, V, L, lane, top_n, tie, evals, overlap_and_flags & 9, RL, &order_tie); // Span.Sort consumes all
            v_dirty = v_dirty || !novis;
            if (!ok) break;
            // equal distances where the heap layout shows, or fewer candidates than MaxEdges (the heuristic
            // then returns them in HEAP order, Heuristic.cs:13-18): this layer again, exact traversal
            exact = tie || top_n < max_edges;
        }
        int rc = 0;
        for (;;) {
            if (exact) {
                if constexpr (NS > 0) {
                    repeat = true;
                    evals = ev0;
                    top_n = 0;
                    RL.n = rl_n0; // the same lists are read again
                    if (v_dirty) V.clear(lane);
                }
                ok = traverse<METRIC, HASHED>(rows, row_sn, dim, sb, G, jb, k, cand_cap, spill, spill_cap, V, L, lane, top_n, evals, RL, nullptr, nullptr,
                                              LAT || overlap != 0);
                v_dirty = true;
                if (!ok) break;
            }
#ifdef EXP_PHASE_CLOCKS
            const long long ph_h0 = __builtin_readcyclecounter();
#endif
            // the candidate heap's LDS area is idle now: the grouped heuristic borrows it
            rc = relative_neighbor_pruning<METRIC, NS == 8>(rows, row_sn, dim, L.top, top_n, max_edges, L, lane, evals, !exact,
#ifdef HNSW_NO_GROUPED
                                                            nullptr, 0);
#else
                                                            reinterpret_cast<float *>(L.cand), sizeof(ND) * (size_t)cand_cap, (overlap_and_flags & 2) != 0);
#endif
#ifdef EXP_PHASE_CLOCKS
            if (lane == 0) atomicAdd(&g_phase[8], (unsigned long long)(__builtin_readcyclecounter() - ph_h0)); // heuristic cycles
#endif
            if (exact || !order_tie) break;
            // Equal distances somewhere in the ascending candidate list, and nothing else open: the SET is the
            // reference's, but Span.Sort (Heuristic.cs:22) leaves such a group in an order only the heap array knows.
            // The greedy pass (:23-40) shows that order only if two members of a group get past the ids accepted before
            // the group (one may then turn the other away, or both enter the list in that order).  A member that was NOT
            // accepted just now, with no member of its group accepted before it, was turned away by ids of smaller
            // distance -- in any order.  So when every member but the last of each group was rejected, the outcome is
            // the reference's whatever its order was (one candidate in seven is accepted on uniform data: most groups
            // are harmless -- 2.1 % of the inserts at C2 used to start over, a third of a percent still do).
            wave_sync();
            bool shows = false;
            for (int p0 = 0; p0 < top_n; p0 += 64) {
                const int pp = p0 + lane;
                if (pp >= 1 && pp < top_n && __float_as_uint(L.top[pp].dist) == __float_as_uint(L.top[pp - 1].dist)) {
                    const int first = L.top[pp - 1].id;
                    for (int a = 0; a < rc; ++a) shows = shows || L.acc[a] == first;
                }
            }
            if (__ballot(shows) == 0ull) break;
            exact = true;
        }
        if (!ok) break;
        int *osel = layer == 0 ? out_sel0 + (size_t)job * sel_stride : out_selU + (size_t)(jb.aux + layer - 1) * sel_stride;
        for (int i = lane; i < rc; i += 64) osel[i] = L.acc[i];
        if (lane == 0) { if (layer == 0) out_cnt0[job] = rc; else out_cntU[jb.aux + layer - 1] = rc; }
        const int next_entry = __builtin_amdgcn_readfirstlane(L.acc[0]); // :216 selected[0] -> bestPeer of the next layer (:179)
        jb.entry = next_entry;
        jb.entry_layer = layer - 1;
        jb.search_layer = layer - 1;
        wave_sync();
    }
    if (v_dirty_out) *v_dirty_out = v_dirty || LAT; // (the latency variants' memory wave marks as it goes)
    if (lane == 0) {
        out_flag[job] = ok ? (repeat ? 2 : 0) : 1; // 2: informational (a layer was answered by the exact traversal)
        if (read_log) read_log[(size_t)job * read_log_cap] = RL.n;
        atomicAdd(eval_counter, evals);
#ifdef EXP_PHASE_CLOCKS
        atomicAdd(&g_phase[9], (unsigned long long)(__builtin_readcyclecounter() - ph_j0)); // whole insert job
#endif
    }
}

template <int METRIC, int NS, bool HASHED, bool LAT = false>
__global__ void __launch_bounds__(LAT ? 128 : 64) __attribute__((amdgpu_waves_per_eu(HNSW_WAVES(LAT ? (NS <= 4 ? 2 : 1) : NS <= 4 ? 3 : 2)))) // up to 256 candidates: 168 VGPRs, three waves per SIMD (LAT: see graph_search_kernel)
graph_insert_search_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                           const int *__restrict__ adj0, int stride0, const int64_t *__restrict__ upper,
                           const int *__restrict__ pool, int strideU, const SearchJob *__restrict__ jobs, int k,
                           int cand_cap, ND *__restrict__ spill, int spill_cap, int max_edges0, unsigned *__restrict__ visited, long long vis_words,
                           int *__restrict__ vis_tab, int vis_tab_cap, int *__restrict__ out_sel0, int *__restrict__ out_cnt0, int *__restrict__ out_selU,
                           int *__restrict__ out_cntU, int sel_stride, int *__restrict__ out_flag,
                           unsigned long long *__restrict__ eval_counter, int nbcap, int njobs, int *__restrict__ job_counter, int overlap,
                           const int *__restrict__ order, int *__restrict__ read_log, int read_log_cap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    VisitedSet<HASHED> V{visited + (size_t)blockIdx.x * (size_t)vis_words, vis_words,
                 vis_tab ? vis_tab + (size_t)blockIdx.x * (size_t)vis_tab_cap : nullptr, (unsigned)(vis_tab_cap - 1), 0, vis_tab_cap / 4 * 3};
    ND *my_spill = spill + (size_t)blockIdx.x * spill_cap;

    TeamPort port{nullptr, 0, 0};
    if constexpr (LAT) {
        // two waves per block (see TeamMail): wave 1 serves the expansions, wave 0 is the traversal.  The mailbox follows
        // the traversal's LDS; its sequence words are zeroed before the roles part (the one barrier both waves meet at).
        TeamMail *mail = reinterpret_cast<TeamMail *>(smem + ((search_lds_bytes(k, cand_cap, dim, true, nbcap) + 15) & ~(size_t)15));
        if (threadIdx.x == 0) { mail->req_seq = 0; mail->rsp_seq = 0; mail->hint_node = -1; }
        __syncthreads();
        if (threadIdx.x >= 64) {
            const GraphView G{adj0, stride0, upper, pool, strideU};
            const SearchLds L = carve_lds(smem, k, cand_cap, dim, nbcap);
            memory_wave<METRIC, HASHED>(rows, row_sn, dim, G, V, L.qs, mail, lane);
            return;
        }
        port.m = mail;
    }
    bool v_dirty = true;
    for (;;) { // persistent, see graph_search_kernel
        int job = 0;
        if (lane == 0) job = atomicAdd(job_counter, 1);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= njobs) break;
        // queue position -> batch item: the items that search several layers are taken first (they run the
        // longest; started last they would be the tail of the launch).  Results are filed by item, so the
        // order of processing changes nothing else.
        if (order) job = __builtin_amdgcn_readfirstlane(order[job]);
        insert_job<METRIC, NS, HASHED, LAT>(rows, row_sn, dim, adj0, stride0, upper, pool, strideU, jobs, k, cand_cap, my_spill, spill_cap, max_edges0, V,
                               out_sel0, out_cnt0, out_selU, out_cntU, sel_stride, out_flag, eval_counter, nbcap, smem, job, overlap, read_log, read_log_cap, &port, &v_dirty);
        if (v_dirty) V.clear(lane);
    }

    if constexpr (LAT) port.post(-1, 0, lane); // the memory wave leaves
}

// Insert, link half, on the HBM mirror.  (a) new nodes' own lists.
#ifdef HNSW_HOST_TU // launched from one place: defined only in the unit that launches it
// Remove, second half (GraphConnector.RemoveConnectionsAtLayer :100-133): one wave per AFFECTED node (an in-edge
// neighbour of the removed node): drop the edge to the removed node (EdgeList.Remove: the last entry takes its
// place), candidates = the remaining neighbours followed by the search candidates that are neither the node itself
// nor among them (:115-129), Distance(candidate, node) for all of them, RelativeNeighborPruning (:131).  Nothing
// is written to the graph: the selection goes back to the host, which applies the difference (:135-164).
// `cands` arrive ascending by distance to the removed node, not in the reference's heap-array order; that order
// shows only if the heuristic returns its input unsorted (fewer candidates than MaxEdges) or sorts equal distances:
// both raise out_flag and the host repeats the step on the exact lock-step path.
template <int METRIC>
__global__ void __launch_bounds__(64)
graph_relink_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, const int *__restrict__ adj0, int stride0,
                    const int64_t *__restrict__ upper, const int *__restrict__ pool, int strideU, const int4 *__restrict__ jobs,
                    const int *__restrict__ cands_all, const int *__restrict__ cand_off, const int *__restrict__ cand_cnt, int max_edges0,
                    int kcap, int nbcap, int *__restrict__ out_sel, int *__restrict__ out_cnt, int *__restrict__ out_flag, int sel_stride,
                    unsigned long long *__restrict__ eval_counter, int heap_order)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x, job = blockIdx.x;
    const SearchLds L = carve_lds(smem, kcap, 0, dim, nbcap);
    const GraphView G{adj0, stride0, upper, pool, strideU};
    // jobs[]: (affected node, layer, removed node, step); the step's search candidates: cands_all[cand_off[step] ..][0 .. cand_cnt[step])
    const int4 jd = jobs[job];
    const int aid = jd.x, layer = jd.y, removed = jd.z;
    const int *cands = cands_all + cand_off[jd.w];
    const int ncand = cand_cnt[jd.w];
    const int max_edges = layer == 0 ? max_edges0 : (max_edges0 >> 1); // GraphData.MaxEdges :247-250
    const float *q = rows + (size_t)aid * dim;
    double sb = 0.0;
    if (METRIC == M_COS) sb = row_sn[aid];
    for (int i = lane; i < dim; i += 64) L.qs[i] = q[i];
    const int *l = G.list(aid, layer);
    int cnt = l[0];
    // RemoveOutEdge :104 (EdgeList.Remove, Node.cs:79-93: swap with the last)
    int pos = -1;
    for (int base = 0; base < cnt && pos < 0; base += 64) {
        const unsigned long long hit = __ballot(base + lane < cnt && l[1 + base + lane] == removed);
        if (hit) pos = base + (int)__builtin_ctzll(hit);
    }
    const int last = cnt - 1;
    if (pos >= 0) --cnt;
    for (int i = lane; i < cnt; i += 64) L.nbuf[i] = (i == pos) ? l[1 + last] : l[1 + i]; // :110-120 the existing neighbours
    wave_sync();
    int n = cnt;
    bool bad = false;
    for (int base = 0; base < ncand; base += 64) { // :123-129
        const int i = base + lane;
        const int c = i < ncand ? cands[i] : -1;
        bool keep = i < ncand && c != aid;
        for (int t = 0; keep && t < cnt; ++t) keep = L.nbuf[t] != c;
        const unsigned long long mask = __ballot(keep);
        const int posn = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        if (n + (int)__popcll(mask) > min(kcap, nbcap)) { bad = true; break; }
        if (keep) L.nbuf[n + posn] = c;
        n += (int)__popcll(mask);
    }
    wave_sync();
    unsigned long long evals = 0;
    int rc = 0;
    // heap_order: `cands` are SearchLayer's heap array itself (the exact two-heap search), so the candidate array is
    // the reference's, element for element, and nothing below depends on anything else
    if (!bad && !heap_order && n < max_edges) bad = true; // Heuristic.cs:13-18 returns the INPUT order: the heap array's
    if (!bad && n == 0) rc = 0;
    if (!bad && n > 0) {
        measure_all<METRIC>(rows, row_sn, dim, L.qs, sb, L.nbuf, L.dbuf, n, lane); // Distance(id, affectedNodeId) :118, :128
        wave_sync();
        evals += (unsigned long long)n;
        for (int i = lane; i < n; i += 64) L.top[i] = ND{L.nbuf[i], L.dbuf[i]};
        wave_sync();
        rc = relative_neighbor_pruning<METRIC>(rows, row_sn, dim, L.top, n, max_edges, L, lane, evals); // sorts L.top
        wave_sync();
        bool odd = false; // equal, NaN or -0 distances: Span.Sort's answer depends on the input order
        for (int i = lane; i < n; i += 64) {
            const float d = L.top[i].dist;
            odd |= key_unsafe(d) || (i + 1 < n && f2key(L.top[i + 1].dist) == f2key(d));
        }
        if (!heap_order && __ballot(odd) != 0ull) bad = true;
    }
    if (!bad) for (int i = lane; i < rc; i += 64) out_sel[(size_t)job * sel_stride + i] = L.acc[i];
    if (lane == 0) {
        out_cnt[job] = bad ? 0 : rc;
        out_flag[job] = bad ? 1 : 0;
        atomicAdd(eval_counter, evals);
    }
}
#endif

#ifdef HNSW_HOST_TU // non-template kernels: only the unit that launches them defines them
__global__ void __launch_bounds__(64)
graph_write_rows_kernel(int *__restrict__ adj0, int stride0, const int64_t *__restrict__ upper, int *__restrict__ pool,
                        int strideU, const int *__restrict__ recs, int row_stride, int *__restrict__ tested0,
                        int *__restrict__ testedU, int max_edges0)
{
    const int *r = recs + (size_t)blockIdx.x * row_stride;
    const int node = r[0], layer = r[1] & 0xffff, cnt = r[2];
    const bool untested = (r[1] >> 30) & 1; // the list is not a heuristic's ordered output (a removal's re-link)
    int *l = layer == 0 ? adj0 + (size_t)node * stride0 : pool + upper[node] + (size_t)(layer - 1) * strideU;
    if (threadIdx.x == 0) {
        l[0] = cnt;
        // a full list can only be the ordered output of the heuristic's greedy pass (fewer candidates
        // than MaxEdges come back unsorted, Heuristic.cs:13-18): its entries are mutually tested
        const int me = layer == 0 ? max_edges0 : (max_edges0 >> 1);
        int *t = layer == 0 ? tested0 + node : testedU + (upper[node] / strideU + (layer - 1));
        *t = (cnt == me && !untested) ? cnt : 0;
    }
    for (int i = threadIdx.x; i < cnt; i += 64) l[1 + i] = r[3 + i];
}
#endif

// (b) one wave per (neighbour, layer) list: every back-edge append of the batch, in item order
// (neighbor.OutEdges[layer].Add(currNode.Id), GraphConnector.cs:207), each overflow pruned in
// place (PruneOverflow :222-262: distances :230-234, sort + heuristic :235).  Lists are
// independent, so the outcome equals the reference's sequential loop.
// next_item(): the next node id to append to this list, in item order, or -1.  out_list (optional):
// [count, ids...] of the final list for the host.
template <int METRIC, class NextItem>
__device__ __forceinline__ void link_group(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, int *__restrict__ adj0,
                  int stride0, const int64_t *__restrict__ upper, int *__restrict__ pool, int strideU,
                  int node, int layer, NextItem next_item, int max_edges0, int k_cap, int *__restrict__ out_list,
                  unsigned long long *__restrict__ eval_counter, int nbcap, int *__restrict__ tested0, int *__restrict__ testedU,
                  unsigned char *smem, int *__restrict__ dry_changed = nullptr, int *__restrict__ dry_drop = nullptr, int dry_item = -1)
{
    const SearchLds L = carve_lds(smem, k_cap, 0, dim, nbcap);
    // shortcut scratch behind the common carve-up: distances of up to kNewMax new entries to every
    // entry of the list, and the sorted order as original positions
    float *Dm = reinterpret_cast<float *>(smem + ((search_lds_bytes(k_cap, 0, dim, true, nbcap) + 15) & ~(size_t)15));
    int *perm = reinterpret_cast<int *>(Dm + kNewMax * nbcap);
    const int lane = threadIdx.x;
    const int max_edges = layer == 0 ? max_edges0 : (max_edges0 >> 1);
    int *l = layer == 0 ? adj0 + (size_t)node * stride0 : pool + upper[node] + (size_t)(layer - 1) * strideU;
    const float *q = rows + (size_t)node * dim;
    double sb = 0.0;
    if (METRIC == M_COS) sb = row_sn[node];
    for (int i = lane; i < dim; i += 64) L.qs[i] = q[i];
    int cnt = l[0];
    for (int i = lane; i < cnt; i += 64) L.nbuf[i] = l[1 + i];
    int *tested_p = layer == 0 ? tested0 + node : testedU + (upper[node] / strideU + (layer - 1));
    int tested = min(max(*tested_p, 0), cnt); // leading entries that are an ordered, mutually tested heuristic output
    wave_sync();
    unsigned long long evals = 0;
    PH_DECL();
    PH(0);
    for (int item = next_item(); item >= 0; item = next_item()) {
        if (lane == 0) L.nbuf[cnt] = item; // :207
        cnt++;
        wave_sync();
        PH_COUNT(6, 1);
        if (cnt > max_edges) { // :209
            measure_all<METRIC>(rows, row_sn, dim, L.qs, sb, L.nbuf, L.dbuf, cnt, lane); // Distance(cand, node.Id) :233
            wave_sync();
            evals += (unsigned long long)cnt;
            int rc = -1;
            // Shortcut.  The first `tested` entries are the output of an earlier greedy pass over this
            // very list (same node, same distances): ascending, and every earlier one already passed
            // the test `dist(s, c) < c.Dist` against every later one (Heuristic.cs:31-35).  Those pairs
            // need not be measured again; only pairs with one of the entries appended since do.  With
            // few new entries (typically one: lists are full, every append overflows) that is one
            // batch of distances per new entry instead of one dependent batch per candidate.
            const int n = cnt, u = n - tested;
            if (tested > 0 && u <= kNewMax && n <= 128) { // entries i = lane and i = lane + 64 on each lane
                const int i1 = lane + 64;
                const float d0 = lane < n ? L.dbuf[lane] : 0.0f, d1 = i1 < n ? L.dbuf[i1] : 0.0f;
                const unsigned k0 = f2key(d0), k1 = f2key(d1);
                bool odd = (lane < n && key_unsafe(d0)) || (i1 < n && key_unsafe(d1));
                int rank0 = 0, rank1 = 0;
                for (int t2 = 0; t2 < n; ++t2) { // Span.Sort :22 -- distinct ordinary distances: rank by counting
                    const unsigned kt = t2 < 64 ? (unsigned)__builtin_amdgcn_readlane((int)k0, t2) : (unsigned)__builtin_amdgcn_readlane((int)k1, t2 - 64);
                    rank0 += kt < k0 ? 1 : 0;
                    rank1 += kt < k1 ? 1 : 0;
                    odd |= lane < n && t2 != lane && kt == k0;
                    odd |= i1 < n && t2 != i1 && kt == k1;
                    // the tested prefix must still be ascending (it is, by construction)
                    odd |= lane < tested && t2 < tested && ((t2 < lane && kt >= k0) || (t2 > lane && kt <= k0));
                    odd |= i1 < tested && t2 < tested && ((t2 < i1 && kt >= k1) || (t2 > i1 && kt <= k1));
                }
                if (__ballot(odd) == 0ull) {
                    if (lane < n) perm[rank0] = lane;
                    if (i1 < n) perm[rank1] = i1;
                    // distances of every new entry to all entries of the list
                    for (int jn = 0; jn < u; ++jn) {
                        const int xid = L.nbuf[tested + jn];
                        const float *xrow = rows + (size_t)xid * dim;
                        wave_sync();
                        for (int t2 = lane; t2 < dim; t2 += 64) L.qs2[t2] = xrow[t2];
                        double sbx = 0.0;
                        if (METRIC == M_COS) sbx = row_sn[xid];
                        wave_sync();
                        // a single new entry only meets the old ones (one pass of <= 32 rows instead of two)
                        const int mrows = u == 1 ? tested : n;
                        measure_all<METRIC>(rows, row_sn, dim, L.qs2, sbx, L.nbuf, Dm + jn * nbcap, mrows, lane);
                        evals += (unsigned long long)(u == 1 ? mrows : n - 1);
                    }
                    wave_sync();
                    // greedy pass :23-40 in sorted order, on the distances at hand
                    bool acc0 = false, acc1 = false; // entries lane / lane + 64 accepted
                    unsigned new_acc = 0u;           // bit j: new entry j accepted
                    rc = 0;
                    for (int p2 = 0; p2 < n && rc < max_edges; ++p2) {
                        const int i = perm[p2];
                        const float di = L.dbuf[i];
                        bool rej;
                        if (i < tested) {           // an old entry: only accepted new ones can object
                            rej = false;
                            for (int jn = 0; jn < u; ++jn)
                                if ((new_acc >> jn) & 1u) rej = rej || Dm[jn * nbcap + i] < di;
                        } else {                    // a new entry: everything accepted so far can object
                            const float *Dj = Dm + (i - tested) * nbcap;
                            const float e0 = lane < n ? Dj[lane] : 0.0f, e1 = i1 < n ? Dj[i1] : 0.0f;
                            rej = __ballot((acc0 && e0 < di) || (acc1 && e1 < di)) != 0ull;
                        }
                        if (!rej) {
                            if (lane == i) acc0 = true;
                            if (i1 == i) acc1 = true;
                            if (i >= tested) new_acc |= 1u << (i - tested);
                            if (lane == 0) L.acc[rc] = L.nbuf[i];
                            rc++;
                        }
                    }
                    wave_sync();
                }
            }
            if (rc < 0) {
                for (int i = lane; i < cnt; i += 64) L.top[i] = ND{L.nbuf[i], L.dbuf[i]};
                rc = relative_neighbor_pruning<METRIC>(rows, row_sn, dim, L.top, cnt, max_edges, L, lane, evals);
            }
            for (int i = lane; i < rc; i += 64) L.nbuf[i] = L.acc[i]; // node.OutEdges[layer] = newOut :236
            cnt = rc;
            tested = rc; // the whole list is a greedy output now
            wave_sync();
        }
    }
    if (dry_changed) { // dry run (exact-window Add): nothing is written; would the list read differently afterwards?
        // code 0: the same sequence of ids.  Otherwise bit 0 set, bit 1 = the appended item stays in the list, bits 8.. = how
        // many ids the list loses (their ids to dry_drop[0..3), at most three; 255 = more than that).
        const int oc = l[0];
        bool diff = cnt != oc;
        for (int i = lane; i < cnt && !diff; i += 64) diff = L.nbuf[i] != l[1 + i];
        int code = 0;
        if (__ballot(diff) != 0ull) {
            bool has = false;
            for (int i = lane; i < cnt; i += 64) has = has || L.nbuf[i] == dry_item;
            code = 1 | (__ballot(has) != 0ull ? 2 : 0);
            int nd = 0;
            for (int base = 0; base < oc; base += 64) {
                const int i = base + lane;
                bool gone = false;
                int x = 0;
                if (i < oc) {
                    x = l[1 + i];
                    gone = true;
                    for (int u = 0; u < cnt; ++u) gone = gone && L.nbuf[u] != x;
                }
                unsigned long long m = __ballot(gone);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const int gid = __builtin_amdgcn_readlane(x, src);
                    if (nd < 3 && dry_drop && lane == 0) dry_drop[nd] = gid;
                    nd++;
                }
            }
            code |= (nd > 3 ? 255 : nd) << 8;
        }
        if (lane == 0) { *dry_changed = code; atomicAdd(eval_counter, evals); }
        wave_sync();
        return;
    }
    if (lane == 0) { l[0] = cnt; *tested_p = tested; if (out_list) out_list[0] = cnt; }
    for (int i = lane; i < cnt; i += 64) { l[1 + i] = L.nbuf[i]; if (out_list) out_list[1 + i] = L.nbuf[i]; }
    if (lane == 0) atomicAdd(eval_counter, evals);
    wave_sync();
}

// groups prepared by the host: one block per group, items in CSR order
template <int METRIC>
__global__ void __launch_bounds__(64)
graph_link_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, int *__restrict__ adj0,
                  int stride0, const int64_t *__restrict__ upper, int *__restrict__ pool, int strideU,
                  const int *__restrict__ g_node, const int *__restrict__ g_layer, const int *__restrict__ g_off,
                  const int *__restrict__ g_count, const int *__restrict__ g_items, int max_edges0, int k_cap,
                  int *__restrict__ out_lists, int list_stride,
                  unsigned long long *__restrict__ eval_counter, int nbcap, int *__restrict__ tested0, int *__restrict__ testedU)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int g = blockIdx.x;
    int t = g_off[g];
    const int t_end = g_count ? t + g_count[g] : g_off[g + 1]; // CSR offsets, or start + count per group
    link_group<METRIC>(rows, row_sn, dim, adj0, stride0, upper, pool, strideU, g_node[g], g_layer[g],
                       [&]() { return t < t_end ? g_items[t++] : -1; }, max_edges0, k_cap,
                       out_lists ? out_lists + (size_t)g * list_stride : (int *)nullptr, eval_counter, nbcap, tested0, testedU, smem);
}

// Dry run of single appends (exact-window Add): job g = (node, layer, item) -- would appending `item` to that list,
// with PruneOverflow if it overflows (GraphConnector.cs:207-212), leave a list that READS differently (another
// sequence of ids)?  A full list whose prune turns the new item away comes out as the very same sequence (the
// earlier entries are a greedy output: ascending, mutually tested), and three out of four appends into a grown
// graph end that way: for every search that read the list, such an append never happened.  Writes nothing.
template <int METRIC>
__global__ void __launch_bounds__(64)
graph_link_dry_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, int *__restrict__ adj0,
                      int stride0, const int64_t *__restrict__ upper, int *__restrict__ pool, int strideU,
                      const int *__restrict__ jobs3, int max_edges0, int k_cap, int *__restrict__ out_changed,
                      unsigned long long *__restrict__ eval_counter, int nbcap, int *__restrict__ tested0, int *__restrict__ testedU)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int g = blockIdx.x;
    int item = jobs3[3 * g + 2];
    const int the_item = item;
    link_group<METRIC>(rows, row_sn, dim, adj0, stride0, upper, pool, strideU, jobs3[3 * g], jobs3[3 * g + 1],
                       [&]() { const int r = item; item = -1; return r; }, max_edges0, k_cap, (int *)nullptr, eval_counter, nbcap,
                       tested0, testedU, smem, out_changed + g, (int *)nullptr, the_item);
}

// The same for the selections an insert search just left on the device (no host step in between): block b stands
// for entry b % sel_stride of selection row b / sel_stride -- rows [0, njobs) are the jobs' layer-0 selections,
// row njobs + u is upper slot u, whose job is upper_owner[u].  out0 / outU (same shape as the selections) are
// preset to 1 by the host; rows a job did not produce (stop_layer), handed-back jobs and entries beyond the
// count keep that.
template <int METRIC>
__global__ void __launch_bounds__(64)
graph_link_dry_sel_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, int *__restrict__ adj0,
                          int stride0, const int64_t *__restrict__ upper, int *__restrict__ pool, int strideU,
                          const SearchJob *__restrict__ jobs, const int *__restrict__ flag, const int *__restrict__ sel0,
                          const int *__restrict__ cnt0, const int *__restrict__ selU, const int *__restrict__ cntU, int sel_stride,
                          const int *__restrict__ upper_owner, int njobs, int max_edges0, int k_cap, int *__restrict__ out0,
                          int *__restrict__ outU, unsigned long long *__restrict__ eval_counter, int nbcap, int *__restrict__ tested0,
                          int *__restrict__ testedU, long long n_nodes, int *__restrict__ drop0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int r = blockIdx.x / sel_stride, e = blockIdx.x % sel_stride;
    int job, layer, cnt;
    const int *sel;
    int *out, *drop = nullptr; // the ids a list would lose are reported for layer 0 (three per entry)
    if (r < njobs) {
        job = r; layer = 0;
        if (jobs[job].stop_layer > 0) return;
        cnt = cnt0[r]; sel = sel0 + (size_t)r * sel_stride; out = out0 + (size_t)r * sel_stride;
        drop = drop0 + ((size_t)r * sel_stride + e) * 3;
    } else {
        const int u = r - njobs;
        job = upper_owner[u];
        if (job < 0 || job >= njobs) return;
        layer = u - jobs[job].aux + 1;
        if (layer < 1 || layer > jobs[job].search_layer || layer < jobs[job].stop_layer) return;
        cnt = cntU[u]; sel = selU + (size_t)u * sel_stride; out = outU + (size_t)u * sel_stride;
    }
    if (flag[job] == 1 || e >= cnt || cnt > (layer == 0 ? max_edges0 : (max_edges0 >> 1))) return;
    const int nb = sel[e];
    int item = ~jobs[job].qref;
    if (nb < 0 || nb >= n_nodes || item < 0 || item >= n_nodes) return;
    const int the_item = item;
    link_group<METRIC>(rows, row_sn, dim, adj0, stride0, upper, pool, strideU, nb, layer,
                       [&]() { const int x = item; item = -1; return x; }, max_edges0, k_cap, (int *)nullptr, eval_counter, nbcap,
                       tested0, testedU, smem, out + e, drop, the_item);
}

// ---- the same with the grouping done on the device (no host work between the insert search and
// the link half).  Per adjacency-list slot (layer 0: the node id; upper layers: cap_n + list index
// in the pool) three counters, all zero between batches: appends, fill cursor, start offset. ----
struct LinkPlan {
    int *cnt, *fill, *off;                      // per list slot
    int *g_node, *g_layer, *g_start, *g_count;  // per group (a list that receives appends), any order
    int *items;                                 // batch positions of the appending items, grouped
    int *counters;                              // [0] groups, [1] item cursor, [3] first guard that fired
    long long cap_n;
    long long n_slots, n_nodes; // capacities, for the guards below: an index outside them is reported, never used
    int g_cap, n_jobs;
};
#define LINK_GUARD(cond, code) if (!(cond)) { atomicCAS(&P.counters[3], 0, (code)); continue; }
__device__ __forceinline__ long long link_slot(const LinkPlan &P, const int64_t *upper, int strideU, int nb, int layer)
{
    return layer == 0 ? (long long)nb : P.cap_n + upper[nb] / strideU + (layer - 1);
}
// pass 1 (count = true): the new nodes' own lists go into the mirror (currNode.OutEdges[layer] =
// selected, GraphConnector.cs:192), appends are counted per target list and the lists that receive
// any are enumerated.  pass 2 (count = false): the appends are filed per list.
template <bool COUNT>
__global__ void __launch_bounds__(64)
link_plan_kernel(const SearchJob *__restrict__ jobs, const int *__restrict__ sel0, const int *__restrict__ cnt0,
                 const int *__restrict__ selU, const int *__restrict__ cntU, int sel_stride, int *__restrict__ adj0, int stride0,
                 const int64_t *__restrict__ upper, int *__restrict__ pool, int strideU, int *__restrict__ tested0,
                 int *__restrict__ testedU, int max_edges0, LinkPlan P)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    const SearchJob jb = jobs[t];
    const int id = ~jb.qref;
    for (int layer = jb.search_layer; layer >= 0; --layer) {
        const int *sel = layer == 0 ? sel0 + (size_t)t * sel_stride : selU + (size_t)(jb.aux + layer - 1) * sel_stride;
        const int sc = layer == 0 ? cnt0[t] : cntU[jb.aux + layer - 1];
        LINK_GUARD(id >= 0 && id < P.n_nodes && sc >= 0 && sc <= sel_stride && sc <= (layer == 0 ? max_edges0 : (max_edges0 >> 1)), 1);
        if (COUNT) {
            int *l = layer == 0 ? adj0 + (size_t)id * stride0 : pool + upper[id] + (size_t)(layer - 1) * strideU;
            if (lane == 0) {
                l[0] = sc;
                const int me = layer == 0 ? max_edges0 : (max_edges0 >> 1);
                int *tp = layer == 0 ? tested0 + id : testedU + (upper[id] / strideU + (layer - 1));
                *tp = sc == me ? sc : 0; // see graph_write_rows_kernel
            }
            for (int e = lane; e < sc; e += 64) l[1 + e] = sel[e];
        }
        for (int e = lane; e < sc; e += 64) {
            const int nb = sel[e];
            LINK_GUARD(nb >= 0 && nb < P.n_nodes, 2);
            const long long slot = link_slot(P, upper, strideU, nb, layer);
            LINK_GUARD(slot >= 0 && slot < P.n_slots, 3);
            if (COUNT) {
                if (atomicAdd(&P.cnt[slot], 1) == 0) {
                    const int g = atomicAdd(&P.counters[0], 1);
                    LINK_GUARD(g < P.g_cap, 4);
                    P.g_node[g] = nb;
                    P.g_layer[g] = layer;
                }
            } else {
                const int p = atomicAdd(&P.fill[slot], 1);
                const long long at = (long long)P.off[slot] + p;
                LINK_GUARD(at >= 0 && at < P.g_cap, 5);
                P.items[at] = t;
            }
        }
    }
}
#ifdef HNSW_HOST_TU // non-template kernels: only the unit that launches them defines them
__global__ void __launch_bounds__(256)
link_offsets_kernel(const int64_t *__restrict__ upper, int strideU, LinkPlan P)
{
    const int G = min(P.counters[0], P.g_cap);
    for (int g = blockIdx.x * 256 + threadIdx.x; g < G; g += gridDim.x * 256) {
        const long long slot = link_slot(P, upper, strideU, P.g_node[g], P.g_layer[g]);
        LINK_GUARD(slot >= 0 && slot < P.n_slots, 6);
        const int c = P.cnt[slot];
        const int start = atomicAdd(&P.counters[1], c);
        LINK_GUARD(c >= 0 && start >= 0 && (long long)start + c <= P.g_cap, 7);
        P.g_start[g] = start;
        P.g_count[g] = c;
        P.off[slot] = start;
    }
}
#endif
// one block per group: its items (batch positions, filed in arbitrary order) become node ids in batch
// order -- repeatedly the smallest position not yet taken; groups are tiny -- and the slot's
// counters return to zero for the next batch
#ifdef HNSW_HOST_TU // non-template kernels: only the unit that launches them defines them
__global__ void __launch_bounds__(64)
link_order_kernel(const SearchJob *__restrict__ jobs, const int64_t *__restrict__ upper, int strideU, int *__restrict__ items_out, LinkPlan P)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    const int node = P.g_node[g], layer = P.g_layer[g], start = P.g_start[g], n_items = P.g_count[g];
    if (!(node >= 0 && node < P.n_nodes && layer >= 0 && start >= 0 && n_items >= 0 && (long long)start + n_items <= P.g_cap)) {
        atomicCAS(&P.counters[3], 0, 8);
        return;
    }
    int last = -1;
    for (int k = 0; k < n_items; ++k) {
        int best = 0x7fffffff;
        for (int i = lane; i < n_items; i += 64) {
            const int p = P.items[start + i];
            if (p > last && p < best) best = p;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
        if (best >= P.n_jobs) { atomicCAS(&P.counters[3], 0, 9); return; }
        last = best;
        if (lane == 0) items_out[start + k] = ~jobs[best].qref;
    }
    if (lane == 0) {
        const long long slot = link_slot(P, upper, strideU, node, layer);
        P.cnt[slot] = 0;
        P.fill[slot] = 0;
    }
}
#endif

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


// Explicit instantiations of the two traversal kernels live in traverse_<metric>_<search|insert>.hip;
// every other unit only declares them.
#define HNSW_FOR_EACH_TRAVERSAL(X, M) \
    X(M, 0, false, false) X(M, 1, false, false) X(M, 2, false, false) X(M, 4, false, false) X(M, 8, false, false) \
    X(M, 0, true, false) X(M, 1, true, false) X(M, 2, true, false) X(M, 4, true, false) X(M, 8, true, false)
// the latency variants (sorted-list kernels only): units of their own
#define HNSW_FOR_EACH_TRAVERSAL_LAT(X, M) \
    X(M, 1, false, true) X(M, 2, false, true) X(M, 4, false, true) X(M, 8, false, true) \
    X(M, 1, true, true) X(M, 2, true, true) X(M, 4, true, true) X(M, 8, true, true)
#define HNSW_SEARCH_SIGNATURE(PREFIX, M, NS, H, LT)                                                                                  \
    PREFIX template __global__ void graph_search_kernel<M, NS, H, LT>(                                                              \
        const float *__restrict__, const double *__restrict__, const float *__restrict__, const double *__restrict__, int,      \
        const int *__restrict__, int, const int64_t *__restrict__, const int *__restrict__, int, const SearchJob *__restrict__,  \
        int, int, ND *__restrict__, int, unsigned *__restrict__, long long, int *__restrict__, int, int, int *__restrict__,      \
        float *__restrict__, int *__restrict__, int *__restrict__, unsigned long long *__restrict__, int, int, int *__restrict__, int, \
        const int *__restrict__);
#define HNSW_INSERT_SIGNATURE(PREFIX, M, NS, H, LT)                                                                                  \
    PREFIX template __global__ void graph_insert_search_kernel<M, NS, H, LT>(                                                       \
        const float *__restrict__, const double *__restrict__, int, const int *__restrict__, int, const int64_t *__restrict__,   \
        const int *__restrict__, int, const SearchJob *__restrict__, int, int, ND *__restrict__, int, int, unsigned *__restrict__, \
        long long, int *__restrict__, int, int *__restrict__, int *__restrict__, int *__restrict__, int *__restrict__, int,      \
        int *__restrict__, unsigned long long *__restrict__, int, int, int *__restrict__, int, const int *__restrict__, int *__restrict__, int);
#define HNSW_DECLARE_TRAVERSAL(M, NS, H, LT) HNSW_SEARCH_SIGNATURE(extern, M, NS, H, LT) HNSW_INSERT_SIGNATURE(extern, M, NS, H, LT)
#define HNSW_DEFINE_TRAVERSAL(M, NS, H, LT) HNSW_SEARCH_SIGNATURE(, M, NS, H, LT) HNSW_INSERT_SIGNATURE(, M, NS, H, LT)
#define HNSW_DEFINE_SEARCH(M, NS, H, LT) HNSW_SEARCH_SIGNATURE(, M, NS, H, LT)
#define HNSW_DEFINE_INSERT(M, NS, H, LT) HNSW_INSERT_SIGNATURE(, M, NS, H, LT)

} // namespace hnsw
