// device_backend.hip -- gfx950 (MI355X / CDNA4) candidate-distance kernels and the device
// context that owns the HBM-resident vector matrix.
//
// What the kernels replace (all citations relative to /root/reference/):
//   SquaredEuclideanMetric.Compute  src/HNSWIndex/Metrics/EuclideanMetric.cs:11-60
//   CosineMetric.UnitCompute        src/HNSWIndex/Metrics/CosineMetric.cs:95-142
//   CosineMetric.Compute            src/HNSWIndex/Metrics/CosineMetric.cs:10-92
// as invoked through GraphData.Distance (src/HNSWIndex/GraphData.cs:255-277) from the
// search / link loops.
//
// Numerical contract: the reference's AVX branch keeps EIGHT partial sums;
// element i feeds partial (i mod 8) in increasing i; L2 uses fma(d,d,acc) with d = a-b;
// dot/norm use acc + (a*b) (two roundings); the eight partials collapse through a fixed
// tree -- L2 ((p0+p4)+(p1+p5))+((p2+p6)+(p3+p7)), cosine family ((p0+p4)+(p2+p6))+((p1+p5)+(p3+p7))
// -- and a scalar mul-then-add tail handles dim % 8.  The kernels reproduce that order
// exactly, so distances are bit-identical to the CPU path and every float compare in the
// traversal branches the same way.  Built with -ffp-contract=off; every fused operation is
// written as __builtin_fmaf.
//
// Mapping to the wavefront: 8 lanes own the 8 partials of one candidate row, so a wave64
// evaluates 8 candidates at a time (up to 4 rows per lane group in one memory round trip); the
// collapse tree is three cross-lane adds.  Wider per-lane loads and lane-ring variants were
// measured no faster (tools/kbench.hip): this mapping gathers random 512-B rows at 4.2-4.9 TB/s.
//
// Contents: (1) slot_distance_kernel / pair_distance_kernel -- batched distances for a host that
// keeps the traversal (the hnswdev_dist_* entry points, the lock-step engine);  (2) the
// graph-resident traversals -- traverse_sorted (one sorted list in registers; the default) and
// traverse (the reference's two heaps in LDS; exact under equal distances), wrapped by the
// persistent graph_search_kernel (KnnQuery) and graph_insert_search_kernel (Add, search half +
// RelativeNeighborPruning);  (3) Add's link half -- link_plan / link_offsets / link_order
// (grouping of the back-edge appends on the device) and graph_link_kernel (appends and
// PruneOverflow with the tested-prefix shortcut);  (4) class Device: HBM matrix, graph mirror,
// per-wave scratch, launches.  DESIGN.md section 3 has the reasoning and the measurements.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "device_backend.h"

namespace hnsw {

// ------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------
static std::mutex g_err_mu;
static std::string g_err;
void set_dev_error(const std::string &msg)
{
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = msg;
}
std::string get_dev_error()
{
    std::lock_guard<std::mutex> lk(g_err_mu);
    return g_err;
}

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            set_dev_error(std::string(#expr) + ": " + hipGetErrorString(_e));                     \
            return false;                                                                         \
        }                                                                                         \
    } while (0)

// ------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------
enum { M_SQ = HNSWDEV_SQ_EUCLID, M_COS = HNSWDEV_COSINE, M_UCOS = HNSWDEV_UCOSINE };

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
__device__ double sqrt_rn(double x)
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
    float p = lane_chain<METRIC>(a, b, dim, j);
    float s = (METRIC == M_SQ) ? collapse_l2(p) : collapse_cos(p);
    if (dim & 7) s = scalar_tail<METRIC>(s, a, b, dim);
    if (METRIC == M_SQ) return s;
    if (METRIC == M_UCOS) return 1.0f - s; // CosineMetric.cs:141
    float denom = (float)(sa * sb);        // :88  (float)(Math.Sqrt(nA) * Math.Sqrt(nB))
    if (denom < 1e-30f) return 1.0f;       // :89-90
    return 1.0f - s / denom;               // :91
}

// One wave per search slot; inputs are the packed per-slot records (device_backend.h).
template <int METRIC>
__global__ void __launch_bounds__(256)
slot_distance_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn,
                     const float *__restrict__ queries, const double *__restrict__ q_sn, int dim,
                     const int *__restrict__ rec, float *__restrict__ out, int stride, int rec_stride, int nslots)
{
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nslots) return;
    const int *r = rec + (size_t)s * rec_stride;
    const int cnt = r[0];
    if (cnt <= 0) return;
    const int qraw = r[1];
    const int *sid = r + 2;
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
        const int id = sid[act ? c : c0]; // idle groups shadow a valid row and discard
        double sa = 0.0;
        if (METRIC == M_COS) sa = row_sn[id];
        float v = group_metric<METRIC>(rows + (size_t)id * dim, q, dim, j, sa, sb);
        if (act && j == 0) so[c] = v;
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

template <int METRIC>
__device__ __forceinline__ void measure_all(const float *rows, const double *row_sn, int dim, const float *qs, double sb,
                                            const int *nbuf, float *dbuf, int m, int lane)
{
    for (int p0 = 0; p0 < m; p0 += 32) {
        int left = m - p0;
        if (left > 24) measure_pass<METRIC, 4>(rows, row_sn, dim, qs, sb, nbuf, dbuf, p0, m, lane);
        else if (left > 16) measure_pass<METRIC, 3>(rows, row_sn, dim, qs, sb, nbuf, dbuf, p0, m, lane);
        else if (left > 8) measure_pass<METRIC, 2>(rows, row_sn, dim, qs, sb, nbuf, dbuf, p0, m, lane);
        else measure_pass<METRIC, 1>(rows, row_sn, dim, qs, sb, nbuf, dbuf, p0, m, lane);
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
    size_t b = sizeof(ND) * (size_t)(k + 1 + cand_cap) + sizeof(float) * (size_t)((dim + 3) & ~3) + 2u * 4u * (size_t)nbcap;
    if (heur) b += 2u * sizeof(float) * (size_t)((dim + 3) & ~3) + 4u * (size_t)nbcap + 4u * 3u * 40u;
    return b;
}
__device__ __forceinline__ SearchLds carve_lds(unsigned char *smem, int k, int cand_cap, int dim, int nbcap)
{
    SearchLds L;
    L.top = reinterpret_cast<ND *>(smem);
    L.cand = L.top + (k + 1);
    L.qs = reinterpret_cast<float *>(L.cand + cand_cap);
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
        for (;;) {
            const int old = atomicCAS(&tab[h], -1, id);
            if (old == -1) return true;
            if (old == id) return false;
            h = (h + 1) & tab_mask;
        }
    }
    __device__ __forceinline__ bool crowded() const { return HASHED && seen > limit; }
    __device__ __forceinline__ void clear(int lane)
    {
        __syncthreads();
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
        __syncthreads();
    }
};

#ifdef EXP_PHASE_CLOCKS // experiment build: shader-clock cycles per traversal phase, summed over waves
__device__ unsigned long long g_phase[12];
__device__ unsigned long long g_phase_link[12];
#define PH_FLUSH_LINK() do { if (lane == 0) for (int ph_i = 0; ph_i < 8; ++ph_i) atomicAdd(&g_phase_link[ph_i], (unsigned long long)ph_acc[ph_i]); } while (0)
#define PH_DECL() long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long ph_t = __builtin_readcyclecounter()
#define PH(i) do { long long ph_n = __builtin_readcyclecounter(); ph_acc[i] += ph_n - ph_t; ph_t = ph_n; } while (0)
#define PH_COUNT(i, v) ph_acc[i] += (v)
#define PH_FLUSH() do { if (lane == 0) for (int ph_i = 0; ph_i < 8; ++ph_i) atomicAdd(&g_phase[ph_i], (unsigned long long)ph_acc[ph_i]); } while (0)
#else
#define PH_DECL() do {} while (0)
#define PH(i) do {} while (0)
#define PH_COUNT(i, v) do {} while (0)
#define PH_FLUSH() do {} while (0)
#define PH_FLUSH_LINK() do {} while (0)
#endif

// FindEntryPoint / FindEntryAtLayer (GraphNavigator.cs:27-82): greedy descent from jb.entry at
// jb.entry_layer down to (not including) jb.search_layer.  Leaves the entry of the search layer
// in `best` and its distance in `cur` (both wave-uniform).
template <int METRIC>
__device__ __forceinline__ void descend(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, double sb,
                                        const GraphView &G, const SearchJob jb, const SearchLds &L, int lane, int &best, float &cur,
                                        unsigned long long &evals)
{
    int *nbuf = L.nbuf;
    float *dbuf = L.dbuf;
    const float *qs = L.qs;
    best = jb.entry;
    __syncthreads();
    if (lane == 0) nbuf[0] = best;
    __syncthreads();
    measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, 1, lane);
    __syncthreads();
    cur = dbuf[0]; // :57
    evals += 1;
    for (int layer = jb.entry_layer; layer > jb.search_layer; --layer) {
        bool changed = true;
        while (changed) { // :60
            changed = false;
            const int *l = G.list(best, layer);
            const int n = l[0];
            __syncthreads();
            for (int i = lane; i < n; i += 64) nbuf[i] = l[1 + i]; // :65 span taken once per pass
            __syncthreads();
            if (n > 0) measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, n, lane);
            __syncthreads();
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
// consumes in order (OrderBy + Take(k), Span.Sort).  (ii) and (iii) raise `tie` and the caller
// repeats the job with the exact two-heap traversal below.  After (i) the survivor (the reference
// may hold its twin instead -- same distance, other id, possibly still a candidate there) is only
// marked DOUBTFUL: the search goes on, and `tie` is raised if a doubtful entry is popped or is still
// in the list at the end; usually the next few insertions push it out and nothing depended on it.
// Equal distances elsewhere in the list are harmless.  Position p lives in lane p & 63 of register
// set p >> 6; id bit 31 = expanded, bit 30 = doubtful (node ids stay below 2^30).
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
template <int METRIC, int NS, bool HASHED>
__device__ __forceinline__ bool traverse_sorted(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, double sb,
                                                const GraphView &G, const SearchJob jb, int k, int ordered_prefix, VisitedSet<HASHED> &V,
                                                const SearchLds &L, int lane, int &top_n_out, bool &tie_out, unsigned long long &evals,
                                                bool overlap)
{
    int *nbuf = L.nbuf;
    float *dbuf = L.dbuf;
    const float *qs = L.qs;
    PH_DECL();
    int best;
    float cur;
    descend<METRIC>(rows, row_sn, dim, sb, G, jb, L, lane, best, cur, evals);
    // ---- SearchLayer (GraphNavigator.cs:123-189) ----
    const int layer = jb.search_layer;
    SortedTop<NS> T;
#pragma unroll
    for (int t = 0; t < NS; ++t) { T.key[t] = 0u; T.id[t] = 0; }
    int top_n = 0;
    bool unsafe = key_unsafe(cur); // NaN / -0 (see f2key)
    bool tie = false, hash_full = false;
    T.insert(f2key(cur), best, top_n, k, lane);                      // :134, :138
    if (lane == 0) (void)V.first_visit(best);                           // :140
    V.seen += 1;
    unsigned far_key = f2key(cur);                                   // farthestResultDist :135
    int pre_id = -1, pre_a = 0, pre_b = 0; // speculative prefetch of the next expansion's list (see traverse)
    const int lstride = layer == 0 ? G.stride0 : G.strideU;
    PH(0);
    constexpr int kDoubt = 0x40000000, kIdMask = 0x3fffffff;
    while (!unsafe && !tie) {
        const int pos = T.first_open(top_n, lane); // :146 closest candidate; none left <=> :147-150 / empty
        if (pos < 0) break;
        const HEnt c = T.at(pos);
        if (c.id & kDoubt) { tie = true; break; } // the reference may be expanding its twin instead
        T.mark(pos, lane);
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
        __syncthreads();
        PH(2);
        // candidate distances and ids of this expansion, one per lane, in adjacency order
        bool have = false;     // this lane holds an unvisited neighbour
        float lane_d = 0.0f;
        int lane_id = 0;
        const bool overlapped = overlap && n <= 64 && !HASHED;
        if (overlapped) {
            // Latency-bound launch (fewer jobs than resident waves): the rows of ALL listed neighbours
            // are fetched together with the visited atomics instead of after them -- one dependent
            // round trip less per expansion; rows of neighbours that turn out visited are wasted
            // bandwidth, of which such a launch has plenty.  Evaluations counted: the unvisited ones.
            const bool in = lane < n;
            if (in) nbuf[lane] = nb_a;
            __syncthreads();
            unsigned old = 0u;
            const unsigned bit = 1u << (nb_a & 31);
            if (in) old = atomicOr(&V.bits[nb_a >> 5], bit); // :181, in flight with the row loads below
            pre_id = -1;
            {
                const int nxt = T.first_open(top_n, lane);
                if (nxt >= 0) {
                    const HEnt e = T.at(nxt);
                    if (e.key == c.key) tie = true; // (ii)
                    pre_id = e.id & kIdMask;
                    const int *pl = G.list(pre_id, layer);
                    pre_a = lane < lstride ? pl[lane] : 0;
                    pre_b = lane + 64 < lstride ? pl[lane + 64] : 0;
                }
            }
            if (n > 0) measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, n, lane); // :163 (and the visited ones)
            __syncthreads();
            have = in && (old & bit) == 0u;
            const unsigned long long mask = __ballot(have);
            m = __popcll(mask);
            lane_d = in ? dbuf[lane] : 0.0f;
            lane_id = nb_a;
            PH(4);
            if (m == 0) continue;
            evals += (unsigned long long)m;
        } else {
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
            if (nxt >= 0) {
                const HEnt e = T.at(nxt);
                if (e.key == c.key) tie = true; // (ii): which of the two the reference pops first is a matter of heap layout
                pre_id = e.id & kIdMask;
                const int *pl = G.list(pre_id, layer);
                pre_a = lane < lstride ? pl[lane] : 0;
                pre_b = lane + 64 < lstride ? pl[lane + 64] : 0;
            }
        }
        __syncthreads();
        if (m == 0) continue;
        V.seen += m;
        if (V.crowded()) { hash_full = true; break; } // the id table is filling up: host traversal
        measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, m, lane); // :163
        __syncthreads();
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
            unsigned long long maybe = __ballot(valid && (top_n < k || my_key < far_key));
            while (maybe) {
                const int src = __builtin_ctzll(maybe);
                maybe &= maybe - 1;
                const unsigned dk = (unsigned)__builtin_amdgcn_readlane((int)my_key, src);
                if (top_n < k || dk < far_key) { // :165
                    const bool evicts = top_n == k;
                    T.insert(dk, __builtin_amdgcn_readlane(my_id, src), top_n, k, lane); // :168-174
                    if (top_n == k) {
                        const unsigned nf = T.key_at(k - 1);                             // :176-177
                        if (evicts && nf == far_key) T.mark_key(nf, top_n, lane, kDoubt); // (i): one of several equally far results was dropped
                        far_key = nf;
                    }
                }
            }
        }
        PH(5);
    }
    PH_FLUSH();
    // ToArray() for the callers: with distinct distances any order-insensitive consumer (OrderBy,
    // Span.Sort) sees the same thing; ascending order is also what they would produce
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int p = lane + 64 * t;
        if (p < top_n) { L.top[p].id = T.id[t] & kIdMask; L.top[p].dist = key2f(T.key[t]); }
    }
    __syncthreads();
    top_n_out = top_n;
    if (T.adjacent_equal(min(top_n, ordered_prefix), lane)) tie = true; // (iii)
    if (T.any_flagged(top_n, lane, kDoubt)) tie = true;                  // (i) left unresolved
    tie_out = tie;
    return !unsafe && !hash_full;
}

// Descent + beam search of one job; result = L.top[0..top_n) in heap order.  Returns false on
// candidate-heap overflow.  The query must already be staged in L.qs.
template <int METRIC, bool HASHED>
__device__ __forceinline__ bool traverse(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, double sb,
                                         const GraphView &G, const SearchJob jb, int k, int cand_cap, ND *spill, int spill_cap,
                                         VisitedSet<HASHED> &V, const SearchLds &L, int lane, int &top_n_out, unsigned long long &evals)
{
    const LdsHeap top{L.top};
    const SpillHeap cand{L.cand, cand_cap, spill};
    const int cand_limit = cand_cap + spill_cap;
    int *nbuf = L.nbuf;
    float *dbuf = L.dbuf;
    const float *qs = L.qs;
    // ---- FindEntryPoint / FindEntryAtLayer (GraphNavigator.cs:27-82) ----
    int best = jb.entry;
    __syncthreads();
    if (lane == 0) nbuf[0] = best;
    __syncthreads();
    measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, 1, lane);
    __syncthreads();
    float cur = dbuf[0]; // :57
    evals += 1;
    for (int layer = jb.entry_layer; layer > jb.search_layer; --layer) {
        bool changed = true;
        while (changed) { // :60
            changed = false;
            const int *l = G.list(best, layer);
            const int n = l[0];
            __syncthreads();
            for (int i = lane; i < n; i += 64) nbuf[i] = l[1 + i]; // :65 span taken once per pass
            __syncthreads();
            if (n > 0) measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, n, lane);
            __syncthreads();
            evals += (unsigned long long)n;
            for (int i = 0; i < n; ++i) { // :67-78
                float d = dbuf[i];
                if (d < cur) { cur = d; best = nbuf[i]; changed = true; }
            }
        }
    }
    // ---- SearchLayer (GraphNavigator.cs:123-189) ----
    const int layer = jb.search_layer;
    int top_n = 0, cand_n = 0;
    bool overflow = false; // also raised for NaN / -0 distances (see f2key)
    bool hash_full = false;
    best = __builtin_amdgcn_readfirstlane(best);
    cur = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cur)));
    if (key_unsafe(cur)) overflow = true;
    {
        HEnt e{best, f2key(cur)};
        heap_push<false>(top, top_n, e); // :134
        heap_push<true>(cand, cand_n, e); // :138
        if (lane == 0) (void)V.first_visit(best);                       // :140
            V.seen += 1;
    }
    unsigned far_key = f2key(cur); // farthestResultDist :135
    // Speculative prefetch of the NEXT expansion's out-edge list: while the current candidate
    // rows are in flight, lanes 0..stride fetch the list of the heap's current root.  If that
    // node is indeed popped next (it is, unless this expansion pushes something closer) its list
    // is already in registers and one dependent memory round trip disappears.
    int pre_id = -1, pre_a = 0, pre_b = 0;
    const int lstride = layer == 0 ? G.stride0 : G.strideU;
    while (cand_n > 0 && !overflow) {
        HEnt c = heap_pop<true>(cand, cand_n);          // :146
        if (c.key > far_key && top_n >= k) break;       // :147-150
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
        __syncthreads();
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
        __syncthreads();
        if (m == 0) continue;
        V.seen += m;
        if (V.crowded()) { hash_full = true; break; } // the id table is filling up: host traversal
        measure_all<METRIC>(rows, row_sn, dim, qs, sb, nbuf, dbuf, m, lane); // :163
        __syncthreads();
        evals += (unsigned long long)m;
        // Replay of the push loop (:165-178) in adjacency order.  farthest never grows once the
        // result heap is full, so a candidate that fails `d < farthest` now can never pass later:
        // only the lanes of the ballot are visited, and the exact test is repeated on each.
        for (int base = 0; base < m && !overflow; base += 64) {
            const int i = base + lane;
            const float my_d = (i < m) ? dbuf[i] : 0.0f;
            const int my_id = (i < m) ? nbuf[i] : 0;
            const unsigned my_key = f2key(my_d);
            if (__ballot(i < m && key_unsafe(my_d))) { overflow = true; break; }
            unsigned long long maybe = __ballot(i < m && (top_n < k || my_key < far_key));
            while (maybe) {
                const int src = __builtin_ctzll(maybe);
                maybe &= maybe - 1;
                const unsigned dk = (unsigned)__builtin_amdgcn_readlane((int)my_key, src);
                if (top_n < k || dk < far_key) { // :165
                    HEnt sel{__builtin_amdgcn_readlane(my_id, src), dk};
                    if (cand_n >= cand_limit) { overflow = true; break; }
                    heap_push<true>(cand, cand_n, sel);               // :168
                    heap_push<false>(top, top_n, sel);                // :171
                    if (top_n > k) (void)heap_pop<false>(top, top_n); // :173-174
                    far_key = top.get(0).key;                         // :176-177
                }
            }
        }
    }
    // back to float distances for the callers (ToArray(): heap order, BinaryHeap.cs:41-44)
    __syncthreads();
    for (int i = lane; i < top_n; i += 64) L.top[i].dist = key2f(__float_as_uint(L.top[i].dist));
    __syncthreads();
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

// Heuristic.RelativeNeighborPruning (Heuristic.cs:11-46) on cands[0..n) (LDS): writes the
// selected ids to L.acc, returns their count.  The candidate under test is staged in L.qs2
// and measured against ALL accepted rows at once (the reference's early break only skips
// evaluations).
template <int METRIC>
__device__ __forceinline__ int relative_neighbor_pruning(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                                                         ND *cands, int n, int max_edges, const SearchLds &L, int lane,
                                                         unsigned long long &evals, bool presorted = false)
{
    int *acc = L.acc;
    __syncthreads();
    if (n < max_edges) { // :13-18 input (heap) order, unsorted
        for (int i = lane; i < n; i += 64) acc[i] = cands[i].id;
        __syncthreads();
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
                __syncthreads();
                if (lane < n) cands[rank] = mine;
                ranked = true;
            }
        }
        if (!ranked) dev_dotnet_sort(cands, n, L.stk); // equal / NaN / -0 distances: the BCL introsort decides
    }
    __syncthreads();
    int rc = 0;
    // The row of candidate i + 1 is fetched while candidate i is being tested (registers, then the
    // other of two LDS buffers): one dependent memory round trip per candidate instead of two.
    constexpr int kPre = 4; // floats per lane: rows up to 256 floats; longer rows (bandwidth-bound anyway) are staged on demand
    const bool prefetch = dim <= 64 * kPre;
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
                __syncthreads();
            }
            // accepted ids are measured in chunks, in acceptance order, stopping at the first chunk
            // that rejects (the reference breaks at the first hit, :34; later pairs cannot change the
            // outcome) -- with long rows this saves most of the traffic of rejected candidates
            const int chunk = dim >= 512 ? 16 : 32;
            for (int a0 = 0; a0 < rc && ok; a0 += chunk) {
                const int an = min(chunk, rc - a0);
                measure_all<METRIC>(rows, row_sn, dim, buf[cur], sbc, acc + a0, L.dbuf, an, lane); // distanceFnc(s.Id, candidateId) :34
                __syncthreads();
                evals += (unsigned long long)an;
                const float dj = lane < an ? L.dbuf[lane] : 0.0f;
                ok = __ballot(lane < an && dj < c.dist) == 0ull;
                __syncthreads();
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
        __syncthreads();
    }
    return rc;
}

// NS > 0: sorted-list traversal with NS register sets (k <= 64 * NS); a wave that meets equal
// distances where the heap layout shows starts over with the exact two-heap traversal (out_flag 2,
// informational).  NS = 0: two-heap traversal only.
// One job on this wave.  `vis` / `spill`: the wave's own scratch (vis all zero on entry; the caller
// clears it afterwards).
template <int METRIC, int NS, bool HASHED>
__device__ __forceinline__ void search_job(const float *__restrict__ rows, const double *__restrict__ row_sn, const float *__restrict__ queries,
                    const double *__restrict__ q_sn, int dim, const int *__restrict__ adj0, int stride0,
                    const int64_t *__restrict__ upper, const int *__restrict__ pool, int strideU,
                    const SearchJob *__restrict__ jobs, int k, int cand_cap, ND *__restrict__ spill,
                    int spill_cap, VisitedSet<HASHED> &V, int k_out, int *__restrict__ out_ids,
                    float *__restrict__ out_d, int *__restrict__ out_cnt, int *__restrict__ out_flag,
                    unsigned long long *__restrict__ eval_counter, int nbcap, unsigned char *smem, int job, bool overlap)
{
    const SearchLds L = carve_lds(smem, k, cand_cap, dim, nbcap);
    const int lane = threadIdx.x;
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
    unsigned long long evals = 0;
    int top_n = 0;
    bool repeated = false;
    if constexpr (NS > 0) {
        bool tie = false;
        // OrderBy + Take(k_out) reads k_out entries in order and decides between entries k_out - 1 and k_out
        const bool ok1 = traverse_sorted<METRIC, NS, HASHED>(rows, row_sn, dim, sb, G, jb, k, k_out + 1, V, L, lane, top_n, tie, evals, overlap);
        if (!(ok1 && tie)) {
            // KnnQuery's tail (HNSWIndex.cs:119-123): OrderBy(Dist).Take(k) of distinct distances is the
            // head of the ascending list; missing results are padded (HNSWIndexExports.cs:144)
            for (int r = lane; r < k_out; r += 64) {
                const bool have = r < top_n;
                out_ids[(size_t)job * k_out + r] = have ? L.top[r].id : -1;
                out_d[(size_t)job * k_out + r] = have ? L.top[r].dist : __uint_as_float(0x7fc00000u);
            }
            if (lane == 0) {
                out_cnt[job] = ok1 ? top_n : 0;
                out_flag[job] = ok1 ? 0 : 1;
                atomicAdd(eval_counter, evals);
            }
            return;
        }
        // equal distances where the heap layout shows: this wave starts over with the exact traversal
        V.clear(lane);
        evals = 0;
        top_n = 0;
        repeated = true;
    }
    const bool ok = traverse<METRIC, HASHED>(rows, row_sn, dim, sb, G, jb, k, cand_cap, spill, spill_cap, V, L, lane, top_n, evals);
    // KnnQuery's tail (HNSWIndex.cs:119-123): OrderBy(c => c.Dist) is a STABLE sort over the heap
    // array (ToArray(), BinaryHeap.cs:41-44) and only the first k_out survive -- so select the
    // k_out smallest (float.CompareTo order: NaN first, -0 == +0) with ties broken by array index:
    // exactly the stable sort's prefix.  Key = (order-preserving bits << 32) | index, wave min.
    __syncthreads();
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
template <int METRIC, int NS, bool HASHED>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NS <= 4 ? 3 : 2))) // 168 VGPRs: three waves per SIMD
graph_search_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, const float *__restrict__ queries,
                    const double *__restrict__ q_sn, int dim, const int *__restrict__ adj0, int stride0,
                    const int64_t *__restrict__ upper, const int *__restrict__ pool, int strideU,
                    const SearchJob *__restrict__ jobs, int k, int cand_cap, ND *__restrict__ spill,
                    int spill_cap, unsigned *__restrict__ visited, long long vis_words, int *__restrict__ vis_tab, int vis_tab_cap, int k_out,
                    int *__restrict__ out_ids, float *__restrict__ out_d, int *__restrict__ out_cnt, int *__restrict__ out_flag,
                    unsigned long long *__restrict__ eval_counter, int nbcap, int njobs, int *__restrict__ job_counter, int overlap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    VisitedSet<HASHED> V{visited + (size_t)blockIdx.x * (size_t)vis_words, vis_words,
                 vis_tab ? vis_tab + (size_t)blockIdx.x * (size_t)vis_tab_cap : nullptr, (unsigned)(vis_tab_cap - 1), 0, vis_tab_cap / 4 * 3};
    ND *my_spill = spill + (size_t)blockIdx.x * spill_cap;
    for (;;) {
        int job = 0;
        if (lane == 0) job = atomicAdd(job_counter, 1);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= njobs) break;
        search_job<METRIC, NS, HASHED>(rows, row_sn, queries, q_sn, dim, adj0, stride0, upper, pool, strideU, jobs, k, cand_cap, my_spill, spill_cap,
                               V, k_out, out_ids, out_d, out_cnt, out_flag, eval_counter, nbcap, smem, job, overlap != 0);
        V.clear(lane);
    }
}

// Insert, search half, fused: for one new item, GraphConnector.AddNewConnections' whole loop
// (GraphConnector.cs:172-181): FindEntryPoint, then for every layer of the item ConnectAtLayer's
// SearchLayer + RelativeNeighborPruning (:189-190) with the next layer's entry = selected[0]
// (:216).  One launch serves every layer of every item of a batch (the few multi-layer items
// clear their visited bitset between layers).  Output per (job, layer): the selected ids in
// selection order (layer 0 -> slot `job`; layer L >= 1 -> upper slot jobs[].aux + L - 1).
// jobs[].search_layer = the item's first layer min(level, top).
template <int METRIC, int NS, bool HASHED>
__device__ __forceinline__ void insert_job(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                           const int *__restrict__ adj0, int stride0, const int64_t *__restrict__ upper,
                           const int *__restrict__ pool, int strideU, const SearchJob *__restrict__ jobs, int k,
                           int cand_cap, ND *__restrict__ spill, int spill_cap, int max_edges0, VisitedSet<HASHED> &V,
                           int *__restrict__ out_sel0, int *__restrict__ out_cnt0, int *__restrict__ out_selU,
                           int *__restrict__ out_cntU, int sel_stride, int *__restrict__ out_flag,
                           unsigned long long *__restrict__ eval_counter, int nbcap, unsigned char *smem, int job, bool overlap)
{
    const SearchLds L = carve_lds(smem, k, cand_cap, dim, nbcap);
    const int lane = threadIdx.x;
    SearchJob jb = jobs[job];
    const GraphView G{adj0, stride0, upper, pool, strideU};
    const int item = ~jb.qref;
    const float *q = rows + (size_t)item * dim;
    double sb = 0.0;
    if (METRIC == M_COS) sb = row_sn[item];
    for (int i = lane; i < dim; i += 64) L.qs[i] = q[i];
    unsigned long long evals = 0;
    bool ok = true, repeat = false;
    const int first_layer = jb.search_layer;
    for (int layer = first_layer; layer >= 0 && ok; --layer) {
        if (layer != first_layer) V.clear(lane); // a fresh SearchLayer: new visited list (VisitedListPool.cs:74-106)
        int top_n = 0;
        const int max_edges = layer == 0 ? max_edges0 : (max_edges0 >> 1); // GraphData.MaxEdges :247-250
        bool exact = NS == 0;
        if constexpr (NS > 0) {
            bool tie = false;
            const unsigned long long ev0 = evals;
            ok = traverse_sorted<METRIC, NS, HASHED>(rows, row_sn, dim, sb, G, jb, k, k, V, L, lane, top_n, tie, evals, overlap); // Span.Sort consumes all
            if (!ok) break;
            // equal distances where the heap layout shows, or fewer candidates than MaxEdges (the heuristic
            // then returns them in HEAP order, Heuristic.cs:13-18): this layer again, exact traversal
            if (tie || top_n < max_edges) {
                exact = true;
                repeat = true;
                evals = ev0;
                top_n = 0;
                V.clear(lane);
            }
        }
        if (exact) {
            ok = traverse<METRIC, HASHED>(rows, row_sn, dim, sb, G, jb, k, cand_cap, spill, spill_cap, V, L, lane, top_n, evals);
            if (!ok) break;
        }
        const int rc = relative_neighbor_pruning<METRIC>(rows, row_sn, dim, L.top, top_n, max_edges, L, lane, evals, !exact);
        int *osel = layer == 0 ? out_sel0 + (size_t)job * sel_stride : out_selU + (size_t)(jb.aux + layer - 1) * sel_stride;
        for (int i = lane; i < rc; i += 64) osel[i] = L.acc[i];
        if (lane == 0) { if (layer == 0) out_cnt0[job] = rc; else out_cntU[jb.aux + layer - 1] = rc; }
        const int next_entry = __builtin_amdgcn_readfirstlane(L.acc[0]); // :216 selected[0] -> bestPeer of the next layer (:179)
        jb.entry = next_entry;
        jb.entry_layer = layer - 1;
        jb.search_layer = layer - 1;
        __syncthreads();
    }
    if (lane == 0) {
        out_flag[job] = ok ? (repeat ? 2 : 0) : 1; // 2: informational (a layer was answered by the exact traversal)
        atomicAdd(eval_counter, evals);
    }
}

template <int METRIC, int NS, bool HASHED>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NS <= 4 ? 3 : 2))) // up to 256 candidates: 168 VGPRs, three waves per SIMD
graph_insert_search_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                           const int *__restrict__ adj0, int stride0, const int64_t *__restrict__ upper,
                           const int *__restrict__ pool, int strideU, const SearchJob *__restrict__ jobs, int k,
                           int cand_cap, ND *__restrict__ spill, int spill_cap, int max_edges0, unsigned *__restrict__ visited, long long vis_words,
                           int *__restrict__ vis_tab, int vis_tab_cap, int *__restrict__ out_sel0, int *__restrict__ out_cnt0, int *__restrict__ out_selU,
                           int *__restrict__ out_cntU, int sel_stride, int *__restrict__ out_flag,
                           unsigned long long *__restrict__ eval_counter, int nbcap, int njobs, int *__restrict__ job_counter, int overlap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    VisitedSet<HASHED> V{visited + (size_t)blockIdx.x * (size_t)vis_words, vis_words,
                 vis_tab ? vis_tab + (size_t)blockIdx.x * (size_t)vis_tab_cap : nullptr, (unsigned)(vis_tab_cap - 1), 0, vis_tab_cap / 4 * 3};
    ND *my_spill = spill + (size_t)blockIdx.x * spill_cap;
    for (;;) { // persistent, see graph_search_kernel
        int job = 0;
        if (lane == 0) job = atomicAdd(job_counter, 1);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= njobs) break;
        insert_job<METRIC, NS, HASHED>(rows, row_sn, dim, adj0, stride0, upper, pool, strideU, jobs, k, cand_cap, my_spill, spill_cap, max_edges0, V,
                               out_sel0, out_cnt0, out_selU, out_cntU, sel_stride, out_flag, eval_counter, nbcap, smem, job, overlap != 0);
        V.clear(lane);
    }
}

// Insert, link half, on the HBM mirror.  (a) new nodes' own lists.
__global__ void __launch_bounds__(64)
graph_write_rows_kernel(int *__restrict__ adj0, int stride0, const int64_t *__restrict__ upper, int *__restrict__ pool,
                        int strideU, const int *__restrict__ recs, int row_stride, int *__restrict__ tested0,
                        int *__restrict__ testedU, int max_edges0)
{
    const int *r = recs + (size_t)blockIdx.x * row_stride;
    const int node = r[0], layer = r[1], cnt = r[2];
    int *l = layer == 0 ? adj0 + (size_t)node * stride0 : pool + upper[node] + (size_t)(layer - 1) * strideU;
    if (threadIdx.x == 0) {
        l[0] = cnt;
        // a full list can only be the ordered output of the heuristic's greedy pass (fewer candidates
        // than MaxEdges come back unsorted, Heuristic.cs:13-18): its entries are mutually tested
        const int me = layer == 0 ? max_edges0 : (max_edges0 >> 1);
        int *t = layer == 0 ? tested0 + node : testedU + (upper[node] / strideU + (layer - 1));
        *t = cnt == me ? cnt : 0;
    }
    for (int i = threadIdx.x; i < cnt; i += 64) l[1 + i] = r[3 + i];
}

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
                  unsigned char *smem)
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
    __syncthreads();
    unsigned long long evals = 0;
    PH_DECL();
    PH(0);
    for (int item = next_item(); item >= 0; item = next_item()) {
        if (lane == 0) L.nbuf[cnt] = item; // :207
        cnt++;
        __syncthreads();
        PH_COUNT(6, 1);
        if (cnt > max_edges) { // :209
            measure_all<METRIC>(rows, row_sn, dim, L.qs, sb, L.nbuf, L.dbuf, cnt, lane); // Distance(cand, node.Id) :233
            __syncthreads();
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
                        __syncthreads();
                        for (int t2 = lane; t2 < dim; t2 += 64) L.qs2[t2] = xrow[t2];
                        double sbx = 0.0;
                        if (METRIC == M_COS) sbx = row_sn[xid];
                        __syncthreads();
                        // a single new entry only meets the old ones (one pass of <= 32 rows instead of two)
                        const int mrows = u == 1 ? tested : n;
                        measure_all<METRIC>(rows, row_sn, dim, L.qs2, sbx, L.nbuf, Dm + jn * nbcap, mrows, lane);
                        evals += (unsigned long long)(u == 1 ? mrows : n - 1);
                    }
                    __syncthreads();
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
                    __syncthreads();
                }
            }
            if (rc < 0) {
                for (int i = lane; i < cnt; i += 64) L.top[i] = ND{L.nbuf[i], L.dbuf[i]};
                rc = relative_neighbor_pruning<METRIC>(rows, row_sn, dim, L.top, cnt, max_edges, L, lane, evals);
            }
            for (int i = lane; i < rc; i += 64) L.nbuf[i] = L.acc[i]; // node.OutEdges[layer] = newOut :236
            cnt = rc;
            tested = rc; // the whole list is a greedy output now
            __syncthreads();
        }
    }
    if (lane == 0) { l[0] = cnt; *tested_p = tested; if (out_list) out_list[0] = cnt; }
    for (int i = lane; i < cnt; i += 64) { l[1 + i] = L.nbuf[i]; if (out_list) out_list[1 + i] = L.nbuf[i]; }
    if (lane == 0) atomicAdd(eval_counter, evals);
    __syncthreads();
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
// one block per group: its items (batch positions, filed in arbitrary order) become node ids in batch
// order -- repeatedly the smallest position not yet taken; groups are tiny -- and the slot's
// counters return to zero for the next batch
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

// Flat id<->id pairs: 8 lanes per pair (hnswdev_dist_pair_batch).
template <int METRIC>
__global__ void __launch_bounds__(256)
pair_distance_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim,
                     const int *__restrict__ a_ids, const int *__restrict__ b_ids, float *__restrict__ out, int n)
{
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 3;
    const int j = threadIdx.x & 7;
    const bool act = g < n;
    const int a = a_ids[act ? g : 0], b = b_ids[act ? g : 0];
    double sa = 0.0, sb = 0.0;
    if (METRIC == M_COS) { sa = row_sn[a]; sb = row_sn[b]; }
    float r = group_metric<METRIC>(rows + (size_t)a * dim, rows + (size_t)b * dim, dim, j, sa, sb);
    if (act && j == 0) out[g] = r;
}

// sqrt((double)|row|^2) with |row|^2 summed in f32 in the reference's lane order
// (CosineMetric.cs:40-41,47 / :43-44,48 and the tail :83-84): 8 lanes per row.
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

// exposed for tests: sqrt_rn over an array
__global__ void sqrt_rn_kernel(const double *in, double *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sqrt_rn(in[i]);
}

// ------------------------------------------------------------------------------------
// host side of the context
// ------------------------------------------------------------------------------------
// C-ABI graph staging (hnswdev_graph_*): the host graph flattened layer by layer
struct Device::HostGraphStage {
    int n = 0, M = 0, stride0 = 0, strideU = 0, top = 0;
    std::vector<int> level, adj0, pool;
    std::vector<int64_t> upper;
};

static inline hipStream_t S(void *p) { return (hipStream_t)p; }

bool Device::bind()
{
    HIP_OK(hipSetDevice(device_));
    return true;
}

Device *Device::create(int device, int dim, int metric, long long capacity)
{
    if (dim <= 0 || metric < 0 || metric > 2 || capacity < 0) {
        set_dev_error("hnswdev_create: bad argument");
        return nullptr;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_dev_error(std::string("no HIP device available (") + hipGetErrorString(e) +
                      "): this library has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        set_dev_error("hnswdev_create: device ordinal out of range");
        return nullptr;
    }
    Device *d = new Device();
    d->device_ = device;
    d->dim_ = dim;
    d->metric_ = metric;
    d->stats_.row_bytes = (uint64_t)dim * sizeof(float);
    auto fail = [&]() -> Device * { delete d; return nullptr; };
    if (hipSetDevice(device) != hipSuccess) { set_dev_error("hipSetDevice failed"); return fail(); }
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { set_dev_error("hipStreamCreate failed"); return fail(); }
    d->stream_ = st;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) d->num_cu_ = cus;
    }
    if (!d->reserve(capacity > 0 ? capacity : 1)) return fail();
    return d;
}

Device::~Device()
{
    if (hipSetDevice(device_) != hipSuccess) return;
    if (stream_) { (void)hipStreamSynchronize(S(stream_)); (void)hipStreamDestroy(S(stream_)); }
    if (d_rows_) (void)hipFree(d_rows_);
    if (d_row_sn_) (void)hipFree(d_row_sn_);
    if (d_queries_) (void)hipFree(d_queries_);
    if (d_q_sn_) (void)hipFree(d_q_sn_);
#ifdef EXP_PHASE_CLOCKS
    {
        unsigned long long h[12] = {0};
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof h) == hipSuccess) {
            double tot = 0;
            for (int i = 0; i < 6; ++i) tot += (double)h[i];
            unsigned long long hl[12] = {0};
            (void)hipMemcpyFromSymbol(hl, HIP_SYMBOL(g_phase_link), sizeof hl);
            double tl = 0;
            for (int i = 0; i < 5; ++i) tl += (double)hl[i];
            fprintf(stderr, "[phase clocks, link] stage %.1f%% measure %.1f%% sort %.1f%% heuristic %.1f%% rest %.1f%% | appends %llu, prunes %llu, cycles/prune %.0f\n",
                    100 * hl[0] / tl, 100 * hl[1] / tl, 100 * hl[2] / tl, 100 * hl[3] / tl, 100 * hl[4] / tl, hl[6], hl[7], tl / (double)std::max(1ull, hl[7]));
            fprintf(stderr, "[phase clocks] descent %.1f%% pop %.1f%% list %.1f%% visited %.1f%% rows %.1f%% push %.1f%% | expansions %llu, prefetch hits %llu (%.1f%%), cycles/expansion %.0f\n",
                    100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot, 100 * h[4] / tot, 100 * h[5] / tot, h[7], h[6],
                    100.0 * h[6] / (double)std::max(1ull, h[7]), tot / (double)std::max(1ull, h[7]));
        }
    }
#endif
    for (void *p : {(void *)g_adj0_, (void *)g_level_, (void *)g_upper_, (void *)g_pool_, (void *)g_tested0_, (void *)g_testedU_, (void *)s_visited_, (void *)s_jobs_,
                    (void *)s_hits_, (void *)s_cnt_, (void *)s_flag_, (void *)s_jobctr_, (void *)s_vistab_, (void *)lp_slot_[0], (void *)lp_slot_[1], (void *)lp_slot_[2], (void *)lp_grp_[0], (void *)lp_grp_[1], (void *)lp_grp_[2], (void *)lp_grp_[3], (void *)lp_grp_[4], (void *)lp_grp_[5], (void *)lp_counters_, (void *)s_evals_, (void *)s_sel_, (void *)s_lcnt_, (void *)s_selU_, (void *)s_cntU_, (void *)s_iflag_, (void *)s_lk_[0], (void *)s_lk_[1], (void *)s_lk_[2], (void *)s_lk_[3], (void *)s_lk_[4], (void *)s_spill_})
        if (p) (void)hipFree(p);
    if (ev0_) (void)hipEventDestroy((hipEvent_t)ev0_);
    if (ev1_) (void)hipEventDestroy((hipEvent_t)ev1_);
    if (h_stage_) (void)hipHostFree(h_stage_);
    if (h_res_) (void)hipHostFree(h_res_);
    for (LinkSet &ls : lset_) {
        if (ls.h_in) (void)hipHostFree(ls.h_in);
        if (ls.h_out) (void)hipHostFree(ls.h_out);
        if (ls.h_ev) (void)hipHostFree(ls.h_ev);
        for (void *e : {ls.ev_start, ls.ev_stop, ls.ev_done}) if (e) (void)hipEventDestroy((hipEvent_t)e);
    }
    delete hg_;
}

bool Device::reserve(long long capacity)
{
    if (capacity <= capacity_) return true;
    if (!bind()) return false;
    float *nr = nullptr;
    double *nsn = nullptr;
    HIP_OK(hipMalloc(&nr, (size_t)capacity * dim_ * sizeof(float)));
    if (metric_ == M_COS) HIP_OK(hipMalloc(&nsn, (size_t)capacity * sizeof(double)));
    if (d_rows_) {
        HIP_OK(hipMemcpyAsync(nr, d_rows_, (size_t)capacity_ * dim_ * sizeof(float), hipMemcpyDeviceToDevice, S(stream_)));
        if (nsn) HIP_OK(hipMemcpyAsync(nsn, d_row_sn_, (size_t)capacity_ * sizeof(double), hipMemcpyDeviceToDevice, S(stream_)));
        HIP_OK(hipStreamSynchronize(S(stream_)));
        HIP_OK(hipFree(d_rows_));
        if (d_row_sn_) HIP_OK(hipFree(d_row_sn_));
    }
    d_rows_ = nr;
    d_row_sn_ = nsn;
    capacity_ = capacity;
    return true;
}

bool Device::upload_rows(int first_id, int n, const float *rows)
{
    if (n <= 0) return true;
    if (first_id < 0 || (long long)first_id + n > capacity_ || !rows) {
        set_dev_error("upload_rows: range outside capacity");
        return false;
    }
    if (!bind()) return false;
    {   // pageable -> pinned bounce buffer -> HBM, 64 MiB at a time
        const size_t row_bytes = (size_t)dim_ * sizeof(float);
        const size_t chunk_rows = std::max<size_t>(1, (64u << 20) / row_bytes);
        char *hs = static_cast<char *>(pinned_stage(std::min<size_t>((size_t)n, chunk_rows) * row_bytes));
        if (!hs) return false;
        for (size_t r0 = 0; r0 < (size_t)n; r0 += chunk_rows) {
            const size_t nr = std::min(chunk_rows, (size_t)n - r0);
            memcpy(hs, rows + r0 * dim_, nr * row_bytes);
            HIP_OK(hipMemcpyAsync(d_rows_ + ((size_t)first_id + r0) * dim_, hs, nr * row_bytes, hipMemcpyHostToDevice, S(stream_)));
            HIP_OK(hipStreamSynchronize(S(stream_))); // the bounce buffer is reused
        }
    }
    if (metric_ == M_COS) {
        int blocks = (int)(((long long)n * 8 + 255) / 256);
        hipLaunchKernelGGL(row_sqrtnorm_kernel, dim3(blocks), dim3(256), 0, S(stream_), d_rows_, dim_, (long long)first_id, n, d_row_sn_);
        HIP_OK(hipGetLastError());
    }
    HIP_OK(hipStreamSynchronize(S(stream_))); // `rows` is borrowed only for this call
    n_rows_hw_ = std::max(n_rows_hw_, (long long)first_id + n);
    return true;
}

bool Device::download_rows(int first_id, int n, float *rows)
{
    if (n <= 0) return true;
    if (first_id < 0 || (long long)first_id + n > capacity_ || !rows) {
        set_dev_error("download_rows: range outside capacity");
        return false;
    }
    if (!bind()) return false;
    HIP_OK(hipMemcpyAsync(rows, d_rows_ + (size_t)first_id * dim_, (size_t)n * dim_ * sizeof(float), hipMemcpyDeviceToHost, S(stream_)));
    HIP_OK(hipStreamSynchronize(S(stream_)));
    return true;
}

bool Device::set_queries(const float *queries, int nq)
{
    if (nq < 0 || (nq > 0 && !queries)) { set_dev_error("set_queries: bad argument"); return false; }
    if (!bind()) return false;
    if (nq > q_capacity_) {
        if (d_queries_) HIP_OK(hipFree(d_queries_));
        if (d_q_sn_) HIP_OK(hipFree(d_q_sn_));
        d_queries_ = nullptr; d_q_sn_ = nullptr;
        long long cap = std::max<long long>(nq, 1024);
        HIP_OK(hipMalloc(&d_queries_, (size_t)cap * dim_ * sizeof(float)));
        if (metric_ == M_COS) HIP_OK(hipMalloc(&d_q_sn_, (size_t)cap * sizeof(double)));
        q_capacity_ = cap;
    }
    n_queries_ = nq;
    if (nq == 0) return true;
    {
        const size_t bytes = (size_t)nq * dim_ * sizeof(float);
        void *hs = bytes <= (256u << 20) ? pinned_stage(bytes) : nullptr;
        if (hs) { memcpy(hs, queries, bytes); HIP_OK(hipMemcpyAsync(d_queries_, hs, bytes, hipMemcpyHostToDevice, S(stream_))); }
        else HIP_OK(hipMemcpyAsync(d_queries_, queries, bytes, hipMemcpyHostToDevice, S(stream_)));
    }
    if (metric_ == M_COS) {
        int blocks = (int)(((long long)nq * 8 + 255) / 256);
        hipLaunchKernelGGL(row_sqrtnorm_kernel, dim3(blocks), dim3(256), 0, S(stream_), d_queries_, dim_, 0LL, nq, d_q_sn_);
        HIP_OK(hipGetLastError());
    }
    HIP_OK(hipStreamSynchronize(S(stream_)));
    return true;
}

StepBuffers *Device::alloc_step(int nslots, int stride)
{
    if (nslots <= 0 || stride <= 0) { set_dev_error("alloc_step: bad argument"); return nullptr; }
    if (!bind()) return nullptr;
    StepBuffers *sb = new StepBuffers();
    sb->nslots = nslots;
    sb->stride = stride;
    sb->rec_stride = stride + 2;
    const size_t rec_bytes = sizeof(int) * (size_t)nslots * sb->rec_stride;
    const size_t dist_bytes = sizeof(float) * (size_t)nslots * stride;
    bool ok = hipHostMalloc((void **)&sb->rec, rec_bytes, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void **)&sb->dist, dist_bytes, hipHostMallocDefault) == hipSuccess &&
              hipMalloc((void **)&sb->d_rec, rec_bytes) == hipSuccess &&
              hipMalloc((void **)&sb->d_dist, dist_bytes) == hipSuccess;
    if (ok) { memset(sb->rec, 0, rec_bytes); memset(sb->dist, 0, dist_bytes); }
    hipEvent_t ev = nullptr, t0 = nullptr, t1 = nullptr;
    ok = ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess &&
         hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess;
    sb->done = ev; sb->t0 = t0; sb->t1 = t1;
    if (!ok) {
        set_dev_error("alloc_step: allocation failed");
        free_step(sb);
        return nullptr;
    }
    return sb;
}

void Device::free_step(StepBuffers *sb)
{
    if (!sb) return;
    (void)hipSetDevice(device_);
    if (sb->rec) (void)hipHostFree(sb->rec);
    if (sb->dist) (void)hipHostFree(sb->dist);
    if (sb->d_rec) (void)hipFree(sb->d_rec);
    if (sb->d_dist) (void)hipFree(sb->d_dist);
    if (sb->done) (void)hipEventDestroy((hipEvent_t)sb->done);
    if (sb->t0) (void)hipEventDestroy((hipEvent_t)sb->t0);
    if (sb->t1) (void)hipEventDestroy((hipEvent_t)sb->t1);
    delete sb;
}

bool Device::launch_step(StepBuffers *sb, int nslots_used, uint64_t evals)
{
    sb->in_flight = false;
    if (nslots_used <= 0 || evals == 0) { sb->evals = 0; sb->timed = false; return true; }
    if (nslots_used > sb->nslots) { set_dev_error("launch_step: too many slots"); return false; }
    hipStream_t st = S(stream_);
    sb->timed = profiling_;
    sb->evals = evals;
    HIP_OK(hipMemcpyAsync(sb->d_rec, sb->rec, sizeof(int) * (size_t)nslots_used * sb->rec_stride, hipMemcpyHostToDevice, st));
    if (sb->timed) HIP_OK(hipEventRecord((hipEvent_t)sb->t0, st));
    dim3 grid((nslots_used + 3) / 4), block(256);
#define LAUNCH(M)                                                                                          \
    hipLaunchKernelGGL(slot_distance_kernel<M>, grid, block, 0, st, d_rows_, d_row_sn_, d_queries_, d_q_sn_, dim_, \
                       sb->d_rec, sb->d_dist, sb->stride, sb->rec_stride, nslots_used)
    if (metric_ == M_SQ) LAUNCH(M_SQ);
    else if (metric_ == M_COS) LAUNCH(M_COS);
    else LAUNCH(M_UCOS);
#undef LAUNCH
    HIP_OK(hipGetLastError());
    if (sb->timed) HIP_OK(hipEventRecord((hipEvent_t)sb->t1, st));
    HIP_OK(hipMemcpyAsync(sb->dist, sb->d_dist, sizeof(float) * (size_t)nslots_used * sb->stride, hipMemcpyDeviceToHost, st));
    HIP_OK(hipEventRecord((hipEvent_t)sb->done, st));
    sb->in_flight = true;
    stats_.launches++;
    stats_.evals += evals;
    return true;
}

bool Device::wait_step(StepBuffers *sb)
{
    if (!sb->in_flight) return true;
    HIP_OK(hipEventSynchronize((hipEvent_t)sb->done));
    sb->in_flight = false;
    if (sb->timed) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)sb->t0, (hipEvent_t)sb->t1));
        stats_.kernel_ms += ms;
        stats_.timed_launches++;
        stats_.timed_evals += sb->evals;
        sb->timed = false;
    }
    sb->evals = 0;
    return true;
}

bool Device::sync()
{
    if (!bind()) return false;
    HIP_OK(hipStreamSynchronize(S(stream_)));
    return true;
}

void Device::get_stats(hnswdev_stats *out) { *out = stats_; }
void Device::reset_stats()
{
    uint64_t rb = stats_.row_bytes;
    stats_ = hnswdev_stats{};
    stats_.row_bytes = rb;
#ifdef EXP_PHASE_CLOCKS
    unsigned long long z[12] = {0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z);
#endif
}


// ---- graph mirror + graph-resident search -------------------------------------------------
bool Device::set_graph(const int *adj0, long long n, int stride0, const int *level, const int64_t *upper, const int *pool,
                       long long pool_len, int strideU)
{
    if (n < 0 || (n > 0 && (!adj0 || !level || !upper))) { set_dev_error("set_graph: bad argument"); return false; }
    if (stride0 - 1 > 128 || strideU - 1 > 128) { set_dev_error("set_graph: MaxEdges > 63 is not supported by the graph-resident kernels (use hnsw_mi355x_set_device_traversal(0))"); return false; }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    if (n > g_cap_n_ || stride0 != g_stride0_) {
        if (g_adj0_) HIP_OK(hipFree(g_adj0_));
        if (g_level_) HIP_OK(hipFree(g_level_));
        if (g_upper_) HIP_OK(hipFree(g_upper_));
        if (g_tested0_) HIP_OK(hipFree(g_tested0_));
        g_adj0_ = nullptr; g_level_ = nullptr; g_upper_ = nullptr; g_tested0_ = nullptr;
        long long cap = std::max<long long>(n, std::max<long long>(capacity_, 1024));
        HIP_OK(hipMalloc(&g_adj0_, sizeof(int) * (size_t)cap * stride0));
        HIP_OK(hipMalloc(&g_level_, sizeof(int) * (size_t)cap));
        HIP_OK(hipMalloc(&g_upper_, sizeof(int64_t) * (size_t)cap));
        HIP_OK(hipMalloc(&g_tested0_, sizeof(int) * (size_t)cap));
        g_cap_n_ = cap;
    }
    if (pool_len > g_pool_cap_) {
        if (g_pool_) HIP_OK(hipFree(g_pool_));
        if (g_testedU_) HIP_OK(hipFree(g_testedU_));
        g_pool_ = nullptr; g_testedU_ = nullptr;
        long long cap = std::max<long long>(pool_len * 2, 4096);
        HIP_OK(hipMalloc(&g_pool_, sizeof(int) * (size_t)cap));
        HIP_OK(hipMalloc(&g_testedU_, sizeof(int) * (size_t)cap)); // indexed by list offset / strideU: never more than cap
        g_pool_cap_ = cap;
    }
    g_n_ = n; g_stride0_ = stride0; g_strideU_ = strideU;
    if (n > 0) {
        HIP_OK(hipMemcpyAsync(g_adj0_, adj0, sizeof(int) * (size_t)n * stride0, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(g_level_, level, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(g_upper_, upper, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
    }
    if (pool_len > 0) HIP_OK(hipMemcpyAsync(g_pool_, pool, sizeof(int) * (size_t)pool_len, hipMemcpyHostToDevice, st));
    // lists that arrive from the host carry no pruning history
    if (g_tested0_) HIP_OK(hipMemsetAsync(g_tested0_, 0, sizeof(int) * (size_t)g_cap_n_, st));
    if (g_testedU_) HIP_OK(hipMemsetAsync(g_testedU_, 0, sizeof(int) * (size_t)g_pool_cap_, st));
    HIP_OK(hipStreamSynchronize(st)); // host arrays are borrowed only for this call
    return true;
}

// LDS part of the candidate heap: sized for the common case (4 x beam width; C2 queries peak
// near 500 entries at ef = 128), the rest spills to HBM (SpillHeap).  A smaller LDS footprint means
// more resident waves to hide memory latency: 7.6 -> 6.1 ms per 10k-query launch going from 1024
// to 512 entries.  Beyond LDS + spill capacity the traversal is flagged for the lock-step path.
// Register sets of the sorted-list traversal (SortedTop<NS>: k <= 64 * NS); 0 = two-heap traversal
// only.  HNSW_MI355X_SORTED_TOP=0 forces the latter (the tests run both).
static int sorted_top_sets(int k)
{
    const char *e = std::getenv("HNSW_MI355X_SORTED_TOP");
    if (e && std::atoi(e) == 0) return 0;
    return k <= 64 ? 1 : k <= 128 ? 2 : k <= 256 ? 4 : k <= 512 ? 8 : 0;
}
constexpr long long kSortedTopMaxNodes = 1LL << 30; // the sorted list keeps two mark bits in the id word

// Blocks (= waves) of a persistent traversal launch: what stays resident on the chip.
template <class K>
static int resident_blocks(K kernel, size_t lds, int num_cu)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, lds) != hipSuccess || per_cu < 1) per_cu = 8;
    return per_cu * std::max(1, num_cu);
}

static int cand_lds_cap(int k, int dim, bool heur, int nbcap)
{
    int cap = std::min(std::max(4 * k, 256), 4096);
    if (const char *e = std::getenv("HNSW_MI355X_CAND_CAP")) cap = std::max(1, std::atoi(e)); // tests: force spill / hand-back
    while (cap > 64 && search_lds_bytes(k, cap, dim, heur, nbcap) > 64 * 1024) cap /= 2;
    return cap;
}
static int spill_cap_for_tests()
{
    if (const char *e = std::getenv("HNSW_MI355X_SPILL_CAP")) return std::max(0, std::min(kSpillCap, std::atoi(e)));
    return kSpillCap;
}

template <class T>
static bool grow_dev(T **p, size_t *cap, size_t need)
{
    if (need <= *cap) return true;
    if (*p) HIP_OK(hipFree(*p));
    *p = nullptr;
    HIP_OK(hipMalloc(p, sizeof(T) * need));
    *cap = need;
    return true;
}

// Row loads overlapped with the visited atomics in launches that do not fill the chip
// (HNSW_MI355X_OVERLAP=0 disables, =2 forces it for every launch: tests).
static int overlap_mode()
{
    const char *e = std::getenv("HNSW_MI355X_OVERLAP");
    return e ? std::atoi(e) : 1;
}

// The per-wave visited-id hash tables (VisitedSet): capacity a power of two, >= 16384 and >= 64 per
// beam entry (a traversal visits roughly 35 ids per beam entry), all entries -1 between jobs.
// HNSW_MI355X_VIS_HASH=1/0 forces / forbids them; HNSW_MI355X_VIS_HASH_CAP overrides the capacity (tests).
bool Device::visited_table(size_t vis_bytes_per_job, int k, int **out, int *out_cap)
{
    *out = nullptr;
    *out_cap = 0;
    const char *e = std::getenv("HNSW_MI355X_VIS_HASH");
    // measured: at 1M nodes (125-KB bitsets) the bitset is faster (2.5 M vs 1.9 M queries/s on C2); at
    // 10M (1.25 MB) the table wins (1.48 M vs 1.28 M with the log-cleared bitset, 0.98 M streaming it)
    const bool want = e ? std::atoi(e) != 0 : vis_bytes_per_job > (512u << 10);
    if (!want) return true;
    int cap = 16384;
    while (cap < 64 * k && cap < (1 << 22)) cap <<= 1;
    if (const char *c = std::getenv("HNSW_MI355X_VIS_HASH_CAP")) { cap = 64; while (cap < std::atoi(c) && cap < (1 << 22)) cap <<= 1; }
    const size_t need = (size_t)max_slots() * (size_t)cap;
    if (need > s_vistab_cap_ || cap != s_vistab_each_) {
        HIP_OK(hipStreamSynchronize(S(stream_)));
        if (s_vistab_) HIP_OK(hipFree(s_vistab_));
        s_vistab_ = nullptr; s_vistab_cap_ = 0;
        HIP_OK(hipMalloc(&s_vistab_, sizeof(int) * need));
        HIP_OK(hipMemsetAsync(s_vistab_, 0xff, sizeof(int) * need, S(stream_)));
        s_vistab_cap_ = need;
        s_vistab_each_ = cap;
    }
    *out = s_vistab_;
    *out_cap = cap;
    return true;
}

// chunk: jobs per launch (job / result buffers); slots: waves of a persistent launch (visited
// bitsets, spill areas).  The visited arena is all zero between launches: zeroed when allocated,
// and every wave clears its bitset after each job.
bool Device::ensure_search_scratch(long long chunk, long long slots, int k, size_t vis_bytes_per_job)
{
    if (vis_bytes_per_job * (size_t)slots > s_visited_bytes_) {
        if (s_visited_) HIP_OK(hipFree(s_visited_));
        s_visited_ = nullptr;
        s_visited_bytes_ = vis_bytes_per_job * (size_t)slots;
        HIP_OK(hipMalloc(&s_visited_, s_visited_bytes_));
        HIP_OK(hipMemsetAsync(s_visited_, 0, s_visited_bytes_, S(stream_)));
    }
    if (!s_jobctr_) HIP_OK(hipMalloc(&s_jobctr_, sizeof(int)));
    if ((size_t)chunk > s_jobs_cap_) {
        if (s_jobs_) HIP_OK(hipFree(s_jobs_));
        if (s_cnt_) HIP_OK(hipFree(s_cnt_));
        if (s_flag_) HIP_OK(hipFree(s_flag_));
        s_jobs_cap_ = (size_t)chunk;
        HIP_OK(hipMalloc(&s_jobs_, sizeof(SearchJob) * s_jobs_cap_));
        HIP_OK(hipMalloc(&s_cnt_, sizeof(int) * s_jobs_cap_));
        HIP_OK(hipMalloc(&s_flag_, sizeof(int) * s_jobs_cap_));
    }
    if (k > 0 && !grow_dev(&s_hits_, &s_hits_cap_, (size_t)chunk * k)) return false;
    if (!grow_dev(&s_spill_, &s_spill_cap_, (size_t)slots * kSpillCap + 8)) return false; // +8: get2 may read one entry past a heap
    if (!s_evals_) HIP_OK(hipMalloc(&s_evals_, sizeof(unsigned long long)));
    if (!ev0_) { hipEvent_t a, b; HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b)); ev0_ = a; ev1_ = b; }
    return true;
}

static bool jobs_valid(const SearchJob *jobs, int njobs, long long g_n, long long n_queries, long long n_rows)
{
    for (int i = 0; i < njobs; ++i) {
        const SearchJob &j = jobs[i];
        bool ok = j.entry >= 0 && j.entry < g_n && j.search_layer >= 0 && j.entry_layer >= j.search_layer &&
                  (j.qref >= 0 ? j.qref < n_queries : (~j.qref) < n_rows);
        if (!ok) return false;
    }
    return true;
}

bool Device::insert_search_batch(const SearchJob *jobs, int njobs, int k, int max_edges0, int n_upper, InsertResults *res)
{
    if (njobs <= 0) return true;
    if (!jobs || !res || k < 1 || n_upper < 0 || max_edges0 < 2) { set_dev_error("insert_search_batch: bad argument"); return false; }
    if (g_n_ <= 0) { set_dev_error("insert_search_batch: no graph uploaded"); return false; }
    for (int i = 0; i < njobs; ++i) {
        const SearchJob &j = jobs[i];
        if (j.qref >= 0) { set_dev_error("insert_search_batch: qref must name a stored row"); return false; }
        if (j.search_layer > 0 && (j.aux < 0 || j.aux + j.search_layer > n_upper)) { set_dev_error("insert_search_batch: upper-layer slot out of range"); return false; }
    }
    if (!jobs_valid(jobs, njobs, g_n_, n_queries_, n_rows_hw_)) { set_dev_error("insert_search_batch: job outside the uploaded graph / rows"); return false; }
    const int cand_cap = cand_lds_cap(k, dim_, true, nbcap());
    const size_t lds = search_lds_bytes(k, cand_cap, dim_, true, nbcap());
    if (lds > 64 * 1024) { set_dev_error("insert_search_batch: beam width / dimension exceed the LDS budget"); return false; }
    const int ns = g_n_ < kSortedTopMaxNodes ? sorted_top_sets(k) : 0;
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    const int sel_stride = max_edges0;
    const long long vis_words = ((g_n_ + 31) / 32 + 3) & ~3LL;
    const size_t vis_bytes_per_job = sizeof(unsigned) * (size_t)vis_words;
    const long long chunk = std::min<long long>(njobs, 1 << 20);
    if (!ensure_search_scratch(chunk, max_slots(), 0, vis_bytes_per_job)) return false;
    int *vis_tab = nullptr;
    int vis_tab_cap = 0;
    if (!visited_table(vis_bytes_per_job, k, &vis_tab, &vis_tab_cap)) return false;
    const size_t nU = (size_t)std::max(n_upper, 1);
    if (!grow_dev(&s_sel_, &s_sel_cap_, (size_t)njobs * sel_stride) || !grow_dev(&s_lcnt_, &s_lcnt_cap_, (size_t)njobs) ||
        !grow_dev(&s_selU_, &s_selU_cap_, nU * sel_stride) || !grow_dev(&s_cntU_, &s_cntU_cap_, nU) ||
        !grow_dev(&s_iflag_, &s_iflag_cap_, (size_t)njobs))
        return false;
    // pinned results: [sel0 | cnt0 | selU | cntU | flag | evals]
    const size_t b_sel0 = 4u * (size_t)njobs * sel_stride, b_cnt0 = 4u * (size_t)njobs, b_selU = 4u * nU * sel_stride, b_cntU = 4u * nU, b_flag = 4u * (size_t)njobs;
    const size_t need = b_sel0 + b_cnt0 + b_selU + b_cntU + b_flag + 16;
    if (need > h_res_cap_) {
        if (h_res_) (void)hipHostFree(h_res_);
        h_res_ = nullptr; h_res_cap_ = 0;
        if (hipHostMalloc(&h_res_, need + need / 2, hipHostMallocDefault) != hipSuccess) { set_dev_error("insert_search_batch: pinned allocation failed"); return false; }
        h_res_cap_ = need + need / 2;
    }
    char *hb = static_cast<char *>(h_res_);
    int *h_sel0 = reinterpret_cast<int *>(hb), *h_cnt0 = reinterpret_cast<int *>(hb + b_sel0);
    int *h_selU = reinterpret_cast<int *>(hb + b_sel0 + b_cnt0), *h_cntU = reinterpret_cast<int *>(hb + b_sel0 + b_cnt0 + b_selU);
    int *h_flag = reinterpret_cast<int *>(hb + b_sel0 + b_cnt0 + b_selU + b_cntU);
    unsigned long long *h_ev = reinterpret_cast<unsigned long long *>(hb + ((b_sel0 + b_cnt0 + b_selU + b_cntU + b_flag + 7) & ~(size_t)7));
    SearchJob *h_jobs = static_cast<SearchJob *>(pinned_stage(sizeof(SearchJob) * (size_t)chunk));
    if (!h_jobs) return false;
    for (long long off = 0; off < njobs; off += chunk) {
        const int nj = (int)std::min<long long>(chunk, njobs - off);
        memcpy(h_jobs, jobs + off, sizeof(SearchJob) * (size_t)nj);
        HIP_OK(hipMemcpyAsync(s_jobs_, h_jobs, sizeof(SearchJob) * (size_t)nj, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemsetAsync(s_jobctr_, 0, sizeof(int), st));
        HIP_OK(hipMemsetAsync(s_evals_, 0, sizeof(unsigned long long), st));
        const bool timed = profiling_;
        if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev0_, st));
#define LAUNCH2(M, NS_, H_, GRID, LDS, CAP) \
    do { \
        const int slots_ = std::min(max_slots(), resident_blocks(graph_insert_search_kernel<M, NS_, H_>, LDS, num_cu_)); \
        hipLaunchKernelGGL((graph_insert_search_kernel<M, NS_, H_>), dim3(std::min<int>(GRID, slots_)), \
                       dim3(64), LDS, st, d_rows_, d_row_sn_, dim_, g_adj0_, g_stride0_, \
                       g_upper_, g_pool_, g_strideU_, s_jobs_, k, CAP, reinterpret_cast<ND *>(s_spill_), spill_cap_for_tests(),  \
                       max_edges0, s_visited_, vis_words, vis_tab, vis_tab_cap, s_sel_ + (size_t)off * sel_stride, s_lcnt_ + off, s_selU_, s_cntU_,        \
                       sel_stride, s_iflag_ + off, s_evals_, nbcap(), GRID, s_jobctr_, (overlap_mode() == 2 || (overlap_mode() == 1 && GRID <= slots_)) ? 1 : 0); \
    } while (0)
#define LAUNCH3(NS_, H_, GRID, LDS, CAP)                                                                              \
    do {                                                                                                                   \
        if (metric_ == M_SQ) LAUNCH2(M_SQ, NS_, H_, GRID, LDS, CAP);                                                  \
        else if (metric_ == M_COS) LAUNCH2(M_COS, NS_, H_, GRID, LDS, CAP);                                           \
        else LAUNCH2(M_UCOS, NS_, H_, GRID, LDS, CAP);                                                                \
    } while (0)
#define LAUNCH(NS_, GRID, LDS, CAP)                                                                                   \
    do {                                                                                                                   \
        if (vis_tab) LAUNCH3(NS_, true, GRID, LDS, CAP);                                                              \
        else LAUNCH3(NS_, false, GRID, LDS, CAP);                                                                     \
    } while (0)
        switch (ns) {
        case 1: LAUNCH(1, nj, lds, cand_cap); break;
        case 2: LAUNCH(2, nj, lds, cand_cap); break;
        case 4: LAUNCH(4, nj, lds, cand_cap); break;
        case 8: LAUNCH(8, nj, lds, cand_cap); break;
        default: LAUNCH(0, nj, lds, cand_cap); break;
        }
        HIP_OK(hipGetLastError());
#undef LAUNCH
#undef LAUNCH3
#undef LAUNCH2
        if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev1_, st));
        HIP_OK(hipMemcpyAsync(h_ev, s_evals_, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st)); // the job staging buffer is reused by the next chunk
        stats_.search_launches++;
        stats_.search_evals += *h_ev;
        if (timed) {
            float ms = 0.f;
            HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)ev0_, (hipEvent_t)ev1_));
            stats_.search_kernel_ms += ms;
            stats_.search_timed_launches++;
            stats_.search_timed_evals += *h_ev;
        }
    }
    HIP_OK(hipMemcpyAsync(h_sel0, s_sel_, b_sel0, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(h_cnt0, s_lcnt_, b_cnt0, hipMemcpyDeviceToHost, st));
    if (n_upper > 0) {
        HIP_OK(hipMemcpyAsync(h_selU, s_selU_, 4u * (size_t)n_upper * sel_stride, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(h_cntU, s_cntU_, 4u * (size_t)n_upper, hipMemcpyDeviceToHost, st));
    }
    HIP_OK(hipMemcpyAsync(h_flag, s_iflag_, b_flag, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (int i = 0; i < njobs; ++i) {
        if (h_flag[i] == 2) { stats_.search_repeats++; h_flag[i] = 0; }
        stats_.search_overflows += (uint64_t)(h_flag[i] != 0);
    }
    last_insert_jobs_ = njobs <= chunk ? njobs : 0; // a single launch left everything in place
    last_insert_upper_ = n_upper;
    last_insert_stride_ = sel_stride;
    *res = InsertResults{h_sel0, h_cnt0, h_selU, h_cntU, h_flag, sel_stride};
    return true;
}

bool Device::traversal_fits(int k, bool with_heuristic, int max_edges) const
{
    if (k < 1 || 2 * max_edges + 1 > 128) return false;
    const int nb = std::max(8, (2 * max_edges + 1 + 7) & ~7);
    const int cap = cand_lds_cap(k, dim_, with_heuristic, nb);
    return search_lds_bytes(k, cap, dim_, with_heuristic, nb) <= 64 * 1024 && search_lds_bytes(nb, 0, dim_, true, nb) <= 64 * 1024;
}

bool Device::graph_append_nodes(long long first, long long n, const int *level, const int64_t *upper, const int *pool,
                                long long pool_from, long long pool_len, bool *need_full_sync)
{
    *need_full_sync = false;
    if (n <= 0) return true;
    if (!g_adj0_ || first != g_n_ || first + n > g_cap_n_ || pool_len > g_pool_cap_) { *need_full_sync = true; return true; }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    HIP_OK(hipMemcpyAsync(g_level_ + first, level + first, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(g_upper_ + first, upper + first, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
    // new nodes start with empty lists (GraphData.NewNode :224-242)
    HIP_OK(hipMemsetAsync(g_adj0_ + (size_t)first * g_stride0_, 0, sizeof(int) * (size_t)n * g_stride0_, st));
    HIP_OK(hipMemsetAsync(g_tested0_ + first, 0, sizeof(int) * (size_t)n, st));
    if (pool_len > pool_from) HIP_OK(hipMemsetAsync(g_testedU_ + pool_from / std::max(1, g_strideU_), 0, sizeof(int) * (size_t)((pool_len - pool_from) / std::max(1, g_strideU_) + 1), st));
    if (pool_len > pool_from) HIP_OK(hipMemcpyAsync(g_pool_ + pool_from, pool + pool_from, sizeof(int) * (size_t)(pool_len - pool_from), hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    g_n_ = first + n;
    return true;
}

bool Device::link_batch(const int *rows, int nrows, int row_stride, const int *g_node, const int *g_layer, const int *g_off,
                        const int *g_items, int ngroups, int max_edges0, int *out_lists, int list_stride)
{
    if (ngroups > 0 && !out_lists) { set_dev_error("link_batch: bad argument"); return false; }
    const int *res = nullptr;
    if (!link_batch_begin(0, rows, nrows, row_stride, g_node, g_layer, g_off, g_items, ngroups, max_edges0, list_stride)) return false;
    if (!link_batch_finish(0, &res)) return false;
    if (ngroups > 0) memcpy(out_lists, res, sizeof(int) * (size_t)ngroups * list_stride);
    return true;
}

bool Device::link_batch_begin(int set, const int *rows, int nrows, int row_stride, const int *g_node, const int *g_layer, const int *g_off,
                              const int *g_items, int ngroups, int max_edges0, int list_stride, bool want_lists)
{
    if (set < 0 || set > 1 || nrows < 0 || ngroups < 0 || (nrows > 0 && !rows) || (ngroups > 0 && (!g_node || !g_layer || !g_off || !g_items))) {
        set_dev_error("link_batch: bad argument");
        return false;
    }
    LinkSet &ls = lset_[set];
    if (ls.busy) { set_dev_error("link_batch: staging set still in flight"); return false; }
    if (g_n_ <= 0) { set_dev_error("link_batch: no graph uploaded"); return false; }
    // host-side validation: a bad id must be an error return, never a GPU fault
    for (int r = 0; r < nrows; ++r) {
        const int *x = rows + (size_t)r * row_stride;
        bool ok = x[0] >= 0 && x[0] < g_n_ && x[1] >= 0 && x[2] >= 0 && x[2] <= row_stride - 3 &&
                  x[2] <= (x[1] == 0 ? max_edges0 : max_edges0 / 2);
        for (int i = 0; ok && i < x[2]; ++i) ok = x[3 + i] >= 0 && x[3 + i] < g_n_;
        if (!ok) { set_dev_error("link_batch: row record outside the graph"); return false; }
    }
    const int total = ngroups > 0 ? g_off[ngroups] : 0;
    for (int g = 0; g < ngroups; ++g)
        if (g_node[g] < 0 || g_node[g] >= g_n_ || g_layer[g] < 0 || g_off[g + 1] < g_off[g]) { set_dev_error("link_batch: group outside the graph"); return false; }
    for (int t = 0; t < total; ++t)
        if (g_items[t] < 0 || g_items[t] >= g_n_) { set_dev_error("link_batch: item outside the graph"); return false; }
    if (list_stride < max_edges0 + 1 || max_edges0 + 1 > nbcap()) { set_dev_error("link_batch: list stride too small"); return false; }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    if (!ensure_search_scratch(1, 1, 0, 16)) return false;
    const size_t need[5] = {(size_t)nrows * row_stride, (size_t)ngroups * 2, (size_t)ngroups + 1, (size_t)std::max(total, 1), (size_t)ngroups * list_stride};
    // device buffers are shared by both sets: the stream orders one sub-batch after the other.
    // Growing one frees the old allocation, which must not be in use any more.
    for (int i = 0; i < 5; ++i) {
        if (std::max<size_t>(need[i], 1) > s_lk_cap_[i]) {
            HIP_OK(hipStreamSynchronize(st));
            if (!grow_dev(&s_lk_[i], &s_lk_cap_[i], std::max<size_t>(need[i], 1) * 2)) return false;
        }
    }
    // pinned staging of this set: [rows | node | layer | off | items], results, evaluation count
    const size_t in_ints = need[0] + need[1] + need[2] + need[3];
    if (in_ints > ls.in_cap) {
        if (ls.h_in) (void)hipHostFree(ls.h_in);
        ls.h_in = nullptr; ls.in_cap = 0;
        if (hipHostMalloc((void **)&ls.h_in, sizeof(int) * in_ints * 2, hipHostMallocDefault) != hipSuccess) { set_dev_error("link_batch: pinned allocation failed"); return false; }
        ls.in_cap = in_ints * 2;
    }
    if (std::max<size_t>(need[4], 1) > ls.out_cap) {
        if (ls.h_out) (void)hipHostFree(ls.h_out);
        ls.h_out = nullptr; ls.out_cap = 0;
        if (hipHostMalloc((void **)&ls.h_out, sizeof(int) * std::max<size_t>(need[4], 1) * 2, hipHostMallocDefault) != hipSuccess) { set_dev_error("link_batch: pinned allocation failed"); return false; }
        ls.out_cap = std::max<size_t>(need[4], 1) * 2;
    }
    if (!ls.h_ev && hipHostMalloc((void **)&ls.h_ev, 16, hipHostMallocDefault) != hipSuccess) { set_dev_error("link_batch: pinned allocation failed"); return false; }
    if (!ls.ev_done) {
        hipEvent_t a, b, c;
        HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b)); HIP_OK(hipEventCreate(&c));
        ls.ev_start = a; ls.ev_stop = b; ls.ev_done = c;
    }
    int *h_rows = ls.h_in, *h_node = h_rows + need[0], *h_off = h_node + need[1], *h_items = h_off + need[2];
    if (nrows > 0) memcpy(h_rows, rows, sizeof(int) * need[0]);
    if (ngroups > 0) {
        memcpy(h_node, g_node, sizeof(int) * (size_t)ngroups);
        memcpy(h_node + ngroups, g_layer, sizeof(int) * (size_t)ngroups);
        memcpy(h_off, g_off, sizeof(int) * ((size_t)ngroups + 1));
        memcpy(h_items, g_items, sizeof(int) * (size_t)total);
    }
    *ls.h_ev = 0;
    ls.ngroups = ngroups;
    ls.timed = false;
    if (nrows > 0) {
        HIP_OK(hipMemcpyAsync(s_lk_[0], h_rows, sizeof(int) * need[0], hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(graph_write_rows_kernel, dim3(nrows), dim3(64), 0, st, g_adj0_, g_stride0_, g_upper_, g_pool_, g_strideU_, s_lk_[0], row_stride,
                           g_tested0_, g_testedU_, max_edges0);
        HIP_OK(hipGetLastError());
    }
    if (ngroups > 0) {
        HIP_OK(hipMemcpyAsync(s_lk_[1], h_node, sizeof(int) * (size_t)ngroups * 2, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(s_lk_[2], h_off, sizeof(int) * ((size_t)ngroups + 1), hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(s_lk_[3], h_items, sizeof(int) * (size_t)total, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemsetAsync(s_evals_, 0, sizeof(unsigned long long), st));
        ls.timed = profiling_;
        if (ls.timed) HIP_OK(hipEventRecord((hipEvent_t)ls.ev_start, st));
        const int k_cap = nbcap();
        const size_t lds = ((search_lds_bytes(k_cap, 0, dim_, true, nbcap()) + 15) & ~(size_t)15) + 4u * (size_t)(kNewMax + 1) * nbcap();
#define LAUNCH(M)                                                                                                          \
    hipLaunchKernelGGL(graph_link_kernel<M>, dim3(ngroups), dim3(64), lds, st, d_rows_, d_row_sn_, dim_, g_adj0_, g_stride0_,  \
                       g_upper_, g_pool_, g_strideU_, s_lk_[1], s_lk_[1] + ngroups, s_lk_[2], (const int *)nullptr, s_lk_[3], max_edges0, k_cap, \
                       s_lk_[4], list_stride, s_evals_, nbcap(), g_tested0_, g_testedU_)
        if (metric_ == M_SQ) LAUNCH(M_SQ);
        else if (metric_ == M_COS) LAUNCH(M_COS);
        else LAUNCH(M_UCOS);
#undef LAUNCH
        HIP_OK(hipGetLastError());
        if (ls.timed) HIP_OK(hipEventRecord((hipEvent_t)ls.ev_stop, st));
        if (want_lists) HIP_OK(hipMemcpyAsync(ls.h_out, s_lk_[4], sizeof(int) * (size_t)ngroups * list_stride, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(ls.h_ev, s_evals_, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    }
    HIP_OK(hipEventRecord((hipEvent_t)ls.ev_done, st));
    ls.busy = true;
    return true;
}

bool Device::link_batch_planned(int njobs, int n_upper, int max_edges0)
{
    if (njobs <= 0) return true;
    if (njobs != last_insert_jobs_ || n_upper != last_insert_upper_ || max_edges0 != last_insert_stride_ || max_edges0 + 1 > nbcap()) {
        set_dev_error("link_batch_planned: no matching insert_search_batch results on the device");
        return false;
    }
    last_insert_jobs_ = 0;
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    if (!ensure_search_scratch(1, 1, 0, 16)) return false;
    // per-slot counters: all zero between batches (the link kernel resets what it used)
    const long long slots = g_cap_n_ + g_pool_cap_ / std::max(1, g_strideU_) + 2;
    if (slots != lp_slots_) {
        HIP_OK(hipStreamSynchronize(st));
        for (int i = 0; i < 3; ++i) {
            if (lp_slot_[i]) HIP_OK(hipFree(lp_slot_[i]));
            lp_slot_[i] = nullptr;
            HIP_OK(hipMalloc(&lp_slot_[i], sizeof(int) * (size_t)slots));
            HIP_OK(hipMemsetAsync(lp_slot_[i], 0, sizeof(int) * (size_t)slots, st));
        }
        lp_slots_ = slots;
    }
    const size_t max_appends = (size_t)(njobs + n_upper) * (size_t)max_edges0 + 1;
    for (int i = 0; i < 6; ++i) {
        if (max_appends > lp_grp_cap_[i]) {
            HIP_OK(hipStreamSynchronize(st));
            if (!grow_dev(&lp_grp_[i], &lp_grp_cap_[i], max_appends * 2)) return false;
        }
    }
    if (!lp_counters_) HIP_OK(hipMalloc(&lp_counters_, sizeof(int) * 4));
    HIP_OK(hipMemsetAsync(lp_counters_, 0, sizeof(int) * 4, st));
    HIP_OK(hipMemsetAsync(s_evals_, 0, sizeof(unsigned long long), st));
    size_t g_cap = lp_grp_cap_[0];
    for (int i = 1; i < 6; ++i) g_cap = std::min(g_cap, lp_grp_cap_[i]);
    LinkPlan P{lp_slot_[0], lp_slot_[1], lp_slot_[2], lp_grp_[0], lp_grp_[1], lp_grp_[2], lp_grp_[3], lp_grp_[4], lp_counters_, g_cap_n_,
               slots, g_n_, (int)std::min<size_t>(g_cap, 0x7fffffff), njobs};
    const bool timed = profiling_;
    if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev0_, st));
    hipLaunchKernelGGL(link_plan_kernel<true>, dim3(njobs), dim3(64), 0, st, s_jobs_, s_sel_, s_lcnt_, s_selU_, s_cntU_, last_insert_stride_,
                       g_adj0_, g_stride0_, g_upper_, g_pool_, g_strideU_, g_tested0_, g_testedU_, max_edges0, P);
    hipLaunchKernelGGL(link_offsets_kernel, dim3(512), dim3(256), 0, st, g_upper_, g_strideU_, P);
    hipLaunchKernelGGL(link_plan_kernel<false>, dim3(njobs), dim3(64), 0, st, s_jobs_, s_sel_, s_lcnt_, s_selU_, s_cntU_, last_insert_stride_,
                       g_adj0_, g_stride0_, g_upper_, g_pool_, g_strideU_, g_tested0_, g_testedU_, max_edges0, P);
    HIP_OK(hipGetLastError());
    // the number of groups comes back to size the last two launches (16 bytes, one short wait)
    unsigned long long *h_ev = static_cast<unsigned long long *>(pinned_stage(32));
    if (!h_ev) return false;
    int *h_ctr = reinterpret_cast<int *>(h_ev + 1);
    HIP_OK(hipMemcpyAsync(h_ctr, lp_counters_, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    const int G = h_ctr[0];
    if (h_ctr[3] != 0 || G < 0 || (size_t)G > g_cap) {
        set_dev_error("link_batch_planned: inconsistent selection data on the device (guard " + std::to_string(h_ctr[3]) + ")");
        return false;
    }
    if (G > 0) {
        hipLaunchKernelGGL(link_order_kernel, dim3(G), dim3(64), 0, st, s_jobs_, g_upper_, g_strideU_, lp_grp_[5], P);
        HIP_OK(hipGetLastError());
            const int k_cap = nbcap();
        const size_t lds = ((search_lds_bytes(k_cap, 0, dim_, true, nbcap()) + 15) & ~(size_t)15) + 4u * (size_t)(kNewMax + 1) * nbcap();
#define LAUNCH(M)                                                                                                          \
    hipLaunchKernelGGL(graph_link_kernel<M>, dim3(G), dim3(64), lds, st, d_rows_, d_row_sn_, dim_, g_adj0_, g_stride0_,       \
                       g_upper_, g_pool_, g_strideU_, lp_grp_[0], lp_grp_[1], lp_grp_[2], lp_grp_[3], lp_grp_[5], max_edges0, k_cap, \
                       (int *)nullptr, 0, s_evals_, nbcap(), g_tested0_, g_testedU_)
        if (metric_ == M_SQ) LAUNCH(M_SQ);
        else if (metric_ == M_COS) LAUNCH(M_COS);
        else LAUNCH(M_UCOS);
#undef LAUNCH
        HIP_OK(hipGetLastError());
    }
    if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev1_, st));
    HIP_OK(hipMemcpyAsync(h_ev, s_evals_, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(h_ctr, lp_counters_, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (h_ctr[3] != 0) {
        set_dev_error("link_batch_planned: inconsistent selection data on the device (guard " + std::to_string(h_ctr[3]) + ")");
        return false;
    }
    stats_.search_launches++;
    stats_.search_evals += *h_ev;
    if (timed) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)ev0_, (hipEvent_t)ev1_));
        stats_.search_kernel_ms += ms;
        stats_.search_timed_launches++;
        stats_.search_timed_evals += *h_ev;
    }
    return true;
}

bool Device::download_graph(int *adj0, long long n, int *pool, long long pool_len)
{
    if (n < 0 || n > g_n_ || pool_len < 0 || pool_len > g_pool_cap_ || (n > 0 && !adj0) || (pool_len > 0 && !pool)) { set_dev_error("download_graph: bad argument"); return false; }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    if (n > 0) HIP_OK(hipMemcpyAsync(adj0, g_adj0_, sizeof(int) * (size_t)n * g_stride0_, hipMemcpyDeviceToHost, st));
    if (pool_len > 0) HIP_OK(hipMemcpyAsync(pool, g_pool_, sizeof(int) * (size_t)pool_len, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return true;
}

bool Device::link_batch_finish(int set, const int **out_lists)
{
    if (set < 0 || set > 1 || !lset_[set].busy) { set_dev_error("link_batch_finish: nothing in flight"); return false; }
    LinkSet &ls = lset_[set];
    if (!bind()) return false;
    HIP_OK(hipEventSynchronize((hipEvent_t)ls.ev_done));
    ls.busy = false;
    if (out_lists) *out_lists = ls.h_out;
    if (ls.ngroups > 0) {
        stats_.search_launches++;
        stats_.search_evals += *ls.h_ev;
        if (ls.timed) {
            float ms = 0.f;
            HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)ls.ev_start, (hipEvent_t)ls.ev_stop));
            stats_.search_kernel_ms += ms;
            stats_.search_timed_launches++;
            stats_.search_timed_evals += *ls.h_ev;
        }
    }
    return true;
}


// Pinned host staging (grown on demand): DMA to/from pageable user memory runs at ~2 GB/s,
// through a pinned bounce buffer at PCIe rate.
void *Device::pinned_stage(size_t bytes)
{
    if (bytes <= h_stage_cap_) return h_stage_;
    if (h_stage_) (void)hipHostFree(h_stage_);
    h_stage_ = nullptr;
    h_stage_cap_ = 0;
    size_t cap = std::max<size_t>(bytes, 1u << 20);
    if (hipHostMalloc(&h_stage_, cap, hipHostMallocDefault) != hipSuccess) { set_dev_error("pinned staging allocation failed"); return nullptr; }
    h_stage_cap_ = cap;
    return h_stage_;
}

// KnnQuery on the device: descent + layer-0 beam search (width k) + the stable top-k_out tail.
// out_ids / out_d: njobs x k_out, final (padded with -1 / NaN); out_flag: 1 = not run to
// completion (candidate heap beyond LDS + spill), caller re-runs that job on the lock-step path.
bool Device::search_batch(const SearchJob *jobs, int njobs, int k, int k_out, int *out_ids, float *out_d, int *out_flag)
{
    if (njobs <= 0) return true;
    if (!jobs || !out_ids || !out_d || !out_flag || k < 1 || k_out < 1) { set_dev_error("search_batch: bad argument"); return false; }
    if (g_n_ <= 0) { set_dev_error("search_batch: no graph uploaded"); return false; }
    if (!jobs_valid(jobs, njobs, g_n_, n_queries_, n_rows_hw_)) { set_dev_error("search_batch: job outside the uploaded graph / rows / queries"); return false; }
    const int cand_cap = cand_lds_cap(k, dim_, false, nbcap());
    const size_t lds = search_lds_bytes(k, cand_cap, dim_, false, nbcap());
    if (lds > 64 * 1024) { set_dev_error("search_batch: beam width / dimension exceed the LDS budget"); return false; }
    const int ns = g_n_ < kSortedTopMaxNodes ? sorted_top_sets(k) : 0;
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    const long long vis_words = ((g_n_ + 31) / 32 + 3) & ~3LL;
    const size_t vis_bytes_per_job = sizeof(unsigned) * (size_t)vis_words;
    const long long chunk = std::min<long long>(njobs, 1 << 20);
    if (!ensure_search_scratch(chunk, max_slots(), k_out, vis_bytes_per_job)) return false;
    int *vis_tab = nullptr;
    int vis_tab_cap = 0;
    if (!visited_table(vis_bytes_per_job, k, &vis_tab, &vis_tab_cap)) return false;
    // pinned layout: [evals (16 B) | jobs | ids | dists | flags]
    const size_t b_jobs = sizeof(SearchJob) * (size_t)chunk, b_res = 4u * (size_t)chunk * k_out;
    char *hs = static_cast<char *>(pinned_stage(16 + b_jobs + 2 * b_res + 4u * (size_t)chunk));
    if (!hs) return false;
    unsigned long long *h_ev = reinterpret_cast<unsigned long long *>(hs);
    SearchJob *h_jobs = reinterpret_cast<SearchJob *>(hs + 16);
    int *h_ids = reinterpret_cast<int *>(hs + 16 + b_jobs);
    float *h_d = reinterpret_cast<float *>(hs + 16 + b_jobs + b_res);
    int *h_flag = reinterpret_cast<int *>(hs + 16 + b_jobs + 2 * b_res);
    int *d_ids = reinterpret_cast<int *>(s_hits_);
    float *d_d = reinterpret_cast<float *>(s_hits_) + (size_t)chunk * k_out;
    for (long long off = 0; off < njobs; off += chunk) {
        const int nj = (int)std::min<long long>(chunk, njobs - off);
        memcpy(h_jobs, jobs + off, sizeof(SearchJob) * (size_t)nj);
        HIP_OK(hipMemcpyAsync(s_jobs_, h_jobs, sizeof(SearchJob) * (size_t)nj, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemsetAsync(s_jobctr_, 0, sizeof(int), st));
        HIP_OK(hipMemsetAsync(s_evals_, 0, sizeof(unsigned long long), st));
        const bool timed = profiling_;
        if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev0_, st));
#define LAUNCH2(M, NS_, H_, GRID, LDS, CAP) \
    do { \
        const int slots_ = std::min(max_slots(), resident_blocks(graph_search_kernel<M, NS_, H_>, LDS, num_cu_)); \
        hipLaunchKernelGGL((graph_search_kernel<M, NS_, H_>), dim3(std::min<int>(GRID, slots_)), \
                       dim3(64), LDS, st, d_rows_, d_row_sn_, d_queries_, d_q_sn_, dim_, \
                       g_adj0_, g_stride0_, g_upper_, g_pool_, g_strideU_, s_jobs_, k, CAP, reinterpret_cast<ND *>(s_spill_), \
                       spill_cap_for_tests(), s_visited_, vis_words, vis_tab, vis_tab_cap, k_out, d_ids, d_d, s_cnt_, s_flag_, s_evals_, nbcap(), GRID, s_jobctr_, (overlap_mode() == 2 || (overlap_mode() == 1 && GRID <= slots_)) ? 1 : 0); \
    } while (0)
#define LAUNCH3(NS_, H_, GRID, LDS, CAP)                                                                              \
    do {                                                                                                                   \
        if (metric_ == M_SQ) LAUNCH2(M_SQ, NS_, H_, GRID, LDS, CAP);                                                  \
        else if (metric_ == M_COS) LAUNCH2(M_COS, NS_, H_, GRID, LDS, CAP);                                           \
        else LAUNCH2(M_UCOS, NS_, H_, GRID, LDS, CAP);                                                                \
    } while (0)
#define LAUNCH(NS_, GRID, LDS, CAP)                                                                                   \
    do {                                                                                                                   \
        if (vis_tab) LAUNCH3(NS_, true, GRID, LDS, CAP);                                                              \
        else LAUNCH3(NS_, false, GRID, LDS, CAP);                                                                     \
    } while (0)
        switch (ns) {
        case 1: LAUNCH(1, nj, lds, cand_cap); break;
        case 2: LAUNCH(2, nj, lds, cand_cap); break;
        case 4: LAUNCH(4, nj, lds, cand_cap); break;
        case 8: LAUNCH(8, nj, lds, cand_cap); break;
        default: LAUNCH(0, nj, lds, cand_cap); break;
        }
        HIP_OK(hipGetLastError());
#undef LAUNCH
#undef LAUNCH3
#undef LAUNCH2
        if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev1_, st));
        HIP_OK(hipMemcpyAsync(h_ids, d_ids, 4u * (size_t)nj * k_out, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(h_d, d_d, 4u * (size_t)nj * k_out, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(h_flag, s_flag_, sizeof(int) * (size_t)nj, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(h_ev, s_evals_, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        memcpy(out_ids + (size_t)off * k_out, h_ids, 4u * (size_t)nj * k_out);
        memcpy(out_d + (size_t)off * k_out, h_d, 4u * (size_t)nj * k_out);
        for (int i = 0; i < nj; ++i) {
            if (h_flag[i] == 2) { stats_.search_repeats++; h_flag[i] = 0; }
        }
        memcpy(out_flag + off, h_flag, sizeof(int) * (size_t)nj);
        const unsigned long long ev = *h_ev;
        stats_.search_launches++;
        stats_.search_evals += ev;
        if (timed) {
            float ms = 0.f;
            HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)ev0_, (hipEvent_t)ev1_));
            stats_.search_kernel_ms += ms;
            stats_.search_timed_launches++;
            stats_.search_timed_evals += ev;
        }
    }
    for (int i = 0; i < njobs; ++i) stats_.search_overflows += (uint64_t)(out_flag[i] != 0);
    return true;
}

// ---- C-ABI graph staging (layer by layer) -------------------------------------------------
bool Device::graph_begin(int n, int max_edges, const int *levels)
{
    if (n <= 0 || max_edges < 1 || !levels) { set_dev_error("graph_begin: bad argument"); return false; }
    delete hg_;
    hg_ = new HostGraphStage();
    HostGraphStage &g = *hg_;
    g.n = n; g.M = max_edges; g.stride0 = 2 * max_edges + 2; g.strideU = max_edges + 2;
    g.level.assign(levels, levels + n);
    g.adj0.assign((size_t)n * g.stride0, 0);
    g.upper.assign((size_t)n, -1);
    size_t pool_len = 0;
    for (int i = 0; i < n; ++i) {
        if (levels[i] < 0 || levels[i] > 200) { set_dev_error("graph_begin: level out of range"); return false; }
        g.top = std::max(g.top, levels[i]);
        if (levels[i] > 0) { g.upper[(size_t)i] = (int64_t)pool_len; pool_len += (size_t)levels[i] * g.strideU; }
    }
    g.pool.assign(pool_len, 0);
    return true;
}

bool Device::graph_set_layer(int layer, const int *counts, const int *edges, int stride)
{
    if (!hg_) { set_dev_error("graph_set_layer: call hnswdev_graph_begin first"); return false; }
    HostGraphStage &g = *hg_;
    if (layer < 0 || !counts || !edges || stride < 1) { set_dev_error("graph_set_layer: bad argument"); return false; }
    const int cap = layer == 0 ? 2 * g.M + 1 : g.M + 1;
    for (int i = 0; i < g.n; ++i) {
        if (g.level[(size_t)i] < layer) continue;
        const int c = counts[i];
        if (c < 0 || c > cap || c > stride) { set_dev_error("graph_set_layer: edge count exceeds MaxEdges(layer) + 1"); return false; }
        int *l = layer == 0 ? g.adj0.data() + (size_t)i * g.stride0 : g.pool.data() + g.upper[(size_t)i] + (size_t)(layer - 1) * g.strideU;
        l[0] = c;
        for (int j = 0; j < c; ++j) {
            const int e = edges[(size_t)i * stride + j];
            if (e < 0 || e >= g.n || g.level[(size_t)e] < layer) { set_dev_error("graph_set_layer: edge to a node outside the layer"); return false; }
            l[1 + j] = e;
        }
    }
    return true;
}

bool Device::graph_commit()
{
    if (!hg_) { set_dev_error("graph_commit: nothing staged"); return false; }
    HostGraphStage &g = *hg_;
    if (g.n > n_rows_hw_) { set_dev_error("graph_commit: graph has more nodes than uploaded rows"); return false; }
    return set_graph(g.adj0.data(), g.n, g.stride0, g.level.data(), g.upper.data(), g.pool.data(), (long long)g.pool.size(), g.strideU);
}

bool Device::knn_search(const float *queries, int nq, int entry_point, int k_beam, int k_out, int *out_ids, float *out_d, int *out_flag)
{
    if (nq <= 0) return true;
    if (!hg_ || g_n_ <= 0) { set_dev_error("knn_search: no graph committed"); return false; }
    if (entry_point < 0 || entry_point >= hg_->n || k_out < 1 || k_beam < k_out) { set_dev_error("knn_search: bad argument"); return false; }
    if (!set_queries(queries, nq)) return false;
    std::vector<SearchJob> jobs((size_t)nq);
    const int top = hg_->level[(size_t)entry_point];
    for (int i = 0; i < nq; ++i) jobs[(size_t)i] = SearchJob{i, entry_point, top, 0, -1};
    return search_batch(jobs.data(), nq, k_beam, k_out, out_ids, out_d, out_flag);
}

// ---- synchronous conveniences behind the C ABI ---------------------------------------
bool Device::dist_query_batch(const float *queries, int nq, const int *offsets, const int *ids, float *out)
{
    if (nq <= 0) return true;
    if (!queries || !offsets || !out) { set_dev_error("dist_query_batch: null argument"); return false; }
    if (offsets[0] != 0) { set_dev_error("dist_query_batch: cand_offsets[0] must be 0"); return false; }
    for (int i = 0; i < nq; ++i)
        if (offsets[i + 1] < offsets[i]) { set_dev_error("dist_query_batch: cand_offsets must be non-decreasing"); return false; }
    const int total = offsets[nq];
    if (total > 0 && !ids) { set_dev_error("dist_query_batch: null cand_ids"); return false; }
    for (int j = 0; j < total; ++j)
        if (ids[j] < 0 || ids[j] >= n_rows_hw_) { set_dev_error("dist_query_batch: candidate id outside uploaded rows"); return false; }
    if (!set_queries(queries, nq)) return false;
    const int stride = 64, NS = 4096;
    StepBuffers *sb = alloc_step(NS, stride);
    if (!sb) return false;
    bool ok = true;
    int qi = 0, pos = 0; // next (query, offset-within-query) to schedule
    while (ok && qi < nq) {
        int used = 0;
        uint64_t ev = 0;
        std::vector<std::pair<int, int>> where; // (global offset, count) per slot
        while (qi < nq && used < NS) {
            int m = offsets[qi + 1] - offsets[qi] - pos;
            if (m <= 0) { ++qi; pos = 0; continue; }
            int take = std::min(m, stride);
            int g = offsets[qi] + pos;
            int *r = sb->rec + (size_t)used * sb->rec_stride;
            r[0] = take;
            r[1] = qi;
            memcpy(r + 2, ids + g, sizeof(int) * (size_t)take);
            where.emplace_back(g, take);
            ev += (uint64_t)take;
            ++used;
            pos += take;
        }
        if (used == 0) break;
        ok = launch_step(sb, used, ev) && wait_step(sb);
        if (ok)
            for (int s = 0; s < used; ++s)
                memcpy(out + where[s].first, sb->dist + (size_t)s * stride, sizeof(float) * (size_t)where[s].second);
    }
    free_step(sb);
    return ok;
}

bool Device::dist_pair_batch(const int *a, const int *b, int n, float *out)
{
    if (n <= 0) return true;
    if (!a || !b || !out) { set_dev_error("dist_pair_batch: null argument"); return false; }
    for (int j = 0; j < n; ++j)
        if (a[j] < 0 || a[j] >= n_rows_hw_ || b[j] < 0 || b[j] >= n_rows_hw_) {
            set_dev_error("dist_pair_batch: id outside uploaded rows");
            return false;
        }
    if (!bind()) return false;
    int *da = nullptr, *db = nullptr;
    float *dout = nullptr;
    hipStream_t st = S(stream_);
    HIP_OK(hipMalloc(&da, sizeof(int) * (size_t)n));
    HIP_OK(hipMalloc(&db, sizeof(int) * (size_t)n));
    HIP_OK(hipMalloc(&dout, sizeof(float) * (size_t)n));
    HIP_OK(hipMemcpyAsync(da, a, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(db, b, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
    dim3 grid((unsigned)(((long long)n * 8 + 255) / 256)), block(256);
#define LAUNCH(M) hipLaunchKernelGGL(pair_distance_kernel<M>, grid, block, 0, st, d_rows_, d_row_sn_, dim_, da, db, dout, n)
    if (metric_ == M_SQ) LAUNCH(M_SQ);
    else if (metric_ == M_COS) LAUNCH(M_COS);
    else LAUNCH(M_UCOS);
#undef LAUNCH
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(out, dout, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    stats_.launches++;
    stats_.evals += (uint64_t)n;
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return true;
}

// test hook (exported through the C ABI as hnswdev_test_sqrt_rn)
bool device_sqrt_rn(int device, const double *in, double *out, int n)
{
    HIP_OK(hipSetDevice(device));
    double *di = nullptr, *dout = nullptr;
    HIP_OK(hipMalloc(&di, sizeof(double) * (size_t)n));
    HIP_OK(hipMalloc(&dout, sizeof(double) * (size_t)n));
    HIP_OK(hipMemcpy(di, in, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sqrt_rn_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, di, dout, n);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpy(out, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    (void)hipFree(di); (void)hipFree(dout);
    return true;
}

} // namespace hnsw

// ------------------------------------------------------------------------------------
// C ABI (B): hnswdev_*
// ------------------------------------------------------------------------------------
using hnsw::Device;

extern "C" {

#define DEV_API __attribute__((visibility("default")))

DEV_API int hnswdev_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { hnsw::set_dev_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); return -1; }
    return n;
}

DEV_API int hnswdev_create(int device, int dim, int metric, long long capacity, void **ctx)
{
    if (!ctx) { hnsw::set_dev_error("hnswdev_create: null ctx"); return -1; }
    *ctx = nullptr;
    Device *d = Device::create(device, dim, metric, capacity);
    if (!d) return -1;
    *ctx = d;
    return 0;
}
DEV_API int hnswdev_destroy(void *ctx)
{
    delete (Device *)ctx;
    return 0;
}
#define CTX_OR_FAIL()                                                     \
    Device *d = (Device *)ctx;                                            \
    if (!d) { hnsw::set_dev_error("null hnswdev context"); return -1; }

DEV_API int hnswdev_reserve(void *ctx, long long capacity) { CTX_OR_FAIL(); return d->reserve(capacity) ? 0 : -1; }
DEV_API int hnswdev_upload_rows(void *ctx, int first_id, int n, const float *rows) { CTX_OR_FAIL(); return d->upload_rows(first_id, n, rows) ? 0 : -1; }
DEV_API int hnswdev_download_rows(void *ctx, int first_id, int n, float *rows) { CTX_OR_FAIL(); return d->download_rows(first_id, n, rows) ? 0 : -1; }
DEV_API int hnswdev_dist_query_batch(void *ctx, const float *queries, int nq, const int *cand_offsets, const int *cand_ids, float *out)
{
    CTX_OR_FAIL();
    return d->dist_query_batch(queries, nq, cand_offsets, cand_ids, out) ? 0 : -1;
}
DEV_API int hnswdev_dist_pair_batch(void *ctx, const int *a_ids, const int *b_ids, int n, float *out)
{
    CTX_OR_FAIL();
    return d->dist_pair_batch(a_ids, b_ids, n, out) ? 0 : -1;
}
DEV_API int hnswdev_graph_begin(void *ctx, int n, int max_edges, const int *levels) { CTX_OR_FAIL(); return d->graph_begin(n, max_edges, levels) ? 0 : -1; }
DEV_API int hnswdev_graph_set_layer(void *ctx, int layer, const int *counts, const int *edges, int stride) { CTX_OR_FAIL(); return d->graph_set_layer(layer, counts, edges, stride) ? 0 : -1; }
DEV_API int hnswdev_graph_commit(void *ctx) { CTX_OR_FAIL(); return d->graph_commit() ? 0 : -1; }
DEV_API int hnswdev_knn_search(void *ctx, const float *queries, int nq, int entry_point, int k_beam, int k_out, int *out_ids, float *out_dists, int *out_flags)
{
    CTX_OR_FAIL();
    return d->knn_search(queries, nq, entry_point, k_beam, k_out, out_ids, out_dists, out_flags) ? 0 : -1;
}
DEV_API int hnswdev_sync(void *ctx) { CTX_OR_FAIL(); return d->sync() ? 0 : -1; }
DEV_API int hnswdev_set_profiling(void *ctx, int enabled) { CTX_OR_FAIL(); d->set_profiling(enabled != 0); return 0; }
DEV_API int hnswdev_get_stats(void *ctx, hnswdev_stats *out)
{
    CTX_OR_FAIL();
    if (!out) return -1;
    d->get_stats(out);
    return 0;
}
DEV_API int hnswdev_reset_stats(void *ctx) { CTX_OR_FAIL(); d->reset_stats(); return 0; }
DEV_API int hnswdev_last_error(char *buf, int buf_len)
{
    std::string s = hnsw::get_dev_error();
    if (buf && buf_len > 0) {
        int w = std::min<int>((int)s.size(), buf_len - 1);
        memcpy(buf, s.data(), (size_t)w);
        buf[w] = 0;
    }
    return (int)s.size();
}
// test hook: correctly rounded device double sqrt (cosine epilogue), checked against the host's
DEV_API int hnswdev_test_sqrt_rn(int device, const double *in, double *out, int n)
{
    return hnsw::device_sqrt_rn(device, in, out, n) ? 0 : -1;
}

} // extern "C"
