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
// evaluates 8 candidates at a time; the collapse tree is three cross-lane adds.  One wave
// serves one search slot (one expansion: <= 2M candidate rows against one query).  Wider
// per-lane loads and lane-ring variants were measured no faster (tools/kbench.hip): with the
// step inputs resident in HBM this mapping gathers random 512-B rows at 4.2-4.9 TB/s.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
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
    HIP_OK(hipMemcpyAsync(d_rows_ + (size_t)first_id * dim_, rows, (size_t)n * dim_ * sizeof(float), hipMemcpyHostToDevice, S(stream_)));
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
    HIP_OK(hipMemcpyAsync(d_queries_, queries, (size_t)nq * dim_ * sizeof(float), hipMemcpyHostToDevice, S(stream_)));
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
