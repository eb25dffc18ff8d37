// tools/kbench.hip -- microbenchmark of wave mappings for the gather-distance kernel
// (development tool, not part of the library).  All variants must produce identical bits.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/kbench.hip -o /tmp/kbench && /tmp/kbench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e = (x);                                                                    \
        if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); }        \
    } while (0)

constexpr int DIM = 128;

__device__ __forceinline__ float xadd(float v, int m) { return v + __shfl_xor(v, m, 64); }

// ---- V1: 8 lanes per candidate, strided dword loads, sequential passes of 8 (current) ----
__global__ void __launch_bounds__(256) v1(const float *rows, const float *queries, const int *cnt, const int *qidx,
                                          const int *ids, float *out, int stride, int nslots)
{
    int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nslots) return;
    int n = cnt[s];
    if (n <= 0) return;
    const float *q = queries + (size_t)qidx[s] * DIM;
    int grp = lane >> 3, j = lane & 7;
    for (int c0 = 0; c0 < n; c0 += 8) {
        int c = c0 + grp;
        bool act = c < n;
        int id = ids[(size_t)s * stride + (act ? c : c0)];
        const float *a = rows + (size_t)id * DIM;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < DIM / 8; ++k) {
            float d = a[8 * k + j] - q[8 * k + j];
            acc = __builtin_fmaf(d, d, acc);
        }
        float t = xadd(acc, 4); t = xadd(t, 1); t = xadd(t, 2);
        if (act && j == 0) out[(size_t)s * stride + c] = t;
    }
}

// ---- V2: 2 lanes per candidate, float4 loads: lane h owns partials 4h..4h+3; 32 candidates per pass ----
__global__ void __launch_bounds__(256) v2(const float *rows, const float *queries, const int *cnt, const int *qidx,
                                          const int *ids, float *out, int stride, int nslots)
{
    int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nslots) return;
    int n = cnt[s];
    if (n <= 0) return;
    const float4 *q4 = reinterpret_cast<const float4 *>(queries + (size_t)qidx[s] * DIM);
    int c_in = lane >> 1, h = lane & 1;
    for (int c0 = 0; c0 < n; c0 += 32) {
        int c = c0 + c_in;
        bool act = c < n;
        int id = ids[(size_t)s * stride + (act ? c : c0)];
        const float4 *a4 = reinterpret_cast<const float4 *>(rows + (size_t)id * DIM);
        float4 r[DIM / 8];
#pragma unroll
        for (int k = 0; k < DIM / 8; ++k) r[k] = a4[2 * k + h];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < DIM / 8; ++k) {
            float4 qq = q4[2 * k + h];
            float d0 = r[k].x - qq.x, d1 = r[k].y - qq.y, d2 = r[k].z - qq.z, d3 = r[k].w - qq.w;
            a0 = __builtin_fmaf(d0, d0, a0); a1 = __builtin_fmaf(d1, d1, a1);
            a2 = __builtin_fmaf(d2, d2, a2); a3 = __builtin_fmaf(d3, d3, a3);
        }
        // p_j + p_{j+4}: partner lane (h^1) holds the other half
        float t0 = xadd(a0, 1), t1 = xadd(a1, 1), t2 = xadd(a2, 1), t3 = xadd(a3, 1);
        float t = (t0 + t1) + (t2 + t3);
        if (act && h == 0) out[(size_t)s * stride + c] = t;
    }
}

// ---- V3: 4 waves cooperate on one slot (block = slot): V1 loads, but all 32 candidates in flight ----
__global__ void __launch_bounds__(256) v3(const float *rows, const float *queries, const int *cnt, const int *qidx,
                                          const int *ids, float *out, int stride, int nslots)
{
    int s = blockIdx.x;
    int n = cnt[s];
    if (n <= 0) return;
    const float *q = queries + (size_t)qidx[s] * DIM;
    int j = threadIdx.x & 7;
    for (int c0 = 0; c0 < n; c0 += 32) {
        int c = c0 + (threadIdx.x >> 3);
        bool act = c < n;
        int id = ids[(size_t)s * stride + (act ? c : c0)];
        const float *a = rows + (size_t)id * DIM;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < DIM / 8; ++k) {
            float d = a[8 * k + j] - q[8 * k + j];
            acc = __builtin_fmaf(d, d, acc);
        }
        float t = xadd(acc, 4); t = xadd(t, 1); t = xadd(t, 2);
        if (act && j == 0) out[(size_t)s * stride + c] = t;
    }
}

// ---- V4: 8 lanes per candidate, float4 loads (a full 128-B line per group per instruction),
//          the serial chain walks around the 4 lanes that hold consecutive positions ----
// lane (s4 = j>>1, h = j&1) of a group loads chunk (2*s4+h) of line t: elements 32t + 8*s4 + 4h + c
// -> partial 4h+c, chain position 4t + s4.  The accumulator token hops s4 = 0->1->2->3 per line.
__device__ __forceinline__ float from_lane(float v, int src) { return __shfl(v, src, 64); }
__global__ void __launch_bounds__(256) v4(const float *rows, const float *queries, const int *cnt, const int *qidx,
                                          const int *ids, float *out, int stride, int nslots)
{
    int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nslots) return;
    int n = cnt[s];
    if (n <= 0) return;
    const float4 *q4 = reinterpret_cast<const float4 *>(queries + (size_t)qidx[s] * DIM);
    int grp = lane >> 3, j = lane & 7, s4 = j >> 1;
    int prev = (lane & ~7) | (((s4 + 3) & 3) << 1) | (j & 1); // lane holding the previous chain position
    for (int c0 = 0; c0 < n; c0 += 8) {
        int c = c0 + grp;
        bool act = c < n;
        int id = ids[(size_t)s * stride + (act ? c : c0)];
        const float4 *a4 = reinterpret_cast<const float4 *>(rows + (size_t)id * DIM);
        float4 r[4], qq[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { r[t] = a4[8 * t + j]; qq[t] = q4[8 * t + j]; }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float d0 = r[t].x - qq[t].x, d1 = r[t].y - qq[t].y, d2 = r[t].z - qq[t].z, d3 = r[t].w - qq[t].w;
#pragma unroll
            for (int hop = 0; hop < 4; ++hop) {
                // every lane computes; only the lane whose s4 == hop holds the live token
                float n0 = __builtin_fmaf(d0, d0, a0), n1 = __builtin_fmaf(d1, d1, a1);
                float n2 = __builtin_fmaf(d2, d2, a2), n3 = __builtin_fmaf(d3, d3, a3);
                bool live = (s4 == hop);
                a0 = live ? n0 : a0; a1 = live ? n1 : a1; a2 = live ? n2 : a2; a3 = live ? n3 : a3;
                // pass the token to the next position's lane
                float p0 = from_lane(a0, prev), p1 = from_lane(a1, prev), p2 = from_lane(a2, prev), p3 = from_lane(a3, prev);
                bool recv = (s4 == ((hop + 1) & 3));
                a0 = recv ? p0 : a0; a1 = recv ? p1 : a1; a2 = recv ? p2 : a2; a3 = recv ? p3 : a3;
            }
        }
        // after the last hop the token sits in s4 == 0 lanes (h = 0,1): partials 0..3 and 4..7
        float t0 = xadd(a0, 1), t1 = xadd(a1, 1), t2 = xadd(a2, 1), t3 = xadd(a3, 1);
        float t = (t0 + t1) + (t2 + t3);
        if (act && j == 0) out[(size_t)s * stride + c] = t;
    }
}

// ---- V5: like V2 (2 lanes/candidate, one pass) but the query chunk is staged once per wave in LDS ----
__global__ void __launch_bounds__(256) v5(const float *rows, const float *queries, const int *cnt, const int *qidx,
                                          const int *ids, float *out, int stride, int nslots)
{
    __shared__ float4 qs[4][DIM / 4];
    int w = threadIdx.x >> 6, lane = threadIdx.x & 63, s = blockIdx.x * 4 + w;
    if (s >= nslots) return;
    int n = cnt[s];
    if (n <= 0) return;
    const float4 *q4 = reinterpret_cast<const float4 *>(queries + (size_t)qidx[s] * DIM);
    if (lane < DIM / 4) qs[w][lane] = q4[lane];
    int c_in = lane >> 1, h = lane & 1;
    for (int c0 = 0; c0 < n; c0 += 32) {
        int c = c0 + c_in;
        bool act = c < n;
        int id = ids[(size_t)s * stride + (act ? c : c0)];
        const float4 *a4 = reinterpret_cast<const float4 *>(rows + (size_t)id * DIM);
        float4 r[DIM / 8];
#pragma unroll
        for (int k = 0; k < DIM / 8; ++k) r[k] = a4[2 * k + h];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < DIM / 8; ++k) {
            float4 qq = qs[w][2 * k + h];
            float d0 = r[k].x - qq.x, d1 = r[k].y - qq.y, d2 = r[k].z - qq.z, d3 = r[k].w - qq.w;
            a0 = __builtin_fmaf(d0, d0, a0); a1 = __builtin_fmaf(d1, d1, a1);
            a2 = __builtin_fmaf(d2, d2, a2); a3 = __builtin_fmaf(d3, d3, a3);
        }
        float t0 = xadd(a0, 1), t1 = xadd(a1, 1), t2 = xadd(a2, 1), t3 = xadd(a3, 1);
        float t = (t0 + t1) + (t2 + t3);
        if (act && h == 0) out[(size_t)s * stride + c] = t;
    }
}

typedef void (*kern_t)(const float *, const float *, const int *, const int *, const int *, float *, int, int);

int main(int argc, char **argv)
{
    long long N = argc > 1 ? atoll(argv[1]) : 1000000;
    int nslots = argc > 2 ? atoi(argv[2]) : 2048;
    int avg = argc > 3 ? atoi(argv[3]) : 21;
    int hostmem = argc > 4 ? atoi(argv[4]) : 0; // 1: ids/cnt/out in mapped pinned host memory (zero-copy)
    const int stride = 40, nq = 10000;
    printf("N=%lld nslots=%d avg_cnt=%d zero_copy=%d\n", N, nslots, avg, hostmem);
    std::mt19937 rng(1);
    std::vector<float> hrows((size_t)N * DIM), hq((size_t)nq * DIM);
    for (auto &v : hrows) v = (float)(rng() >> 8) / (1 << 24);
    for (auto &v : hq) v = (float)(rng() >> 8) / (1 << 24);
    float *rows, *queries;
    CK(hipMalloc(&rows, hrows.size() * 4)); CK(hipMalloc(&queries, hq.size() * 4));
    CK(hipMemcpy(rows, hrows.data(), hrows.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(queries, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    const int NSETS = 16; // rotate id sets so that consecutive launches do not re-read the same rows
    std::vector<int> hcnt((size_t)NSETS * nslots), hqi((size_t)NSETS * nslots), hids((size_t)NSETS * nslots * stride);
    long long evals_per = 0;
    for (int z = 0; z < NSETS; ++z)
        for (int s = 0; s < nslots; ++s) {
            int c = std::min(32, std::max(1, avg + (int)(rng() % 11) - 5));
            hcnt[(size_t)z * nslots + s] = c;
            hqi[(size_t)z * nslots + s] = rng() % nq;
            for (int i = 0; i < stride; ++i) hids[((size_t)z * nslots + s) * stride + i] = (int)(rng() % N);
            if (z == 0) evals_per += c;
        }
    int *cnt, *qi, *ids; float *out;
    size_t b_cnt = hcnt.size() * 4, b_ids = hids.size() * 4;
    if (hostmem) {
        int *h; CK(hipHostMalloc(&h, b_cnt, hipHostMallocMapped)); memcpy(h, hcnt.data(), b_cnt); CK(hipHostGetDevicePointer((void **)&cnt, h, 0));
        CK(hipHostMalloc(&h, b_cnt, hipHostMallocMapped)); memcpy(h, hqi.data(), b_cnt); CK(hipHostGetDevicePointer((void **)&qi, h, 0));
        CK(hipHostMalloc(&h, b_ids, hipHostMallocMapped)); memcpy(h, hids.data(), b_ids); CK(hipHostGetDevicePointer((void **)&ids, h, 0));
        float *hf; CK(hipHostMalloc(&hf, b_ids, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void **)&out, hf, 0));
    } else {
        CK(hipMalloc(&cnt, b_cnt)); CK(hipMalloc(&qi, b_cnt)); CK(hipMalloc(&ids, b_ids)); CK(hipMalloc(&out, b_ids));
        CK(hipMemcpy(cnt, hcnt.data(), b_cnt, hipMemcpyHostToDevice)); CK(hipMemcpy(qi, hqi.data(), b_cnt, hipMemcpyHostToDevice));
        CK(hipMemcpy(ids, hids.data(), b_ids, hipMemcpyHostToDevice));
    }
    struct { const char *name; kern_t k; int slots_per_block; } V[] = {
        {"v1 8lanes dword seq-pass", v1, 4}, {"v2 2lanes float4 1-pass", v2, 4}, {"v3 block/slot dword", v3, 1},
        {"v4 8lanes float4 ring", v4, 4}, {"v5 v2 + LDS query", v5, 4}};
    std::vector<float> ref, got((size_t)nslots * stride);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &v : V) {
        CK(hipMemset(out, 0, (size_t)nslots * stride * 4));
        int grid = (nslots + v.slots_per_block - 1) / v.slots_per_block;
        hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, rows, queries, cnt, qi, ids, out, stride, nslots);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDefault));
        bool same = true;
        if (ref.empty()) ref = got;
        else for (int s = 0; s < nslots && same; ++s) for (int c = 0; c < hcnt[s]; ++c) if (memcmp(&ref[(size_t)s * stride + c], &got[(size_t)s * stride + c], 4)) { same = false; break; }
        // per-launch time, events around each launch (as the library does), rotating id sets
        const int REP = 200;
        double ms_sum = 0;
        for (int r = 0; r < REP; ++r) {
            int z = r % NSETS;
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, rows, queries, cnt + (size_t)z * nslots, qi + (size_t)z * nslots,
                               ids + (size_t)z * nslots * stride, out, stride, nslots);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 20) ms_sum += ms;
        }
        double us = 1e3 * ms_sum / (REP - 20);
        // back-to-back launches (no sync between): throughput view
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < REP; ++r) {
            int z = r % NSETS;
            hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, rows, queries, cnt + (size_t)z * nslots, qi + (size_t)z * nslots,
                               ids + (size_t)z * nslots * stride, out, stride, nslots);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float msb; CK(hipEventElapsedTime(&msb, e0, e1));
        double usb = 1e3 * msb / REP;
        double bytes = (double)evals_per * DIM * 4;
        printf("%-28s bits_equal=%d  single %.2f us (%.0f GB/s)   back-to-back %.2f us (%.0f GB/s)\n", v.name, (int)same, us,
               bytes / us / 1e3, usb, bytes / usb / 1e3);
    }
    return 0;
}
