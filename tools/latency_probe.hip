// tools/latency_probe.hip -- what ONE wavefront pays for a dependent chain of random row gathers (development tool).
//
// A traversal in a launch that does not fill the chip (B = 1 Add, the exact window's rounds, single-query calls) is a
// chain of expansions on one wave: 32 random 512-B rows, reduce, decide, next.  This measures the memory part of such
// a step in shader clocks, by table size (L2 / Infinity Cache / HBM / beyond the TLB's reach) and lane mapping:
//   0: 8 lanes x 64 strided dword loads (measure_pass<.., 4>)     1: 2 lanes x 16 dwordx4 loads (measure_pass2)
//   2: ONE dword per lane from 64 random rows (pure round trip)    3: 32 rows, one dwordx4 per lane pair (first touch only)
//
//   hipcc --offload-arch=gfx950 -O3 tools/latency_probe.hip -o tools/latency_probe && tools/latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V>
__global__ void __launch_bounds__(64) probe(const float *__restrict__ table, const int *__restrict__ ids, int iters, long long *out, float *sink)
{
    const int lane = threadIdx.x, grp = lane >> 3, j = lane & 7, r2 = lane >> 1, h = lane & 1;
    const int *my = ids + (size_t)blockIdx.x * iters * 64;
    float acc = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int dep = (int)acc; // 0 (the table is all zero), unknown to the compiler: the next ids wait for this step's data
        if (V == 0) {
            float v[4][16];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float *row = table + (size_t)(my[it * 64 + grp + 8 * p] + dep) * 128;
#pragma unroll
                for (int k = 0; k < 16; ++k) v[p][k] = row[8 * k + j];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc += v[p][k];
        } else if (V == 1) {
            const float4 *row = reinterpret_cast<const float4 *>(table + (size_t)(my[it * 64 + r2] + dep) * 128) + h;
            float4 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = row[2 * k];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
        } else if (V == 2) {
            acc += table[(size_t)(my[it * 64 + lane] + dep) * 128];
        } else {
            const float4 *row = reinterpret_cast<const float4 *>(table + (size_t)(my[it * 64 + r2] + dep) * 128) + h;
            const float4 v = row[0];
            acc += v.x + v.y + v.z + v.w;
        }
        acc += __shfl_xor(acc, 1, 64) * 0.0f; // a cross-lane step, as the collapse of a distance has
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const int iters = 2000;
    const int max_blocks = 4096;
    float *sink; long long *out; int *ids;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&out, sizeof(long long) * max_blocks));
    CK(hipMalloc(&ids, sizeof(int) * (size_t)max_blocks * iters * 64));
    const char *names[4] = {"8 lanes x 64 dword loads, 32 rows", "2 lanes x 16 dwordx4 loads, 32 rows", "one dword per lane, 64 rows", "one dwordx4 per lane pair, 32 rows (first touch)"};
    for (double mib : {1.0, 64.0, 512.0, 5120.0}) {
        const long long rows = (long long)(mib * 1048576.0 / 512.0);
        float *table;
        CK(hipMalloc(&table, (size_t)rows * 512));
        CK(hipMemset(table, 0, (size_t)rows * 512));
        std::vector<int> h((size_t)max_blocks * iters * 64);
        unsigned long long s = 88172645463325252ull;
        for (size_t i = 0; i < h.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % (unsigned long long)rows); }
        CK(hipMemcpy(ids, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
        for (int blocks : {1, 64, 1024, 3072}) {
            for (int v = 0; v < 4; ++v) {
                std::vector<long long> t((size_t)blocks);
                for (int rep = 0; rep < 2; ++rep) {
                    if (v == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(64), 0, 0, table, ids, iters, out, sink);
                    else if (v == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(64), 0, 0, table, ids, iters, out, sink);
                    else if (v == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(64), 0, 0, table, ids, iters, out, sink);
                    else hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(64), 0, 0, table, ids, iters, out, sink);
                    CK(hipDeviceSynchronize());
                }
                CK(hipMemcpy(t.data(), out, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
                double sum = 0; for (long long x : t) sum += (double)x;
                printf("table %6.0f MiB  waves %4d  [%s]: %.0f clocks per step\n", mib, blocks, names[v], sum / blocks / iters);
            }
        }
        CK(hipFree(table));
        fflush(stdout);
    }
    return 0;
}
