// tools/pool_probe.hip -- what ONE wave pays for the pool's operations (development tool): clocks per replace-the-farthest
// insertion and per closest-open lookup, alone in a kernel (no spills, nothing else alive).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ihnswindex.net_amd/csrc -Iinclude tools/pool_probe.hip -o tools/pool_probe
#include "device_kernels.h"
#include <cstdio>
#include <vector>
using namespace hnsw;
template <int NS>
__global__ void __launch_bounds__(64) probe(const unsigned *keys, int iters, long long *out, unsigned *sink)
{
    const int lane = threadIdx.x;
    PoolTop<NS> T;
    T.init();
    for (int t = 0; t < NS; ++t) { T.key[t] = keys[lane + 64 * t] | 0x80000000u; T.okey[t] = T.key[t]; T.id[t] = lane + 64 * t; }
    unsigned far_key = T.max_key();
    unsigned x = keys[1000 + lane];
    long long t0 = __builtin_readcyclecounter();
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const unsigned dk = ((unsigned)__builtin_amdgcn_readfirstlane((int)x) >> 1) | 0x80000000u;
        if (dk < far_key) {
            int slot, twins;
            T.template locate<false>(far_key, slot, twins);
            if (twins == 1) { T.put(slot, dk, it); far_key = T.max_key(); }
            else { T.put(slot, dk, it); T.mark_key(far_key, 0x40000000); }
        }
        acc += far_key;
    }
    long long t1 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        unsigned mk; int slot, eid, ns;
        T.min_open(mk, slot, eid, ns);
        if (slot >= 0) T.mark_expanded(slot, eid);
        acc += mk;
        if (slot < 0) { // refill
            for (int t = 0; t < NS; ++t) T.okey[t] = T.key[t];
        }
    }
    long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { out[0] = t1 - t0; out[1] = t2 - t1; sink[0] = acc; }
    for (int t = 0; t < NS; ++t) sink[1 + lane + 64 * t] = T.key[t] + T.id[t];
}
int main()
{
    const int iters = 20000;
    std::vector<unsigned> h(2000);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = s >> 1; }
    unsigned *keys, *sink; long long *out;
    hipMalloc(&keys, h.size() * 4); hipMalloc(&sink, 4096 * 4); hipMalloc(&out, 16);
    hipMemcpy(keys, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, keys, iters, out, sink);
        hipDeviceSynchronize();
    }
    long long o[2];
    hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
    printf("NS=4: replace-the-farthest %.0f clocks per candidate offered, closest-open + mark %.0f clocks per pop\n", (double)o[0] / iters, (double)o[1] / iters);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, keys, iters, out, sink);
        hipDeviceSynchronize();
    }
    hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
    printf("NS=2: replace-the-farthest %.0f clocks per candidate offered, closest-open + mark %.0f clocks per pop\n", (double)o[0] / iters, (double)o[1] / iters);
    return 0;
}
