// tools/pool_probe.hip -- what ONE wave pays for the pool's operations (development tool): clocks per replace-the-farthest
// insertion and per closest-open lookup, alone in a kernel (no spills, nothing else alive): the shipped PoolTop (PlainPool),
// and the two forms tried on the way to it -- v_writelane under `if (slot >> 6 == t)` ladders with per-lane cached extremes
// (CachedPool: 460 / 613 clocks at NS = 4 against 484 / 641 without the cache) and the first straight-line form (FlatPool:
// 352 / 463), which the shipped one follows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ihnswindex.net_amd/csrc -Iinclude tools/pool_probe.hip -o tools/pool_probe
#include "device_kernels.h"
#include <cstdio>
#include <vector>
using namespace hnsw;

template <class POOL>
__global__ void __launch_bounds__(64) probe(const unsigned *keys, int iters, long long *out, unsigned *sink)
{
    const int lane = threadIdx.x;
    POOL T;
    T.init();
    for (int s = 0; s < 200; ++s) T.put(s, (unsigned)__builtin_amdgcn_readfirstlane((int)keys[s]) | 0x80000000u, s);
    unsigned far_key = T.max_key();
    unsigned acc = 0;
    // (1) every candidate offered is accepted: just below the farthest key
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned dk = far_key - 1u - (unsigned)(it & 7);
        int twins;
        T.replace_far(far_key, dk, it, twins);
        far_key = T.max_key();
        acc += far_key + twins;
    }
    long long t1 = __builtin_readcyclecounter();
    // (2) pop the closest open entry, mark it, look for the next one (what an expansion does besides insertions)
    for (int it = 0; it < iters; ++it) {
        unsigned mk; int slot, eid;
        T.pop(mk, slot, eid);
        if (slot < 0) { T.reopen(); continue; }
        acc += mk + (unsigned)eid;
    }
    long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { out[0] = t1 - t0; out[1] = t2 - t1; sink[0] = acc; }
    T.dump(sink + 1, lane);
}

template <int NS>
struct PlainPool : PoolTop<NS> { // the shipped operations under the probe's interface
    __device__ void replace_far(unsigned k0, unsigned dk, int did, int &twins) { unsigned long long hb[NS]; this->hits(k0, hb, twins); if (twins == 1) this->replace(hb, dk, did); }
    __device__ void pop(unsigned &mk, int &slot, int &eid) { PoolTop<NS>::min_open(mk, slot, eid); if (slot >= 0) this->mark_expanded(slot, eid); }
    __device__ void reopen() { for (int t = 0; t < NS; ++t) this->okey[t] = this->key[t] ? this->key[t] : 0xffffffffu; }
    __device__ void dump(unsigned *p, int lane) const { for (int t = 0; t < NS; ++t) p[lane + 64 * t] = this->key[t] + this->id[t]; }
};

// per-lane cached extremes: lmax = the largest of the lane's keys, lmin = the smallest of its open keys; a wave-wide extreme is
// then ONE reduction over one register, and the slot holding it one ballot away
template <int NS>
struct CachedPool {
    unsigned key[NS], okey[NS], lmax, lmin;
    int id[NS];
    __device__ __forceinline__ void init()
    {
        for (int t = 0; t < NS; ++t) { key[t] = 0u; okey[t] = 0xffffffffu; id[t] = (int)0x80000000; }
        lmax = 0u; lmin = 0xffffffffu;
    }
    __device__ __forceinline__ void extremes()
    {
        unsigned a = key[0], b = okey[0];
#pragma unroll
        for (int t = 1; t < NS; ++t) { a = max(a, key[t]); b = min(b, okey[t]); }
        lmax = a; lmin = b;
    }
    __device__ __forceinline__ void put(int slot, unsigned k0, int i0)
    {
        const int l = slot & 63;
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if ((slot >> 6) == t) {
                key[t] = (unsigned)lane_write((int)k0, l, (int)key[t]);
                okey[t] = (unsigned)lane_write((int)k0, l, (int)okey[t]);
                id[t] = lane_write(i0, l, id[t]);
            }
        extremes();
    }
    __device__ __forceinline__ unsigned max_key() const { return wave_max_u32(lmax); }
    __device__ __forceinline__ void locate_far(unsigned k0, int &slot, int &twins) const
    {
        const unsigned long long bm = __ballot(lmax == k0);
        twins = (int)__popcll(bm);
        const int l = (int)__builtin_ctzll(bm);
        int tsel = NS - 1;
#pragma unroll
        for (int t = NS - 2; t >= 0; --t) tsel = key[t] == k0 ? t : tsel;
        slot = 64 * __builtin_amdgcn_readlane(tsel, l) + l;
    }
    __device__ __forceinline__ void min_open(unsigned &mk, int &slot, int &eid) const
    {
        mk = wave_min_u32(lmin);
        slot = -1; eid = 0;
        if (mk == 0xffffffffu) return;
        const unsigned long long bm = __ballot(lmin == mk);
        const int l = (int)__builtin_ctzll(bm);
        int tsel = NS - 1;
#pragma unroll
        for (int t = NS - 2; t >= 0; --t) tsel = okey[t] == mk ? t : tsel;
        const int ts = __builtin_amdgcn_readlane(tsel, l);
        slot = 64 * ts + l;
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if (ts == t) eid = __builtin_amdgcn_readlane(id[t], l);
    }
    __device__ __forceinline__ void mark_expanded(int slot, int idword)
    {
        const int l = slot & 63;
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if ((slot >> 6) == t) {
                okey[t] = (unsigned)lane_write(-1, l, (int)okey[t]);
                id[t] = lane_write(idword | (int)0x80000000, l, id[t]);
            }
        unsigned b = okey[0];
#pragma unroll
        for (int t = 1; t < NS; ++t) b = min(b, okey[t]);
        lmin = b;
    }
    __device__ void replace_far(unsigned k0, unsigned dk, int did, int &twins) { int slot; locate_far(k0, slot, twins); put(slot, dk, did); }
    __device__ void pop(unsigned &mk, int &slot, int &eid) { min_open(mk, slot, eid); if (slot >= 0) mark_expanded(slot, eid); }
    __device__ void reopen() { for (int t = 0; t < NS; ++t) okey[t] = key[t] ? key[t] : 0xffffffffu; extremes(); }
    __device__ void dump(unsigned *p, int lane) const { for (int t = 0; t < NS; ++t) p[lane + 64 * t] = key[t] + id[t]; }
};

// branch-free form: no v_writelane through M0 inside `if (slot >> 6 == t)` ladders (whose merges the compiler pays for in
// register copies): the lane(s) holding the key replace it themselves under the compare's own mask
template <int NS>
struct FlatPool {
    unsigned key[NS], okey[NS];
    int id[NS];
    __device__ __forceinline__ void init()
    {
        for (int t = 0; t < NS; ++t) { key[t] = 0u; okey[t] = 0xffffffffu; id[t] = (int)0x80000000; }
    }
    __device__ __forceinline__ void put(int slot, unsigned k0, int i0)
    {
        const unsigned long long bit = 1ull << (slot & 63);
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const bool mine = __builtin_amdgcn_inverse_ballot_w64((slot >> 6) == t ? bit : 0ull);
            key[t] = mine ? k0 : key[t];
            okey[t] = mine ? k0 : okey[t];
            id[t] = mine ? i0 : id[t];
        }
    }
    __device__ __forceinline__ unsigned max_key() const
    {
        unsigned v = key[0];
#pragma unroll
        for (int t = 1; t < NS; ++t) v = max(v, key[t]);
        return wave_max_u32(v);
    }
    __device__ __forceinline__ void replace_far(unsigned k0, unsigned dk, int did, int &twins)
    {
        bool hit[NS];
        int c = 0;
#pragma unroll
        for (int t = 0; t < NS; ++t) { hit[t] = key[t] == k0; c += (int)__popcll(__ballot(hit[t])); }
        twins = c;
        if (c == 1) {
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                key[t] = hit[t] ? dk : key[t];
                okey[t] = hit[t] ? dk : okey[t];
                id[t] = hit[t] ? did : id[t];
            }
        }
    }
    __device__ __forceinline__ void pop(unsigned &mk, int &slot, int &eid)
    {
        unsigned v = okey[0];
#pragma unroll
        for (int t = 1; t < NS; ++t) v = min(v, okey[t]);
        mk = wave_min_u32(v);
        slot = -1; eid = 0;
        if (mk == 0xffffffffu) return;
        unsigned long long h = 0ull;
        int ts = 0, sel = id[NS - 1];
#pragma unroll
        for (int t = NS - 1; t >= 0; --t) {
            const bool e = okey[t] == mk;
            const unsigned long long b = __ballot(e);
            if (b) { h = b; ts = t; }
            sel = e ? id[t] : sel;
        }
        const int l = (int)__builtin_ctzll(h);
        slot = 64 * ts + l;
        eid = __builtin_amdgcn_readlane(sel, l);
        const unsigned long long bit = 1ull << l;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const bool mine = __builtin_amdgcn_inverse_ballot_w64(ts == t ? bit : 0ull);
            okey[t] = mine ? 0xffffffffu : okey[t];
            id[t] = mine ? (id[t] | (int)0x80000000) : id[t];
        }
    }
    __device__ void reopen() { for (int t = 0; t < NS; ++t) okey[t] = key[t] ? key[t] : 0xffffffffu; }
    __device__ void dump(unsigned *p, int lane) const { for (int t = 0; t < NS; ++t) p[lane + 64 * t] = key[t] + id[t]; }
};

int main()
{
    const int iters = 20000;
    std::vector<unsigned> h(2000);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = s >> 1; }
    unsigned *keys, *sink; long long *out;
    (void)hipMalloc(&keys, h.size() * 4); (void)hipMalloc(&sink, 4096 * 4); (void)hipMalloc(&out, 16);
    (void)hipMemcpy(keys, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    long long o[2];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe<PlainPool<4>>, dim3(1), dim3(64), 0, 0, keys, iters, out, sink); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
    printf("PoolTop<4>:  replace-the-farthest %.0f clocks per accepted candidate, closest-open + mark %.0f clocks per pop\n", (double)o[0] / iters, (double)o[1] / iters);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe<PlainPool<2>>, dim3(1), dim3(64), 0, 0, keys, iters, out, sink); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
    printf("PoolTop<2>:  replace-the-farthest %.0f clocks per accepted candidate, closest-open + mark %.0f clocks per pop\n", (double)o[0] / iters, (double)o[1] / iters);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe<CachedPool<4>>, dim3(1), dim3(64), 0, 0, keys, iters, out, sink); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
    printf("CachedPool<4>: replace-the-farthest %.0f clocks per accepted candidate, closest-open + mark %.0f clocks per pop\n", (double)o[0] / iters, (double)o[1] / iters);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe<CachedPool<2>>, dim3(1), dim3(64), 0, 0, keys, iters, out, sink); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
    printf("CachedPool<2>: replace-the-farthest %.0f clocks per accepted candidate, closest-open + mark %.0f clocks per pop\n", (double)o[0] / iters, (double)o[1] / iters);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe<FlatPool<4>>, dim3(1), dim3(64), 0, 0, keys, iters, out, sink); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
    printf("FlatPool<4>: replace-the-farthest %.0f clocks per accepted candidate, closest-open + mark %.0f clocks per pop\n", (double)o[0] / iters, (double)o[1] / iters);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe<FlatPool<2>>, dim3(1), dim3(64), 0, 0, keys, iters, out, sink); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
    printf("FlatPool<2>: replace-the-farthest %.0f clocks per accepted candidate, closest-open + mark %.0f clocks per pop\n", (double)o[0] / iters, (double)o[1] / iters);
    return 0;
}
