// dk_search_common.h -- device code, part of device_kernels.h: what the traversals share: LDS carve-up, graph view, visited set, phase clocks, read log, FindEntryAtLayer.
#pragma once
#include "dk_measure.h"

namespace hnsw {

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

} // namespace hnsw
