// dk_heaps.h -- device code, part of device_kernels.h: keys, heap entries, the two BinaryHeaps in LDS (exact sift rules), wave-parallel pop.
#pragma once
#include "dk_metric.h"

namespace hnsw {

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

} // namespace hnsw
