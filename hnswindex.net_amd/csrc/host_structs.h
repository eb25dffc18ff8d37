// host_structs.h -- host-side types whose exact behaviour decides tie order and therefore
// neighbour ids.  They are NOT accelerated; they are restated so that the
// lock-step driver walks the graph exactly as the reference does.  Citations relative to
// /root/reference/.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace hnsw {

// src/HNSWIndex/NodeDistance.cs:5-14
struct NodeDist {
    int id;
    float dist;
};

// float.CompareTo: NaN below every number, NaN == NaN.
static inline int float_compare_to(float x, float y)
{
    if (x < y) return -1;
    if (x > y) return 1;
    if (x == y) return 0;
    if (std::isnan(x)) return std::isnan(y) ? 0 : -1;
    return 1;
}
// src/HNSWIndex/DistanceComparer.cs:9-14 ("farther first": larger distance compares greater)
struct FartherFirst {
    static inline int cmp(const NodeDist &x, const NodeDist &y)
    {
        if (x.dist < y.dist) return -1;
        if (x.dist > y.dist) return 1;
        return float_compare_to(x.dist, y.dist);
    }
};
// src/HNSWIndex/DistanceComparer.cs:20-25 ("closer first")
struct CloserFirst {
    static inline int cmp(const NodeDist &x, const NodeDist &y)
    {
        if (x.dist > y.dist) return -1;
        if (x.dist < y.dist) return 1;
        return float_compare_to(y.dist, x.dist);
    }
};

// src/HNSWIndex/BinaryHeap.cs:30-107.  Root = greatest under Cmp.  Sift rules are what
// make the heap's array order (ToArray, :41-44) and tie handling match the reference:
// SiftUp stops on cmp <= 0 (:98); SiftDown takes the right child only if left < right
// strictly (:77) and stops on cmp <= 0 (:79).
template <class Cmp>
struct BinaryHeap {
    std::vector<NodeDist> buf;
    int count = 0;

    void reset(int capacity)
    {
        if ((int)buf.size() < capacity) buf.resize((size_t)capacity);
        count = 0;
    }
    inline const NodeDist &peek() const { return buf[0]; }
    inline void push(NodeDist item)
    {
        if (count == (int)buf.size()) buf.resize(buf.empty() ? 16 : buf.size() * 2);
        int i = count++;
        NodeDist *b = buf.data();
        while (i > 0) {
            int p = (i - 1) >> 1;
            NodeDist parent = b[p];
            if (Cmp::cmp(item, parent) <= 0) break;
            b[i] = parent;
            i = p;
        }
        b[i] = item;
    }
    inline NodeDist pop()
    {
        NodeDist *b = buf.data();
        NodeDist result = b[0];
        int n = --count;
        NodeDist item = b[n];
        if (n != 0) {
            int i = 0, half = n >> 1;
            while (i < half) {
                int left = (i << 1) + 1, right = left + 1;
                int mc = (right < n && Cmp::cmp(b[left], b[right]) < 0) ? right : left;
                if (Cmp::cmp(b[mc], item) <= 0) break;
                b[i] = b[mc];
                i = mc;
            }
            b[i] = item;
        }
        return result;
    }
};

// System.Random(int seed): .NET's seeded-compat generator (Knuth subtractive).  BCL, not in
// /root/reference; used at src/HNSWIndex/GraphData.cs:42 and :216.  Restated from the public
// dotnet/runtime algorithm; known answers checked in tests/test_host_logic.py.
struct DotnetRandom {
    int sa[56];
    int inext, inextp;
    explicit DotnetRandom(int seed = 0) { init(seed); }
    void init(int seed)
    {
        int subtraction = (seed == INT32_MIN) ? INT32_MAX : std::abs(seed);
        int mj = 161803398 - subtraction;
        std::memset(sa, 0, sizeof sa);
        sa[55] = mj;
        int mk = 1, ii = 0;
        for (int i = 1; i < 55; i++) {
            if ((ii += 21) >= 55) ii -= 55;
            sa[ii] = mk;
            mk = mj - mk;
            if (mk < 0) mk += INT32_MAX;
            mj = sa[ii];
        }
        for (int k = 1; k < 5; k++)
            for (int i = 1; i < 56; i++) {
                int n = i + 30;
                if (n >= 55) n -= 55;
                sa[i] = (int)((uint32_t)sa[i] - (uint32_t)sa[1 + n]);
                if (sa[i] < 0) sa[i] += INT32_MAX;
            }
        inext = 0;
        inextp = 21;
    }
    int internal_sample()
    {
        int li = inext, lp = inextp;
        if (++li >= 56) li = 1;
        if (++lp >= 56) lp = 1;
        int ret = (int)((uint32_t)sa[li] - (uint32_t)sa[lp]);
        if (ret == INT32_MAX) ret--;
        if (ret < 0) ret += INT32_MAX;
        sa[li] = ret;
        inext = li;
        inextp = lp;
        return ret;
    }
    double sample() { return internal_sample() * (1.0 / INT32_MAX); }
    // NextSingle of the seeded generator: (float)Sample(), drawn again while the cast rounds up to
    // 1.0f (samples >= 2147483583 do; the BCL rejects them so that the result stays in [0, 1)).
    static inline float single_of_sample(int internal) { return (float)(internal * (1.0 / INT32_MAX)); }
    float next_single()
    {
        for (;;) {
            const float f = single_of_sample(internal_sample());
            if (f < 1.0f) return f;
        }
    }
};

// src/HNSWIndex/GraphData.cs:211-219.  random == 0 gives +inf: reported as -1, the value the
// reference's `topLayer < 0 => return -1` guard (:82) sees on x64.
static inline int level_from_uniform(float random, double dist_rate)
{
    double v = -std::log((double)random) * dist_rate;
    if (!(v < 2147483648.0) || !(v > -2147483649.0)) return -1;
    return (int)v;
}

// MemoryExtensions.Sort(Span<NodeDistance>, DistanceComparer) -- the BCL introsort called at
// src/HNSWIndex/Heuristic.cs:22 (insertion sort <= 16, median of three, heapsort at depth
// limit 2*(log2(n)+1)).  Unstable: this decides the order of equal distances.
namespace dotnet_sort_detail {
using C = FartherFirst;
static inline void swap_at(NodeDist *k, int i, int j) { NodeDist t = k[i]; k[i] = k[j]; k[j] = t; }
static inline void swap_if_greater(NodeDist *k, int i, int j) { if (C::cmp(k[i], k[j]) > 0) swap_at(k, i, j); }
static inline void insertion(NodeDist *k, int n)
{
    for (int i = 0; i < n - 1; i++) {
        NodeDist t = k[i + 1];
        int j = i;
        while (j >= 0 && C::cmp(t, k[j]) < 0) { k[j + 1] = k[j]; j--; }
        k[j + 1] = t;
    }
}
static inline void down_heap(NodeDist *k, int i, int n)
{
    NodeDist d = k[i - 1];
    while (i <= (n >> 1)) {
        int child = 2 * i;
        if (child < n && C::cmp(k[child - 1], k[child]) < 0) child++;
        if (!(C::cmp(d, k[child - 1]) < 0)) break;
        k[i - 1] = k[child - 1];
        i = child;
    }
    k[i - 1] = d;
}
static inline void heap_sort(NodeDist *k, int n)
{
    for (int i = n >> 1; i >= 1; i--) down_heap(k, i, n);
    for (int i = n; i > 1; i--) { swap_at(k, 0, i - 1); down_heap(k, 1, i - 1); }
}
static inline int partition(NodeDist *k, int n)
{
    int hi = n - 1, mid = hi >> 1;
    swap_if_greater(k, 0, mid);
    swap_if_greater(k, 0, hi);
    swap_if_greater(k, mid, hi);
    NodeDist pivot = k[mid];
    swap_at(k, mid, hi - 1);
    int left = 0, right = hi - 1;
    while (left < right) {
        while (C::cmp(k[++left], pivot) < 0) {}
        while (C::cmp(pivot, k[--right]) < 0) {}
        if (left >= right) break;
        swap_at(k, left, right);
    }
    if (left != hi - 1) swap_at(k, left, hi - 1);
    return left;
}
static inline void intro(NodeDist *k, int n, int depth)
{
    int ps = n;
    while (ps > 1) {
        if (ps <= 16) {
            if (ps == 2) { swap_if_greater(k, 0, 1); return; }
            if (ps == 3) { swap_if_greater(k, 0, 1); swap_if_greater(k, 0, 2); swap_if_greater(k, 1, 2); return; }
            insertion(k, ps);
            return;
        }
        if (depth == 0) { heap_sort(k, ps); return; }
        depth--;
        int p = partition(k, ps);
        intro(k + p + 1, ps - (p + 1), depth);
        ps = p;
    }
}
} // namespace dotnet_sort_detail

static inline void dotnet_sort(NodeDist *k, int n)
{
    if (n > 1) dotnet_sort_detail::intro(k, n, 2 * ((31 - __builtin_clz((unsigned)n)) + 1));
}

// LINQ OrderBy(c => c.Dist) (src/HNSWIndex/HNSWIndex.cs:121): stable, float.CompareTo keys.
static inline void stable_sort_by_dist(NodeDist *k, int n)
{
    for (int i = 1; i < n; i++) {
        NodeDist t = k[i];
        int j = i - 1;
        while (j >= 0 && float_compare_to(t.dist, k[j].dist) < 0) { k[j + 1] = k[j]; j--; }
        k[j + 1] = t;
    }
}

// Visited set of one search (set semantics of src/HNSWIndex/VisitedListPool.cs:10-67):
// a bitset over node ids plus the list of touched words, so that clearing costs
// O(visited), not O(capacity).
struct Visited {
    std::vector<uint64_t> bits;
    std::vector<uint32_t> touched;
    void begin(int capacity)
    {
        size_t words = ((size_t)capacity + 63) >> 6;
        if (bits.size() < words) { bits.assign(words, 0); touched.clear(); return; }
        for (uint32_t w : touched) bits[w] = 0;
        touched.clear();
    }
    // returns true if id was already present; inserts it otherwise
    inline bool test_and_set(int id)
    {
        uint32_t w = (uint32_t)id >> 6;
        uint64_t m = 1ull << (id & 63);
        uint64_t v = bits[w];
        if (v & m) return true;
        if (v == 0) touched.push_back(w);
        bits[w] = v | m;
        return false;
    }
};

// Graph adjacency.  The reference keeps one `Node` object with per-layer EdgeList (int[] +
// Count, append / replace; src/HNSWIndex/Node.cs:7-107).  Here layer 0 -- where nearly all
// traversal happens -- is one fixed-stride array (row id*stride0: [count, e0, e1, ...]),
// upper layers live in a side pool; ids, append order and replace semantics are unchanged.
struct Graph {
    int max_edges = 16;
    int stride0 = 0; // 1 + 2M + 1
    int strideU = 0; // 1 + M + 1
    std::vector<int> level;     // per node: MaxLayer
    std::vector<int> adj0;      // n * stride0
    std::vector<int64_t> upper; // per node: offset into pool (-1 if level 0)
    std::vector<int> pool;      // blocks of strideU per (node, layer >= 1)
    int entry = -1;
    int length = 0; // slots ever allocated (GraphData.Length)
    // ActiveSet (src/HNSWIndex/ActiveSet.cs:10-97: dense/sparse, swap-with-last removal),
    // IsRemoved flags (Node.cs:18) and RemovedIndexes (GraphData.cs:19: LIFO slot reuse)
    std::vector<int> dense, sparse, removed_stack;
    std::vector<char> removed;
    int count = 0; // GraphData.Count

    void configure(int M)
    {
        max_edges = M;
        stride0 = 2 * M + 2;
        strideU = M + 2;
    }
    inline int max_edges_at(int layer) const { return layer == 0 ? max_edges * 2 : max_edges; } // GraphData.cs:247-250
    void reserve(int capacity)
    {
        level.reserve((size_t)capacity);
        upper.reserve((size_t)capacity);
        adj0.reserve((size_t)capacity * stride0);
        removed.reserve((size_t)capacity);
        dense.reserve((size_t)capacity);
        sparse.reserve((size_t)capacity);
    }
    // Room for n more fresh slots in every per-node array, in one step (slots [length, slots) exist, zeroed, unused).
    size_t slots = 0;
    void grow_for(int n)
    {
        const size_t want = (size_t)length + (size_t)std::max(n, 0);
        if (want <= slots) return;
        level.resize(want, 0);
        adj0.resize(want * (size_t)stride0, 0);
        upper.resize(want, -1);
        removed.resize(want, 0);
        sparse.resize(want, 0);
        dense.resize(want, 0);
        slots = want;
    }
    // GraphData.AddItem :85-115 + NewNode :224-242.  reuse: pop the most recently vacated slot.
    int add_node(int top_layer, bool reuse, bool *reused = nullptr)
    {
        int id;
        if (reuse && !removed_stack.empty()) {
            id = removed_stack.back();
            removed_stack.pop_back();
            level[(size_t)id] = top_layer;
            std::memset(adj0.data() + (size_t)id * stride0, 0, sizeof(int) * (size_t)stride0);
            removed[(size_t)id] = 0;
            if (reused) *reused = true;
        } else {
            if ((size_t)length >= slots) grow_for(std::max(1024, length / 2));
            id = length++;
            level[(size_t)id] = top_layer;
        }
        if (top_layer > 0) {
            upper[(size_t)id] = (int64_t)pool.size();
            pool.resize(pool.size() + (size_t)top_layer * strideU, 0);
        } else {
            upper[(size_t)id] = -1;
        }
        dense[(size_t)count] = id; // ActiveSet.Add :72-80
        sparse[(size_t)id] = count;
        ++count;
        return id;
    }
    // GraphData.RemoveItem :124-128 + ActiveSet.Remove :85-97
    void retire(int id)
    {
        removed_stack.push_back(id);
        const int idx = sparse[(size_t)id], last = --count, last_id = dense[(size_t)last];
        dense[(size_t)idx] = last_id;
        sparse[(size_t)last_id] = idx;
    }
    // pointer to [count, e0, e1, ...] of (id, layer)
    inline int *list(int id, int layer)
    {
        return layer == 0 ? adj0.data() + (size_t)id * stride0 : pool.data() + upper[id] + (size_t)(layer - 1) * strideU;
    }
    inline const int *list(int id, int layer) const
    {
        return layer == 0 ? adj0.data() + (size_t)id * stride0 : pool.data() + upper[id] + (size_t)(layer - 1) * strideU;
    }
    inline int top_layer() const { return level[entry]; } // GraphData.GetTopLayer :195-198
};

// Order-sensitive digest of the graph (entry point, levels, every list in EdgeList order); the
// test oracle computes the same value for whole-graph parity checks.
inline uint64_t graph_hash_of(const Graph &g)
{
    uint64_t x = 1469598103934665603ULL;
    auto mix = [&](int v) {
        uint32_t u = (uint32_t)v;
        for (int b = 0; b < 4; ++b) { x ^= (u >> (8 * b)) & 0xff; x *= 1099511628211ULL; }
    };
    mix(g.entry);
    for (int i = 0; i < g.length; ++i) {
        if (g.removed[(size_t)i]) { mix(-2); continue; } // a vacated slot: its stale lists are unreachable
        mix(g.level[(size_t)i]);
        for (int l = 0; l <= g.level[(size_t)i]; ++l) {
            const int *e = g.list(i, l);
            mix(e[0]);
            for (int j = 1; j <= e[0]; ++j) mix(e[j]);
        }
    }
    return x;
}

} // namespace hnsw
