// range_replay.h -- RangeQuery's result ORDER when two results have the same distance.
#pragma once
#include <algorithm>
#include <cstdint>
#include <limits>
#include <vector>

#include "host_structs.h"

namespace hnsw {

// SearchLayerRange's two heaps (GraphNavigator.cs:262-325) replayed from `entry` with every distance already
// known: `found` is the query's whole result set as the device kernel measured it, so a neighbour that is not in it
// is out of range -- it would be marked visited and dropped (:302, :318) -- and needs no evaluation; the entry
// point's own distance is never compared when it is out of range (it is alone in `candidates` when popped, and
// farthestResultDist is still MaxValue at :286).  Result: topCandidates' array, stably sorted (HNSWIndex.cs:155).
// list_of(id) -> the node's layer-0 list as [count, ids...]; Hit: {int id; float dist}.
template <class ListOf, class Hit>
inline void replay_range_heaps(ListOf list_of, int max_edges0, int entry, float range, const Hit *found, int m, std::vector<NodeDist> &out)
{
    size_t cap = 16;
    while (cap < 2 * (size_t)m) cap <<= 1;
    struct Slot { int id; float dist; bool visited; };
    std::vector<Slot> tab(cap, Slot{-1, 0.0f, false});
    auto slot_of = [&](int id) -> Slot * { // the slot holding id, or nullptr
        size_t h = ((size_t)(uint32_t)id * 2654435761u) & (cap - 1);
        while (tab[h].id != -1) {
            if (tab[h].id == id) return &tab[h];
            h = (h + 1) & (cap - 1);
        }
        return nullptr;
    };
    for (int i = 0; i < m; ++i) {
        size_t h = ((size_t)(uint32_t)found[i].id * 2654435761u) & (cap - 1);
        while (tab[h].id != -1) h = (h + 1) & (cap - 1);
        tab[h] = Slot{found[i].id, found[i].dist, false};
    }
    BinaryHeap<FartherFirst> top;
    BinaryHeap<CloserFirst> cand;
    top.reset(max_edges0);      // :265
    cand.reset(max_edges0 * 2); // :266
    float farthest = std::numeric_limits<float>::max(); // :269
    Slot *es = slot_of(entry);
    NodeDist e{entry, es ? es->dist : std::numeric_limits<float>::infinity()};
    if (es) { top.push(e); farthest = e.dist; es->visited = true; } // :271-275, :279
    cand.push(e);                                                  // :277
    while (cand.count > 0) {
        const NodeDist closest = cand.peek();                                        // :285
        if (es == nullptr && closest.id == entry) { /* :286-289 cannot fire: farthest is MaxValue */ }
        else if (closest.dist > farthest && closest.dist > range) break;
        cand.pop();                                                                  // :290
        const int *l = list_of(closest.id);
        for (int i = 1; i <= l[0]; ++i) {
            Slot *sl = slot_of(l[i]);
            if (!sl || sl->visited) continue; // :297, or out of range (:302 fails, :318)
            sl->visited = true;
            NodeDist sel{sl->id, sl->dist};
            cand.push(sel);                                   // :305
            top.push(sel);                                    // :308
            if (top.peek().dist > range) top.pop();           // :310-311
            if (top.count > 0) farthest = top.peek().dist;    // :313-314
        }
    }
    out.assign(top.buf.begin(), top.buf.begin() + top.count);
    std::stable_sort(out.begin(), out.end(), [](const NodeDist &a, const NodeDist &b) { return float_compare_to(a.dist, b.dist) < 0; });
}

} // namespace hnsw
