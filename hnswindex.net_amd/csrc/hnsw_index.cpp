// hnsw_index.cpp -- Add / KnnQuery on the lock-step engine.  Citations relative to
// /root/reference/.
#include "hnsw_index.h"
#include "diag.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <limits>
#include <thread>
#include <unordered_map>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "range_replay.h"
#include "snapshot_io.h"

namespace hnsw {

namespace {

// ------------------------------------------------------------------------------------
// Pieces of the reference's traversal, cut at every distance evaluation.
// ------------------------------------------------------------------------------------

// GraphNavigator.FindEntryPoint / FindEntryAtLayer (src/HNSWIndex/GraphNavigator.cs:27-82):
// greedy descent from the entry point's top layer to dst_layer (exclusive).
struct Descent {
    const Graph *g = nullptr;
    int layer = 0, dst = 0, best = -1;
    float cur = 0.f;
    bool need_init = true;

    void begin(const Graph *graph, int entry, int dst_layer)
    {
        g = graph;
        best = entry;
        layer = g->level[entry];
        dst = dst_layer;
        need_init = true;
    }
    // true: a task was emitted.  false: descent complete, (best, cur) final.
    bool prepare(SlotIO &io)
    {
        if (need_init) { // :57 currDist = Distance(start, query) -- once; lower layers reuse it
            io.ids[0] = best;
            *io.cnt = 1;
            return true;
        }
        while (layer > dst) {
            const int *l = g->list(best, layer);
            int n = l[0];
            if (n == 0) { --layer; continue; } // empty span: `changed` stays false
            std::memcpy(io.ids, l + 1, sizeof(int) * (size_t)n); // :65 span taken once per pass
            *io.cnt = n;
            return true;
        }
        return false;
    }
    void consume(const SlotIO &io)
    {
        if (need_init) { cur = io.dist[0]; need_init = false; return; }
        bool changed = false;
        const int n = *io.cnt;
        for (int i = 0; i < n; ++i) { // :67-78
            float d = io.dist[i];
            if (d < cur) { cur = d; best = io.ids[i]; changed = true; }
        }
        if (!changed) --layer; // :60 loop ends; FindEntryPoint moves one layer down (:30-31)
    }
};

// GraphNavigator.SearchLayer / SearchLayerQuery (GraphNavigator.cs:123-256), no filter.
struct Expand {
    const Graph *g = nullptr;
    int layer = 0, k = 0;
    int exclude = -1; // the filter `id => id != removedNode.Id` of GraphConnector.cs:96 (-1: none)
    float farthest = 0.f;

    void begin(const Graph *graph, SlotScratch &sc, int capacity, int entry, float entry_dist, int at_layer, int kk, int exclude_id = -1)
    {
        g = graph;
        layer = at_layer;
        k = kk;
        exclude = exclude_id;
        sc.top.reset(k);      // :126
        sc.cand.reset(k * 2); // :127
        NodeDist e{entry, entry_dist};
        farthest = std::numeric_limits<float>::max(); // TDistance.MaxValue :130
        if (entry != exclude) { // filterFnc(entryPointId) :132
            sc.top.push(e);       // :134
            farthest = entry_dist; // :135
        }
        sc.cand.push(e);      // :138
        sc.visited.begin(capacity);
        sc.visited.test_and_set(entry); // :140
    }
    bool prepare(SlotIO &io, SlotScratch &sc)
    {
        while (sc.cand.count > 0) {
            NodeDist c = sc.cand.pop();                              // :146
            if (c.dist > farthest && sc.top.count >= k) return false; // :147-150
            const int *l = g->list(c.id, layer);
            int n = 0;
            for (int i = 1; i <= l[0]; ++i) {
                int nb = l[i];
                if (!sc.visited.test_and_set(nb)) io.ids[n++] = nb; // :161 / :181 (lists hold no duplicates)
            }
            if (n > 0) { *io.cnt = n; return true; }
        }
        return false;
    }
    void consume(const SlotIO &io, SlotScratch &sc)
    {
        const int n = *io.cnt;
        for (int i = 0; i < n; ++i) {
            float d = io.dist[i];
            if (sc.top.count < k || d < farthest) { // :165
                NodeDist sel{io.ids[i], d};
                sc.cand.push(sel);                  // :168
                if (sel.id != exclude) sc.top.push(sel); // :170-171
                if (sc.top.count > k) sc.top.pop(); // :173-174
                if (sc.top.count > 0) farthest = sc.top.peek().dist; // :176-177
            }
        }
    }
};

// Heuristic.RelativeNeighborPruning (src/HNSWIndex/Heuristic.cs:11-46).  One task per
// candidate: its distances to the already accepted ids (the reference's inner loop :31-35;
// its early break only skips evaluations, never changes the outcome).
struct Prune {
    std::vector<NodeDist> cands;
    std::vector<int> acc;
    int max_edges = 0, i = 0;
    bool sorted_path = false;

    void begin(int maxE)
    {
        max_edges = maxE;
        acc.clear();
        i = 0;
        const int n = (int)cands.size();
        if (n < max_edges) { // :13-18 ids in input (heap) order, unsorted
            sorted_path = false;
            for (const auto &c : cands) acc.push_back(c.id);
            i = n;
            return;
        }
        sorted_path = true;
        dotnet_sort(cands.data(), n); // :22
    }
    bool prepare(SlotIO &io)
    {
        const int n = (int)cands.size();
        while (i < n && (int)acc.size() < max_edges) { // :23
            if (acc.empty()) { acc.push_back(cands[(size_t)i].id); ++i; continue; }
            *io.qidx = ~cands[(size_t)i].id; // distanceFnc(s.Id, candidateId) :34
            std::memcpy(io.ids, acc.data(), sizeof(int) * acc.size());
            *io.cnt = (int)acc.size();
            return true;
        }
        return false;
    }
    void consume(const SlotIO &io)
    {
        const NodeDist c = cands[(size_t)i];
        bool ok = true;
        const int n = *io.cnt;
        for (int j = 0; j < n; ++j)
            if (io.dist[j] < c.dist) { ok = false; break; }
        if (ok) acc.push_back(c.id);
        ++i;
    }
};

// ------------------------------------------------------------------------------------
// KnnQuery job: HNSWIndex.KnnQuery (src/HNSWIndex/HNSWIndex.cs:107-124), layer 0, no filter.
// ------------------------------------------------------------------------------------
struct QueryJob : Job {
    const Graph *g;
    int capacity, qi, ef, k;
    int *out_ids;
    float *out_d;
    Descent desc;
    Expand exp;
    int stage = 0; // 0 descent, 1 expand

    bool prepare(SlotIO &io, SlotScratch &sc) override
    {
        *io.qidx = qi;
        if (stage == 0) {
            if (desc.prepare(io)) return true;
            exp.begin(g, sc, capacity, desc.best, desc.cur, 0, ef); // :117 (entry distance reused, same bits)
            stage = 1;
        }
        return exp.prepare(io, sc);
    }
    void consume(const SlotIO &io, SlotScratch &sc) override
    {
        if (stage == 0) desc.consume(io);
        else exp.consume(io, sc);
    }
    void finish(SlotScratch &sc)
    {
        int n = sc.top.count;
        sc.tmp.assign(sc.top.buf.begin(), sc.top.buf.begin() + n); // ToArray(): heap order
        stable_sort_by_dist(sc.tmp.data(), n);                     // OrderBy(c => c.Dist) :121
        int m = std::min(n, k);
        for (int j = 0; j < m; ++j) { out_ids[j] = sc.tmp[(size_t)j].id; out_d[j] = sc.tmp[(size_t)j].dist; }
        for (int j = m; j < k; ++j) { out_ids[j] = -1; out_d[j] = std::numeric_limits<float>::quiet_NaN(); } // Exports.cs:144
    }
};

struct QuerySource : JobSource {
    std::vector<QueryJob> jobs;
    std::atomic<int> next{0};
    Job *acquire(SlotScratch &) override
    {
        int i = next.fetch_add(1, std::memory_order_relaxed);
        if (i >= (int)jobs.size()) return nullptr;
        QueryJob &j = jobs[(size_t)i];
        j.desc.begin(j.g, j.g->entry, 0); // FindEntryPointQuery(layer 0) :116
        j.stage = 0;
        return &j;
    }
    void release(Job *job, SlotScratch &sc) override { static_cast<QueryJob *>(job)->finish(sc); }
};

// ------------------------------------------------------------------------------------
// RangeQuery job: HNSWIndex.RangeQuery (src/HNSWIndex/HNSWIndex.cs:144-156) =
// FindEntryPointQuery + GraphNavigator.SearchLayerRange (GraphNavigator.cs:262-325), no filter.
// ------------------------------------------------------------------------------------
struct RangeJob : Job {
    const Graph *g;
    int capacity, qi;
    float range, farthest;
    std::vector<NodeDist> *out;
    Descent desc;
    int stage = 0;

    void begin_search(SlotScratch &sc)
    {
        const int maxE = g->max_edges_at(0);
        sc.top.reset(maxE);      // :265
        sc.cand.reset(maxE * 2); // :266
        NodeDist e{desc.best, desc.cur}; // :268 (entry distance reused, same bits)
        farthest = std::numeric_limits<float>::max(); // TDistance.MaxValue :269
        if (e.dist <= range) { sc.top.push(e); farthest = e.dist; } // :271-275
        sc.cand.push(e);         // :277
        sc.visited.begin(capacity);
        sc.visited.test_and_set(e.id); // :279
    }
    bool prepare(SlotIO &io, SlotScratch &sc) override
    {
        *io.qidx = qi;
        if (stage == 0) {
            if (desc.prepare(io)) return true;
            begin_search(sc);
            stage = 1;
        }
        while (sc.cand.count > 0) {
            const NodeDist closest = sc.cand.peek();                             // :285
            if (closest.dist > farthest && closest.dist > range) return false;  // :286-289
            sc.cand.pop();                                                       // :290
            const int *l = g->list(closest.id, 0);
            int n = 0;
            for (int i = 1; i <= l[0]; ++i)
                if (!sc.visited.test_and_set(l[i])) io.ids[n++] = l[i];          // :297 / :318
            if (n > 0) { *io.cnt = n; return true; }
        }
        return false;
    }
    void consume(const SlotIO &io, SlotScratch &sc) override
    {
        if (stage == 0) { desc.consume(io); return; }
        const int n = *io.cnt;
        for (int i = 0; i < n; ++i) {
            const float d = io.dist[i];
            if (d <= range) { // :302
                NodeDist sel{io.ids[i], d};
                sc.cand.push(sel);                                   // :305
                sc.top.push(sel);                                    // :308
                if (sc.top.peek().dist > range) sc.top.pop();        // :310-311
                if (sc.top.count > 0) farthest = sc.top.peek().dist; // :313-314
            }
        }
    }
    void finish(SlotScratch &sc)
    {
        out->assign(sc.top.buf.begin(), sc.top.buf.begin() + sc.top.count);
        std::stable_sort(out->begin(), out->end(), [](const NodeDist &a, const NodeDist &b) { return float_compare_to(a.dist, b.dist) < 0; }); // OrderBy :155
    }
};

struct RangeSource : JobSource {
    std::vector<RangeJob> jobs;
    std::atomic<int> next{0};
    Job *acquire(SlotScratch &) override
    {
        int i = next.fetch_add(1, std::memory_order_relaxed);
        if (i >= (int)jobs.size()) return nullptr;
        RangeJob &j = jobs[(size_t)i];
        j.desc.begin(j.g, j.g->entry, 0);
        j.stage = 0;
        return &j;
    }
    void release(Job *job, SlotScratch &sc) override { static_cast<RangeJob *>(job)->finish(sc); }
};

// ------------------------------------------------------------------------------------
// Insert, search half: GraphConnector.AddNewConnections (src/HNSWIndex/GraphConnector.cs:172-181)
// with ConnectAtLayer's search + heuristic (:189-190) for every layer of the new node,
// against the graph as it stands at the start of the batch.
// ------------------------------------------------------------------------------------
struct InsertJob : Job {
    const Graph *g;
    int capacity, id, level, efc;
    std::vector<std::vector<int>> selected; // per layer
    Descent desc;
    Expand exp;
    Prune prune;
    int stage = 0; // 0 descent, 1 expand, 2 prune
    int layer = 0;

    void start()
    {
        selected.assign((size_t)level + 1, {});
        desc.begin(g, g->entry, level); // FindEntryPoint(currNode.MaxLayer, item) :174
        stage = 0;
        layer = std::min(level, g->top_layer()); // :176
    }
    bool prepare(SlotIO &io, SlotScratch &sc) override
    {
        for (;;) {
            *io.qidx = ~id; // the item is already a stored row
            if (stage == 0) {
                if (desc.prepare(io)) return true;
                exp.begin(g, sc, capacity, desc.best, desc.cur, layer, efc); // SearchLayer(bestPeer, layer, MaxCandidates, item) :189
                stage = 1;
            }
            if (stage == 1) {
                if (exp.prepare(io, sc)) return true;
                prune.cands.assign(sc.top.buf.begin(), sc.top.buf.begin() + sc.top.count);
                prune.begin(g->max_edges_at(layer)); // :190
                stage = 2;
            }
            if (prune.prepare(io)) return true;
            selected[(size_t)layer] = prune.acc;
            if (layer == 0) return false;
            // next layer: entry = selected[0] (:216, :179); its distance to the item is the
            // candidate's own search distance (cands[0] in both Heuristic paths)
            int entry = prune.acc[0];
            float ed = prune.cands[0].dist;
            --layer;
            exp.begin(g, sc, capacity, entry, ed, layer, efc);
            stage = 1;
        }
    }
    void consume(const SlotIO &io, SlotScratch &sc) override
    {
        if (stage == 0) desc.consume(io);
        else if (stage == 1) exp.consume(io, sc);
        else prune.consume(io);
    }
};

struct InsertSource : JobSource {
    std::vector<InsertJob> jobs;
    std::atomic<int> next{0};
    Job *acquire(SlotScratch &) override
    {
        int i = next.fetch_add(1, std::memory_order_relaxed);
        if (i >= (int)jobs.size()) return nullptr;
        jobs[(size_t)i].start();
        return &jobs[(size_t)i];
    }
    void release(Job *, SlotScratch &) override {}
};

// ------------------------------------------------------------------------------------
// Insert, link half: the back-edge loop of ConnectAtLayer (:196-214) and PruneOverflow
// (:222-262), regrouped per (neighbour, layer): every append to one adjacency list, in
// item order, with its overflow prune.  Distinct lists never interact, so the groups run
// as independent jobs and the outcome equals the sequential loop's.
// ------------------------------------------------------------------------------------
struct LinkJob : Job {
    Graph *g;
    int nb, layer;
    std::vector<int> items;
    size_t idx = 0;
    int stage = 0; // 0 idle, 1 waiting for node<->edge distances, 2 pruning
    Prune prune;

    bool prepare(SlotIO &io, SlotScratch &) override
    {
        const int maxE = g->max_edges_at(layer);
        for (;;) {
            if (stage == 2) {
                if (prune.prepare(io)) return true;
                int *l = g->list(nb, layer); // node.OutEdges[layer] = newOut :236
                l[0] = (int)prune.acc.size();
                std::memcpy(l + 1, prune.acc.data(), sizeof(int) * prune.acc.size());
                stage = 0;
            }
            if (idx == items.size()) return false;
            int *l = g->list(nb, layer);
            l[1 + l[0]] = items[idx++]; // neighbor.OutEdges[layer].Add(currNode.Id) :207
            l[0]++;
            if (l[0] > maxE) { // :209
                *io.qidx = ~nb; // Distance(cand, node.Id) :233
                std::memcpy(io.ids, l + 1, sizeof(int) * (size_t)l[0]);
                *io.cnt = l[0];
                stage = 1;
                return true;
            }
        }
    }
    void consume(const SlotIO &io, SlotScratch &) override
    {
        if (stage == 1) {
            const int n = *io.cnt;
            prune.cands.resize((size_t)n);
            for (int i = 0; i < n; ++i) prune.cands[(size_t)i] = NodeDist{io.ids[i], io.dist[i]}; // :230-234
            prune.begin(g->max_edges_at(layer)); // :235
            stage = 2;
        } else {
            prune.consume(io);
        }
    }
};

struct LinkSource : JobSource {
    std::vector<LinkJob> jobs;
    std::atomic<int> next{0};
    Job *acquire(SlotScratch &) override
    {
        int i = next.fetch_add(1, std::memory_order_relaxed);
        if (i >= (int)jobs.size()) return nullptr;
        return &jobs[(size_t)i];
    }
    void release(Job *, SlotScratch &) override {}
};

// ------------------------------------------------------------------------------------
// Removal (GraphConnector.RemoveConnectionsAtLayer, src/HNSWIndex/GraphConnector.cs:90-167).
// ------------------------------------------------------------------------------------
// :96  SearchLayer(removedNode.Id, layer, RemoveMaxCandidates, Items[removedNode.Id], id => id != removedNode.Id)
struct RemoveSearchJob : Job {
    const Graph *g;
    int capacity, removed, layer, k;
    std::vector<NodeDist> result; // heap order
    Expand exp;
    int stage = 0;
    bool prepare(SlotIO &io, SlotScratch &sc) override
    {
        *io.qidx = ~removed;
        if (stage == 0) { io.ids[0] = removed; *io.cnt = 1; return true; } // :129 Distance(entry, query)
        return exp.prepare(io, sc);
    }
    void consume(const SlotIO &io, SlotScratch &sc) override
    {
        if (stage == 0) { exp.begin(g, sc, capacity, removed, io.dist[0], layer, k, removed); stage = 1; return; }
        exp.consume(io, sc);
    }
    void finish(SlotScratch &sc) { result.assign(sc.top.buf.begin(), sc.top.buf.begin() + sc.top.count); }
};

// :100-165  one affected node: drop the edge to the removed node, re-select its neighbours
// among (old neighbours + search candidates), apply the difference to its own list.
struct AffectedJob : Job {
    Graph *g;
    int aid, layer, removed;
    const std::vector<NodeDist> *sc_cands;
    std::vector<int> old_ids;
    std::vector<NodeDist> cands;
    std::vector<int> in_remove, in_add; // in-edge deltas: (old -> aid) dropped, (w -> aid) added
    size_t pos = 0;
    int stage = 0; // 0 gather, 1 distances, 2 prune
    Prune prune;

    static bool has(const std::vector<int> &v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); }
    static void swap_remove(int *l, int x) // EdgeList.Remove, Node.cs:79-93
    {
        for (int i = 1; i <= l[0]; ++i)
            if (l[i] == x) { int last = l[0]--; if (i != last) l[i] = l[last]; return; }
    }
    void gather() // RemoveOutEdge + the existing neighbours
    {
        int *l = g->list(aid, layer);
        swap_remove(l, removed); // RemoveOutEdge :104
        old_ids.assign(l + 1, l + 1 + l[0]); // :110-111
    }
    bool prepare(SlotIO &io, SlotScratch &) override
    {
        if (stage == 0) {
            gather();
            cands.clear();
            for (int id : old_ids) cands.push_back(NodeDist{id, 0.f}); // :115-120
            for (const NodeDist &c : *sc_cands) { // :123-129
                if (c.id == aid || has(old_ids, c.id)) continue;
                cands.push_back(NodeDist{c.id, 0.f});
            }
            pos = 0;
            stage = 1;
        }
        if (stage == 1) {
            if (pos < cands.size()) { // Distance(id, affectedNodeId) in chunks of one task
                const int n = (int)std::min<size_t>((size_t)io.stride, cands.size() - pos);
                *io.qidx = ~aid;
                for (int i = 0; i < n; ++i) io.ids[i] = cands[pos + (size_t)i].id;
                *io.cnt = n;
                return true;
            }
            prune.cands = cands;
            prune.begin(g->max_edges_at(layer)); // :131
            stage = 2;
        }
        return prune.prepare(io);
    }
    void consume(const SlotIO &io, SlotScratch &) override
    {
        if (stage == 1) {
            const int n = *io.cnt;
            for (int i = 0; i < n; ++i) cands[pos + (size_t)i].dist = io.dist[i];
            pos += (size_t)n;
        } else {
            prune.consume(io);
        }
    }
    void finish() { apply(prune.acc); }
    void apply(const std::vector<int> &nw) // the new selection against the old list (:135-164)
    {
        int *l = g->list(aid, layer);
        for (int o : old_ids) { // :135-143
            if (has(nw, o)) continue;
            swap_remove(l, o);
            in_remove.push_back(o);
        }
        for (int w : nw) { // :146-164
            if (has(old_ids, w)) continue;
            if (g->removed[(size_t)w]) continue; // :155
            l[1 + l[0]] = w;
            l[0]++;
            in_add.push_back(w);
        }
    }
};

template <class J>
struct VecSource : JobSource {
    std::vector<J> jobs;
    std::atomic<int> next{0};
    Job *acquire(SlotScratch &) override
    {
        int i = next.fetch_add(1, std::memory_order_relaxed);
        return i < (int)jobs.size() ? &jobs[(size_t)i] : nullptr;
    }
    void release(Job *job, SlotScratch &sc) override { finish_job(static_cast<J *>(job), sc); }
    static void finish_job(RemoveSearchJob *j, SlotScratch &sc) { j->finish(sc); }
    static void finish_job(AffectedJob *j, SlotScratch &) { j->finish(); }
};

template <class F>
void parallel_for(int n, int threads, F fn)
{
    threads = std::max(1, std::min(threads, n / 64));
    if (threads == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    std::vector<std::thread> th;
    std::atomic<int> next{0};
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&] {
            for (;;) {
                int i0 = next.fetch_add(64, std::memory_order_relaxed);
                if (i0 >= n) break;
                for (int i = i0, e = std::min(n, i0 + 64); i < e; ++i) fn(i);
            }
        });
    for (auto &t : th) t.join();
}

// Growth rule of the snapshot schedule: a batch never exceeds 1/4 of the linked graph while that is below
// 65 536 nodes, 1/16 of it afterwards.  Measured on 1M x 128 (tools/ramp_study.py, three data seeds, 4 000
// queries each; profiles/r2_ramp_study.json): against 1/16 throughout, the build takes 1.24 s instead of
// 1.44 s (the first 65 k inserts were 133 latency-bound batches, now 50) at unchanged recall@10 -- uniform
// 0.2343 / 0.2377 / 0.2354 vs 0.2321 / 0.2381 / 0.2362, clustered 0.9872 / 0.9840 / 0.9876 vs 0.9876 / 0.9827 /
// 0.9875.  Growing the LATE batches the same way (1/4 throughout: 1.20 s) did cost recall on clustered data
// (0.9843 vs 0.9907; 1/16 early + 1/4 late: 0.9817), so the large batches keep the 1/16 rule.
// "Early" is relative to what the call will leave behind: the 1/4 rule applies below min(65 536, count after
// the call / 16) linked nodes, so that on a small index the last batches are not a quarter of the graph (a
// 32 768-node build under 1/4 throughout lost 1.3 points of recall@10 against 1/16).
constexpr int kBatchGrowthDiv = 16, kEarlyGrowthDiv = 4, kEarlyLinked = 65536;

// Optional phase timing (HNSW_MI355X_TRACE=1): printed when the index is destroyed.
struct PhaseTimers {
    double sync_graph = 0, search_half = 0, collect = 0, link_host = 0, link_dev = 0, post = 0, query_dev = 0, set_queries = 0;
    double add_nodes = 0, add_upload = 0, add_total = 0;
    double rq_batch = 0, rq_sort = 0, rq_refresh = 0, rq_replay = 0; long rq_replayed = 0, rq_queries = 0; // RangeQuery on the device: launch + copies, host sort, list refresh, heap replays
    double xw_total = 0, xw_todo = 0, xw_launch = 0, xw_parse = 0, xw_pairs = 0, xw_valid = 0, xw_link = 0; // the exact window's rounds
    long xw_end[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // what ended a round's prefix: 0 window exhausted, 1 stale result (searched on an older graph, list written since), 2 second change of a list,
                                               // 3 change with unknown lost ids, 4 reader answered by the exact traversal, 5 expansion without a bound, 6 the change shows (pair distance), 7 upper layers / overflow
    long rounds = 0, batches = 0, prune_jobs = 0;
    bool on = diag("trace", 0) != 0;
};
PhaseTimers g_pt;
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Tick {
    double &acc; double t0;
    explicit Tick(double &a) : acc(a), t0(g_pt.on ? now_s() : 0) {}
    ~Tick() { if (g_pt.on) acc += now_s() - t0; }
}; // a snapshot batch never exceeds 1/32 of the linked graph

} // namespace

// ------------------------------------------------------------------------------------
HnswIndex *HnswIndex::create(int metric, const Params &p, std::string &err)
{
    if (metric < 0 || metric > 3) { err = "Unsupported distance metric"; return nullptr; }
    int ndev = hnswdev_device_count();
    if (ndev <= 0) {
        err = "HNSWIndex MI355X backend: no HIP device available (" + get_dev_error() +
              "); this library has no CPU fallback";
        return nullptr;
    }
    int dev = p.device;
    if (dev < 0) dev = 0;
    if (dev < 0 || dev >= ndev) { err = "HNSWIndex MI355X backend: device ordinal out of range"; return nullptr; }
    if (p.max_edges < 1) { err = "MaxEdges must be >= 1"; return nullptr; }
    int devices = p.devices;
    if (devices <= 0) devices = 1;
    if (devices < 1 || devices > 64) { err = "HNSWIndex MI355X backend: device contexts must be between 1 and 64"; return nullptr; }
    HnswIndex *ix = new HnswIndex();
    ix->metric_ = metric;
    ix->p_ = p;
    ix->p_.devices = devices;
    ix->device_ordinal_ = dev;
    ix->graph_.configure(p.max_edges);
    // RandomSeed < 0 means an unseeded Random() in the reference (GraphData.cs:42); a
    // time-based seed reproduces that behaviour.
    int seed = p.random_seed;
    if (seed < 0) seed = (int)(std::chrono::steady_clock::now().time_since_epoch().count() & 0x7fffffff);
    ix->rng_.init(seed);
    ix->capacity_ = std::max(1, p.collection_size);
    int th = p.host_threads;
    if (th <= 0) th = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    ix->threads_ = th;
    return ix;
}

int HnswIndex::host_parallelism() { return (int)std::max(1u, std::thread::hardware_concurrency()); }

HnswIndex::~HnswIndex()
{
    if (g_pt.on)
        fprintf(stderr, "[hnsw trace] add: total=%.3fs nodes=%.3fs upload=%.3fs\n", g_pt.add_total, g_pt.add_nodes, g_pt.add_upload),
        fprintf(stderr, "[hnsw trace] batches=%ld sync_graph=%.3fs search_half=%.3fs collect=%.3fs link_host=%.3fs link_dev=%.3fs (rounds=%ld prune_jobs=%ld) | query: set_queries=%.3fs dev=%.3fs post=%.3fs\n",
                g_pt.batches, g_pt.sync_graph, g_pt.search_half, g_pt.collect, g_pt.link_host, g_pt.link_dev, g_pt.rounds, g_pt.prune_jobs, g_pt.set_queries, g_pt.query_dev, g_pt.post);
    if (g_pt.on && g_pt.xw_total > 0)
        fprintf(stderr, "[hnsw trace] exact window: total=%.3fs | what to search %.3fs, search launch + wait %.3fs, results -> specs %.3fs, pair distances %.3fs, validation %.3fs, prefix link %.3fs\n"
                        "[hnsw trace] exact window, what ended the rounds' prefixes: window exhausted %ld, stale result %ld, second change of a list %ld, change with unknown lost ids %ld, "
                        "reader took the exact traversal %ld, expansion without a bound %ld, the change shows %ld, upper layers / no read log %ld\n",
                g_pt.xw_total, g_pt.xw_todo, g_pt.xw_launch, g_pt.xw_parse, g_pt.xw_pairs, g_pt.xw_valid, g_pt.xw_link,
                g_pt.xw_end[0], g_pt.xw_end[1], g_pt.xw_end[2], g_pt.xw_end[3], g_pt.xw_end[4], g_pt.xw_end[5], g_pt.xw_end[6], g_pt.xw_end[7]);
    if (g_pt.on && g_pt.rq_queries > 0)
        fprintf(stderr, "[hnsw trace] range queries=%ld: device launch + copies %.4fs, host sort %.4fs, host list refresh %.4fs, heap replays %.4fs (%ld lists held equal distances)\n",
                g_pt.rq_queries, g_pt.rq_batch, g_pt.rq_sort, g_pt.rq_refresh, g_pt.rq_replay, g_pt.rq_replayed);
    engine_.reset(); // before the device it allocates from
    for (auto &l : lanes_) l.reset();
    replicas_.clear();
    dev_.reset();
}

bool HnswIndex::ensure_dim(int dim, std::string &err)
{
    if (dim_ == 0) {
        Device *d = Device::create(device_ordinal_, dim, metric_, capacity_);
        if (!d) { err = get_dev_error(); return false; }
        dev_.reset(d);
        dev_->set_profiling(profiling_);
        dim_ = dim;
        engine_stride_ = ((2 * p_.max_edges + 1) + 7) & ~7; // the lock-step engine itself is created on first use (engine())
        graph_.reserve((int)std::min<long long>(capacity_, 1 << 26));
        return true;
    }
    if (dim != dim_) {
        err = "dimension mismatch: index holds dim=" + std::to_string(dim_) + ", got dim=" + std::to_string(dim);
        return false;
    }
    return true;
}

// The lock-step engine (pinned step buffers, worker threads) is only needed by the host traversal,
// RangeQuery, removal and the rare hand-backs of the device kernels: created when first used.
LockStepEngine *HnswIndex::engine()
{
    if (!engine_) engine_.reset(new LockStepEngine(dev_.get(), std::max(2, p_.search_slots), engine_stride_, threads_));
    return engine_.get();
}

bool HnswIndex::ensure_capacity(long long need, std::string &err)
{
    if (need <= capacity_) return true;
    long long cap = capacity_;
    while (cap < need) cap *= 2; // GraphData.cs:100
    if (!dev_->reserve(cap)) { err = get_dev_error(); return false; }
    capacity_ = cap;
    return true;
}

// ---- search half --------------------------------------------------------------------------
// Host traversal on the lock-step engine for the listed items (indices into the batch).
bool HnswIndex::search_half_lockstep(const std::vector<int> &bid, const std::vector<int> &items, Selection &sel, std::string &err)
{
    if (items.empty()) return true;
    if (!refresh_host_lists(err)) return false; // host traversal reads the neighbour lists
    InsertSource src;
    src.jobs.resize(items.size());
    for (size_t t = 0; t < items.size(); ++t) {
        InsertJob &j = src.jobs[t];
        j.g = &graph_;
        j.capacity = (int)capacity_;
        j.id = bid[(size_t)items[t]];
        j.level = graph_.level[(size_t)j.id];
        j.efc = p_.max_candidates;
    }
    if (!engine()->run(src, (long long)items.size())) { err = get_dev_error(); return false; }
    if (sel.own.empty()) sel.own.resize((size_t)sel.n);
    for (size_t t = 0; t < items.size(); ++t) {
        sel.own[(size_t)items[t]] = std::move(src.jobs[t].selected);
        sel.has_own[(size_t)items[t]] = 1;
    }
    return true;
}

// Graph-resident traversal: ONE fused kernel launch (descent + per-layer search + heuristic) for
// the whole batch; items the device hands back are redone on the lock-step path.
bool HnswIndex::search_half_device(const std::vector<int> &bid, Selection &sel, std::string &err)
{
    { Tick t(g_pt.sync_graph); if (!sync_graph(err)) return false; }
    Tick t_all(g_pt.search_half);
    g_pt.batches++;
    const int n = (int)bid.size();
    const int top = graph_.top_layer(), ep = graph_.entry;
    // one launch: every item walks all its layers on the device (GraphConnector.cs:172-181)
    std::vector<SearchJob> jobs((size_t)n);
    sel.upper_base.assign((size_t)n, -1);
    int n_upper = 0;
    for (int i = 0; i < n; ++i) {
        const int id = bid[(size_t)i];
        const int l0 = std::min(graph_.level[(size_t)id], top); // :176
        if (l0 > 0) { sel.upper_base[(size_t)i] = n_upper; n_upper += l0; }
        jobs[(size_t)i] = SearchJob{~id, ep, top, l0, sel.upper_base[(size_t)i]}; // FindEntryPoint from the top (:174)
    }
    sel.n_upper = n_upper;
    if (!dev_->insert_search_batch(jobs.data(), n, p_.max_candidates, 2 * p_.max_edges, n_upper, &sel.dev)) { err = get_dev_error(); return false; }
    std::vector<int> again;
    for (int i = 0; i < n; ++i) if (sel.dev.flag[i]) again.push_back(i);
    return search_half_lockstep(bid, again, sel, err); // items the device handed back
}

// ---- link half ----------------------------------------------------------------------------
namespace {
struct LinkGroup {
    int nb, layer;
    std::vector<int> items;
    size_t idx = 0;
};
// currNode.OutEdges[layer] = selected (:192) and the back-edge appends grouped per
// (neighbour, layer), in item order.
template <class Sel>
void collect_groups(Graph &g, const std::vector<int> &bid, const Sel &sel, std::vector<LinkGroup> &groups)
{
    std::unordered_map<uint64_t, size_t> where;
    const int n = (int)bid.size();
    where.reserve((size_t)n * 40);
    const int top = g.top_layer();
    for (int i = 0; i < n; ++i) {
        const int id = bid[(size_t)i];
        for (int layer = std::min(g.level[(size_t)id], top); layer >= 0; --layer) {
            const int *sp; int sc;
            sel.get(i, layer, sp, sc);
            int *l = g.list(id, layer);
            l[0] = sc;
            std::memcpy(l + 1, sp, sizeof(int) * (size_t)sc);
            for (int e = 0; e < sc; ++e) {
                const int nb = sp[e];
                const uint64_t key = ((uint64_t)(uint32_t)nb << 8) | (uint64_t)(uint32_t)layer;
                auto it = where.find(key);
                if (it == where.end()) {
                    where.emplace(key, groups.size());
                    groups.push_back(LinkGroup{nb, layer, {id}, 0});
                } else {
                    groups[it->second].items.push_back(id);
                }
            }
        }
    }
}
} // namespace

bool HnswIndex::link_half_lockstep(const std::vector<int> &bid, const Selection &sel, std::string &err)
{
    if (!refresh_host_lists(err)) return false;
    std::vector<LinkGroup> groups;
    collect_groups(graph_, bid, sel, groups);
    LinkSource links;
    links.jobs.resize(groups.size());
    for (size_t t = 0; t < groups.size(); ++t) {
        LinkJob &lj = links.jobs[t];
        lj.g = &graph_;
        lj.nb = groups[t].nb;
        lj.layer = groups[t].layer;
        lj.items = std::move(groups[t].items);
    }
    if (!links.jobs.empty() && !engine()->run(links, (long long)links.jobs.size())) { err = get_dev_error(); return false; }
    return true;
}

// The whole link half as ONE launch on the HBM mirror (graph_link_kernel): the host only groups
// the back-edge appends per (neighbour, layer) list -- array-indexed for layer 0 -- and copies the
// final lists back into its own graph.  The mirror stays in step, so no re-upload follows.
bool HnswIndex::link_half_device(const std::vector<int> &bid, const Selection &sel, std::string &err)
{
    const int n = (int)bid.size();
    const int M2 = 2 * p_.max_edges, row_stride = 3 + M2, list_stride = graph_.stride0;
    const int top = graph_.top_layer();
    // Everything the link half needs is already on the device when no item was handed back to the
    // host: own lists, grouping of the appends and the appends themselves run there, the host does
    // nothing in between (HNSW_MI355X_LINK_PLAN=0 keeps the host-grouped path below, which is also
    // the one used after a hand-back).
    const bool plan_on_device = diag("link_plan", 1) != 0;
    bool any_own = false;
    for (int i = 0; i < n; ++i) any_own = any_own || sel.has_own[(size_t)i];
    if (plan_on_device && !any_own && n <= (1 << 20)) { // (a larger batch is searched in several launches: host path)
        Tick t(g_pt.link_dev);
        g_pt.rounds++;
        host_lists_stale_ = true;
        if (!dev_->link_batch_planned(n, sel.n_upper, M2)) { err = get_dev_error(); return false; }
        return true;
    }
    if (!dev_->fetch_insert_selections(&sel.dev)) { err = get_dev_error(); return false; } // the host groups: it needs the ids
    // Host-grouped path: the batch is linked in up to four sub-batches of consecutive items.  Appending the items of one
    // adjacency list sub-batch after sub-batch is the same sequence as appending them all in item
    // order, so the outcome is unchanged -- but while the GPU links one sub-batch the host groups
    // the next and files the previous one's lists.
    constexpr int split = 4;
    const int S = n >= 2048 ? split : 1;
    // The lists the link kernel leaves behind stay in the HBM mirror; the host copy is marked stale
    // and fetched back when something on the host needs it (refresh_host_lists).
    host_lists_stale_ = true;
    bool pend[2] = {false, false};
    auto finish = [&](int set) -> bool {
        if (!pend[set]) return true;
        Tick t(g_pt.link_dev);
        if (!dev_->link_batch_finish(set, nullptr)) { err = get_dev_error(); return false; }
        pend[set] = false;
        return true;
    };
    // per-batch work arrays live in the index (their capacity settles after the first full batch)
    std::vector<int> &rows = lk_rows_, &g_node = lk_node_, &g_layer = lk_layer_, &g_cnt = lk_cnt_;
    std::vector<int> &g_off = lk_off_, &g_items = lk_items_, &fill = lk_fill_;
    std::vector<std::pair<int, int>> &seq = lk_seq_; // (group, item id) in append order
    if ((int)grp_of_node0_.size() < graph_.length) grp_of_node0_.resize((size_t)graph_.length, -1);
    for (int s = 0; s < S; ++s) {
        const int set = s & 1;
        if (!finish(set)) return false; // sub-batch s - 2: its staging set is needed again
        const int i0 = (int)((long long)n * s / S), i1 = (int)((long long)n * (s + 1) / S);
        {
            Tick t(g_pt.collect);
            rows.clear(); g_node.clear(); g_layer.clear(); g_cnt.clear(); seq.clear();
            std::unordered_map<uint64_t, int> upper_groups;
            rows.reserve((size_t)(i1 - i0) * row_stride);
            seq.reserve((size_t)(i1 - i0) * M2);
            for (int i = i0; i < i1; ++i) {
                const int id = bid[(size_t)i];
                for (int layer = std::min(graph_.level[(size_t)id], top); layer >= 0; --layer) {
                    const int *sp; int sc;
                    sel.get(i, layer, sp, sc);
                    size_t r0 = rows.size(); // currNode.OutEdges[layer] = selected (:192), written to the mirror by the launch
                    rows.resize(r0 + (size_t)row_stride); // the tail beyond sc ids is never read (link_batch validates [0, sc))
                    rows[r0] = id; rows[r0 + 1] = layer; rows[r0 + 2] = sc;
                    std::memcpy(rows.data() + r0 + 3, sp, sizeof(int) * (size_t)sc);
                    for (int e = 0; e < sc; ++e) {
                        const int nb = sp[e];
                        int gi;
                        if (layer == 0) {
                            gi = grp_of_node0_[(size_t)nb];
                            if (gi < 0) { gi = (int)g_node.size(); grp_of_node0_[(size_t)nb] = gi; g_node.push_back(nb); g_layer.push_back(0); g_cnt.push_back(0); }
                        } else {
                            const uint64_t key = ((uint64_t)(uint32_t)nb << 8) | (uint64_t)(uint32_t)layer;
                            auto it = upper_groups.find(key);
                            if (it == upper_groups.end()) { gi = (int)g_node.size(); upper_groups.emplace(key, gi); g_node.push_back(nb); g_layer.push_back(layer); g_cnt.push_back(0); }
                            else gi = it->second;
                        }
                        g_cnt[(size_t)gi]++;
                        seq.emplace_back(gi, id);
                    }
                }
            }
            for (size_t g = 0; g < g_node.size(); ++g) if (g_layer[g] == 0) grp_of_node0_[(size_t)g_node[g]] = -1;
            const int G = (int)g_node.size();
            g_off.resize((size_t)G + 1);
            g_items.resize(seq.size());
            fill.assign((size_t)G, 0);
            g_off[0] = 0;
            for (int g = 0; g < G; ++g) g_off[(size_t)g + 1] = g_off[(size_t)g] + g_cnt[(size_t)g];
            for (const auto &pr : seq) g_items[(size_t)(g_off[(size_t)pr.first] + fill[(size_t)pr.first]++)] = pr.second;
        }
        {
            Tick t(g_pt.link_dev);
            g_pt.rounds++;
            if (!dev_->link_batch_begin(set, rows.data(), (int)(rows.size() / (size_t)row_stride), row_stride, g_node.data(), g_layer.data(),
                                        g_off.data(), g_items.data(), (int)g_node.size(), M2, list_stride, false)) { err = get_dev_error(); return false; }
        }
        pend[set] = true;
    }
    // in the order they were begun
    const int first = S >= 2 ? (S & 1) : 0;
    if (!finish(first)) return false;
    return finish(first ^ 1);
}

// One snapshot batch: the nodes `bid` (in insertion order) have no edges yet.
bool HnswIndex::insert_batch(const std::vector<int> &bid, std::string &err)
{
    const int n = (int)bid.size();
    Selection sel;
    sel.n = n;
    sel.has_own.assign((size_t)n, 0);
    if (p_.device_traversal && dev_->traversal_fits(p_.max_candidates, true, p_.max_edges)) {
        if (!search_half_device(bid, sel, err)) return false;
        return link_half_device(bid, sel, err); // keeps the HBM mirror in step
    }
    std::vector<int> all((size_t)n);
    for (int i = 0; i < n; ++i) all[(size_t)i] = i;
    if (!search_half_lockstep(bid, all, sel, err)) return false;
    graph_dirty_ = true;
    return link_half_lockstep(bid, sel, err);
}


// ---- reference-exact Add at window speed ------------------------------------------------------
// HNSWIndex.Add(item) one item after the other (HNSWIndex.cs:55-65) is the only Add whose graph the reference
// defines (its own determinism recipe, bindings/__tests__/parameters_test.py:65-68), and it is a chain: item
// j searches the graph that items < j left behind.  But a search depends on nothing except the stored rows
// (immutable) and the adjacency lists it READS: the node a descent pass scans (GraphNavigator.cs:65) and the
// candidates a beam search expands (:152-156); RelativeNeighborPruning reads rows only (Heuristic.cs:23-40).
// And linking item i WRITES only its own lists (which no search on an older graph can reach) and the lists of
// the neighbours it selected (back-edge append + PruneOverflow, GraphConnector.cs:196-214).  So:
//   * W consecutive items search ONE snapshot in one launch, each recording the lists it read;
//   * the items are then taken in order: item j's result is the sequential one iff none of the lists it read
//     has been written since its snapshot (by an item < j) -- it is linked -- otherwise the round ends;
//   * the next round searches again whatever is no longer valid (the frontier item always is, and its search on
//     the exact graph is valid by definition: at least one item per round) plus new items up to W; results that
//     are still valid are kept.
// What counts as a write.  In a grown graph the lists are full and three out of four back-edge appends end in a
// PruneOverflow that turns the new item away and keeps everything else: the list then reads exactly as before
// (same ids, same order -- its entries are a greedy output, ascending and mutually tested), and for every search
// that read it the append never happened.  The launch that searches an item therefore also DRY-RUNS its appends
// (graph_link_dry_sel_kernel, same code as the link kernel, nothing written); the prediction for a list is exact
// while nothing has changed that list since, and is otherwise replaced by "changed".  16 items per round instead
// of 5 on the 1M x 128 graph.
// Every job of a launch is ONE beam search long.  A multi-layer item (one in sixteen) would make its launch twice
// as long, so its upper layers are searched ahead of time (they read upper-layer lists only, which are rarely
// written) as soon as the item comes within kAhead windows of the frontier, and its layer 0 -- entered at the
// upper layers' selected[0] (GraphConnector.cs:179,216) -- is a job of its own in a later round.  Only a frontier
// item that still lacks its upper layers runs all of them in one job.
// Items whose level exceeds the top layer go alone (they move the entry point, which every search reads,
// GraphConnector.cs:27-41), and so do items the device hands back (NaN / -0 distances).  The links of a
// round's prefix are one launch of the batched link half -- appends grouped per list in item order, which is
// the sequential order -- enqueued without waiting: the next round's searches follow in the same stream.  The
// graph is the sequential one by construction; the GPU tests hold it to the CPU restatement's sequential Add.
bool HnswIndex::insert_exact_window(const std::vector<int> &fresh, int &p, int W, bool background, std::string &err)
{
    const int m = (int)fresh.size();
    const int M2 = 2 * p_.max_edges;
    const int log_cap = std::max(2048, 16 * p_.max_candidates); // ints per job: [count, -, (node, far key) pairs...]
    W = std::max(2, W);
    constexpr int kAhead = 4; // upper layers are searched up to kAhead windows ahead of the frontier
    const int ring = W * (1 + kAhead);
    struct Part { // one half of an item's search: the upper layers (with the descent) or layer 0
        bool has = false, overflow = false;
        uint32_t snap = 0; // items linked (seq_) when the search ran
    };
    struct Spec {
        int t = -1;      // position in `fresh` this slot holds
        int l0 = 0;      // min(level, top): layers l0 .. 1 are the upper part
        bool handback = false;
        Part up, lo;     // for a single-layer item `up` is the descent alone, searched together with `lo`
        std::vector<std::vector<int>> sel;   // per layer
        std::vector<std::vector<int>> code;  // per layer and entry, the dry run's verdict on the back-edge append: 0 = the neighbour's
                                             // list reads as before; else bit 0, bit 1 = the item stays in it, bits 8.. = ids it loses (255: unknown)
        std::vector<int> drop;    // layer 0: up to three lost ids per entry
        std::vector<int> r0;      // layer-0 lists read ...
        std::vector<uint32_t> f0; // ... and the key of the farthest result when each was expanded (0xffffffff: result list not full yet)
        std::vector<uint64_t> rU; // upper-layer lists read: layer << 32 | node
        bool repeated = false;    // a layer was answered by the exact two-heap traversal (equal distances): order-sensitive
        int blocker = -1;         // this round: the first item of the window whose linking invalidates this result (-1: none)
        int why = 0;              // (trace) what set the blocker: see PhaseTimers::xw_end
    };
    std::vector<Spec> spec((size_t)ring);
    // Write stamps are compared with snapshots taken inside THIS call only, and every stamp an earlier call left is at most
    // the seq_ this call starts from -- as good as none.  So the upper-layer map starts empty each call (it would otherwise
    // grow by one entry per upper-layer list ever written), and the counter restarts long before it can wrap.
    modU_.clear();
    if (seq_ > 0x7fff0000u) { std::fill(mod0_.begin(), mod0_.end(), 0u); seq_ = 0; }
    if (mod0_.size() < (size_t)graph_.length) mod0_.resize((size_t)graph_.length, 0u);
    auto modU_at = [&](uint64_t k) -> uint32_t { auto it = modU_.find(k); return it == modU_.end() ? 0u : it->second; };
    auto up_valid = [&](const Spec &s, uint32_t cur) {
        if (!s.up.has || s.handback) return false;
        if (s.up.snap == cur) return true;
        if (s.up.overflow) return false;
        for (uint64_t k : s.rU) if (modU_at(k) > s.up.snap) return false;
        return true;
    };
    auto lo_valid = [&](const Spec &s, uint32_t cur) {
        if (!s.lo.has || s.handback) return false;
        if (s.lo.snap == cur) return true;
        if (s.lo.overflow) return false;
        for (int v : s.r0) if (mod0_[(size_t)v] > s.lo.snap) return false;
        return true;
    };
    enum { kFull = 0, kUpper = 1, kLower = 2 };
    struct Todo { int t, kind; };
    std::vector<Todo> todo;
    std::vector<int> bid, upper_base, upper_owner;
    std::vector<SearchJob> jobs;
    const bool dry_on = diag("xw_dry", 1) != 0, stage_on = diag("xw_stage", 1) != 0;
    // links in flight: two staging sets, finished when their set is needed again and before anything returns
    bool pend[2] = {false, false};
    int next_set = 0;
    auto finish = [&](int set) -> bool {
        if (!pend[set]) return true;
        pend[set] = false;
        Tick t(g_pt.link_dev);
        if (!dev_->link_batch_finish(set, nullptr)) { err = get_dev_error(); return false; }
        return true;
    };
    struct Drain { std::function<void()> f; ~Drain() { f(); } } drain{[&] { std::string e2; for (int s2 = 0; s2 < 2; ++s2) if (pend[s2]) { pend[s2] = false; (void)dev_->link_batch_finish(s2, nullptr); } }};
    Tick t_xw(g_pt.xw_total);
    while (p < m) {
        if (graph_.entry < 0) { graph_.entry = fresh[(size_t)p++]; continue; } // GraphConnector.cs:28-33
        const int top = graph_.top_layer();
        double t_ph = g_pt.on ? now_s() : 0;
        auto phase = [&](double &acc) { if (g_pt.on) { const double n = now_s(); acc += n - t_ph; t_ph = n; } };
        {
            Spec &sf = spec[(size_t)(p % ring)];
            if (graph_.level[(size_t)fresh[(size_t)p]] > top || (sf.t == p && sf.handback)) { // alone: entry-point lock (:36-41) / exact host path
                if (!finish(0) || !finish(1)) return false;
                bid.assign(1, fresh[(size_t)p]);
                if (background && !dev_->upload_rows_wait((long long)bid.back() + 1)) { err = get_dev_error(); return false; }
                if (!insert_batch(bid, err)) return false;
                if (graph_.level[(size_t)bid[0]] > top) graph_.entry = bid[0];
                ++p;
                ++seq_;
                for (Spec &s : spec) s.t = -1; // what that insert wrote is not tracked: nothing speculative survives it
                ++xw_alone_;
                continue;
            }
        }
        // the window follows what the graph lets through: searching 64 items a round when 4 of them link (a small graph: every
        // insert reads the same hubs) only lengthens the launch -- twice the recent prefix, within [8, W]; the upper layers
        // searched ahead follow the live window too.  (Measured, same box: at 1M nodes the prefix is 29-30 items whatever W
        // is, and W = 256 does 15.0 k adds/s beside W = 64's 15.4 k -- 13.4 k with three times the prefix and the look-ahead
        // sized by W; at 10M the prefix is 86 and W = 256 does 26.4 k beside W = 64's 21.2 k.)
        constexpr double xw_factor = 2.0; // swept in rounds 3 and 4 (1.5 / 2 / 3 / 4: 17.5 / 18.2 / 17.7 / 16.4 k adds/s at 1M)
        const int W_now = std::min(W, std::max(8, (int)(xw_factor * xw_prefix_ema_) + 4));
        int hi = std::min(m, p + W_now), hi_up = std::min(m, p + std::min(ring, W_now * (1 + kAhead)));
        for (int t = p + 1; t < hi_up; ++t) if (graph_.level[(size_t)fresh[(size_t)t]] > top) { hi_up = t; break; }
        hi = std::min(hi, hi_up);
        if (!stage_on) hi_up = hi;
        const uint32_t R = seq_;
        // ---- what has to be searched (again) ----
        todo.clear();
        for (int t = p; t < hi_up; ++t) {
            Spec &s = spec[(size_t)(t % ring)];
            const int l0 = std::min(graph_.level[(size_t)fresh[(size_t)t]], top); // GraphConnector.cs:176
            if (t >= hi && l0 == 0) continue; // beyond the window only upper layers are searched ahead
            if (s.t != t) { s.t = t; s.l0 = l0; s.handback = false; s.up = Part{}; s.lo = Part{}; }
            if (s.handback) continue;
            const bool uv = up_valid(s, R);
            if (l0 == 0 || !stage_on) { if (!uv || !lo_valid(s, R)) todo.push_back(Todo{t, kFull}); continue; }
            if (!uv) { s.lo.has = false; todo.push_back(Todo{t, t == p ? kFull : kUpper}); continue; } // layer 0 enters where the upper layers end
            if (t < hi && !lo_valid(s, R)) todo.push_back(Todo{t, kLower});
        }
        phase(g_pt.xw_todo);
        if (!todo.empty()) {
            int max_t = 0;
            for (const Todo &d : todo) max_t = std::max(max_t, d.t);
            if (background && !dev_->upload_rows_wait((long long)fresh[(size_t)max_t] + 1)) { err = get_dev_error(); return false; }
            { Tick t(g_pt.sync_graph); if (!sync_graph(err)) return false; }
            Tick t_all(g_pt.search_half);
            const int n = (int)todo.size(), ep = graph_.entry;
            jobs.resize((size_t)n);
            upper_base.assign((size_t)n, -1);
            upper_owner.clear();
            for (int i = 0; i < n; ++i) {
                const Spec &s = spec[(size_t)(todo[(size_t)i].t % ring)];
                const int id = fresh[(size_t)todo[(size_t)i].t], kind = todo[(size_t)i].kind;
                if (kind == kLower) { jobs[(size_t)i] = SearchJob{~id, s.sel[1][0], 0, 0, -1, 0}; continue; } // :179 bestPeer = selected[0] of the layer above
                if (s.l0 > 0) { upper_base[(size_t)i] = (int)upper_owner.size(); upper_owner.insert(upper_owner.end(), (size_t)s.l0, i); }
                jobs[(size_t)i] = SearchJob{~id, ep, top, s.l0, upper_base[(size_t)i], kind == kUpper ? 1 : 0}; // FindEntryPoint from the top (:174)
            }
            const int n_upper = (int)upper_owner.size();
            Device::InsertResults res{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
            Device::WindowExtras win{log_cap, upper_owner.data(), nullptr, nullptr, nullptr};
            if (!dev_->insert_search_batch(jobs.data(), n, p_.max_candidates, M2, n_upper, &res, &win)) { err = get_dev_error(); return false; }
            phase(g_pt.xw_launch);
            xw_searches_ += (uint64_t)n;
            for (int i = 0; i < n; ++i) {
                Spec &s = spec[(size_t)(todo[(size_t)i].t % ring)];
                const int kind = todo[(size_t)i].kind;
                if (res.flag[i] != 0) { s.handback = true; continue; }
                s.sel.resize((size_t)s.l0 + 1);
                s.code.resize((size_t)s.l0 + 1);
                const int lfrom = kind == kLower ? 0 : s.l0, lto = kind == kUpper ? 1 : 0;
                bool bad = false;
                for (int layer = lfrom; layer >= lto; --layer) {
                    const int *ids, *dry; int cnt;
                    if (layer == 0) { ids = res.sel0 + (size_t)i * res.sel_stride; dry = win.dry0 + (size_t)i * res.sel_stride; cnt = res.cnt0[i]; }
                    else { const size_t u = (size_t)(upper_base[(size_t)i] + layer - 1); ids = res.selU + u * res.sel_stride; dry = win.dryU + u * res.sel_stride; cnt = res.cntU[u]; }
                    if (cnt < 1 || cnt > (layer == 0 ? M2 : M2 / 2)) { bad = true; break; }
                    s.sel[(size_t)layer].assign(ids, ids + cnt);
                    s.code[(size_t)layer].resize((size_t)cnt);
                    for (int e = 0; e < cnt; ++e) s.code[(size_t)layer][(size_t)e] = dry_on ? dry[e] : -1; // -1: changed, lost ids unknown
                    if (layer == 0) s.drop.assign(win.drop0 + (size_t)i * res.sel_stride * 3, win.drop0 + ((size_t)i * res.sel_stride + (size_t)cnt) * 3);
                }
                if (kind != kUpper) s.repeated = win.repeated[i] != 0;
                else if (win.repeated[i]) s.repeated = true;
                if (bad) { err = "exact window: the device returned an impossible selection count"; return false; }
                // the read log: upper-layer entries belong to the upper part, layer-0 entries to the lower part
                const int *lg = win.read_log + (size_t)i * log_cap;
                const bool overflow = lg[0] < 0 || lg[0] > (log_cap - 2) / 2;
                if (kind != kLower) { s.rU.clear(); s.up = Part{true, overflow, R}; }
                if (kind != kUpper) { s.r0.clear(); s.f0.clear(); s.lo = Part{true, overflow, R}; }
                if (!overflow) {
                    int layer = 0;
                    for (int e = 0; e < lg[0]; ++e) {
                        const int v = lg[2 + 2 * e];
                        if (v < 0) layer = -v - 1;
                        else if (layer == 0) { s.r0.push_back(v); s.f0.push_back((uint32_t)lg[3 + 2 * e]); }
                        else s.rU.push_back(((uint64_t)(uint32_t)layer << 32) | (uint32_t)v);
                    }
                }
            }
        }
        // ---- a change the reader does not see ----
        // Within this round the items are linked in order, and an item's result stays the sequential one as long as no earlier
        // item of the window changes a list it read IN A WAY IT WOULD HAVE NOTICED.  The dry run says what each back-edge append
        // does to its list: nothing (code 0), or the list gains the item and / or loses up to three ids.  A search that expanded
        // node v when its result list was full with farthest key f pushes a neighbour only if the neighbour's key is below f
        // (GraphNavigator.cs:165), and f never grows: an id v's list GAINS whose key is not below f would have been measured and
        // turned away; an id it LOSES whose key is not below f was turned away when it was measured (or had been seen before), and
        // is turned away again wherever the search meets it later.  Either way every push and every pop is the same (the evaluation
        // count is not) -- provided the search does not depend on the ORDER of a list, which the sorted-list traversal does not
        // (a job answered by the exact two-heap traversal does, and is not given this benefit).  The distances that decides it,
        // reader row against gained / lost row, are one batch of id<->id distances on the device (the kernels' own arithmetic, so
        // the keys compare exactly as they would have inside the search).  Per layer-0 list only the FIRST change of a round is
        // known this way (the dry run saw the list as it was); a second one, or a change whose lost ids are unknown, blocks.
        phase(g_pt.xw_parse);
        int hi_link = p;
        for (; hi_link < hi; ++hi_link) {
            const Spec &s = spec[(size_t)(hi_link % ring)];
            if (s.t != hi_link || !up_valid(s, R) || !lo_valid(s, R)) break;
        }
        struct Change { int t, e; bool known; };
        xw_first_.clear(); xw_second_.clear();
        for (int t = p; t < hi_link; ++t) {
            Spec &s = spec[(size_t)(t % ring)];
            s.blocker = -1;
            const std::vector<int> &sel0 = s.sel[0];
            for (size_t e = 0; e < sel0.size(); ++e) {
                const int v = sel0[e], c = s.code[0][e];
                const bool fresh_list = mod0_[(size_t)v] <= s.lo.snap; // unchanged since the dry run saw it
                auto it = xw_first_.find(v);
                if (it == xw_first_.end()) {
                    if (fresh_list && c == 0) continue; // the append leaves it as it reads
                    xw_first_.emplace(v, XwChange{t, (int)e, fresh_list && c > 0 && ((c >> 8) & 0xff) <= 3});
                } else if (!xw_second_.count(v)) xw_second_.emplace(v, t); // whatever it does, its list is no longer the one the dry run saw
            }
        }
        if (!xw_first_.empty()) {
            pa_.clear(); pb_.clear(); pfar_.clear(); powner_.clear();
            for (int t = p + 1; t < hi_link; ++t) {
                Spec &s = spec[(size_t)(t % ring)];
                auto block = [&](int by, int why) { if (by < t && (s.blocker < 0 || by < s.blocker)) { s.blocker = by; s.why = why; } };
                const int jid = fresh[(size_t)t];
                for (size_t r = 0; r < s.r0.size(); ++r) {
                    auto it = xw_first_.find(s.r0[r]);
                    if (it == xw_first_.end() || it->second.t >= t) continue;
                    const XwChange &c1 = it->second;
                    auto i2 = xw_second_.find(s.r0[r]);
                    if (i2 != xw_second_.end()) block(i2->second, 2);
                    const uint32_t far = s.f0[r];
                    if (!c1.known || s.repeated || far == 0xffffffffu) { block(c1.t, !c1.known ? 3 : s.repeated ? 4 : 5); continue; }
                    const Spec &w = spec[(size_t)(c1.t % ring)];
                    const int code = w.code[0][(size_t)c1.e], nd = (code >> 8) & 0xff;
                    if (code & 2) { pa_.push_back(jid); pb_.push_back(fresh[(size_t)c1.t]); pfar_.push_back(far); powner_.push_back(std::make_pair(t, c1.t)); }
                    for (int d = 0; d < nd; ++d) { pa_.push_back(jid); pb_.push_back(w.drop[(size_t)c1.e * 3 + (size_t)d]); pfar_.push_back(far); powner_.push_back(std::make_pair(t, c1.t)); }
                }
            }
            if (!pa_.empty()) {
                pd_.resize(pa_.size());
                phase(g_pt.xw_valid);
                if (!dev_->dist_pair_batch(pa_.data(), pb_.data(), (int)pa_.size(), pd_.data())) { err = get_dev_error(); return false; }
                phase(g_pt.xw_pairs);
                xw_pairs_ += (uint64_t)pa_.size();
                for (size_t q = 0; q < pa_.size(); ++q) {
                    uint32_t u;
                    std::memcpy(&u, &pd_[q], 4);
                    const bool odd = pd_[q] != pd_[q] || u == 0x80000000u;         // NaN / -0: no key
                    const uint32_t key = (u & 0x80000000u) ? ~u : (u | 0x80000000u); // the kernels' f2key
                    if (odd || key < pfar_[q]) {                                     // it would have been pushed: the change shows
                        Spec &s = spec[(size_t)(powner_[q].first % ring)];
                        if (s.blocker < 0 || powner_[q].second < s.blocker) { s.blocker = powner_[q].second; s.why = 6; }
                    }
                }
            }
        }
        // ---- the valid prefix, in item order ----
        bid.clear();
        uint32_t cur = R;
        int t = p;
        Selection sel;
        for (; t < hi_link; ++t) {
            Spec &s = spec[(size_t)(t % ring)];
            if (s.blocker >= 0) { g_pt.xw_end[s.why]++; break; }
            if (!up_valid(s, cur)) { g_pt.xw_end[7]++; break; }                   // (upper layers: any write since the snapshot counts)
            if (s.lo.overflow && !(s.lo.snap == cur)) { g_pt.xw_end[7]++; break; } // no read log: only good on the very graph it searched
            ++cur;
            for (size_t layer = 0; layer < s.sel.size(); ++layer) {
                const uint32_t snap = layer == 0 ? s.lo.snap : s.up.snap; // the dry run saw the graph of that snapshot
                for (size_t e = 0; e < s.sel[layer].size(); ++e) {
                    const int nb = s.sel[layer][e];
                    const bool same = s.code[layer][e] == 0;
                    if (layer == 0) {
                        if (same && mod0_[(size_t)nb] <= snap) continue; // unchanged since, and the append leaves it as it reads
                        mod0_[(size_t)nb] = cur;
                    } else {
                        const uint64_t key = ((uint64_t)layer << 32) | (uint32_t)nb;
                        if (same && modU_at(key) <= snap) continue;
                        modU_[key] = cur;
                    }
                }
            }
            bid.push_back(fresh[(size_t)t]);
            sel.own.push_back(std::move(s.sel));
            s.t = -1;
        }
        ++xw_rounds_;
        if (t == hi_link) g_pt.xw_end[hi_link < hi ? 1 : 0]++;
        phase(g_pt.xw_valid);
        if (bid.empty()) continue; // the frontier item was handed back: the next iteration takes it alone
        sel.n = (int)bid.size();
        if (!finish(next_set)) return false;
        if (!link_prefix_begin(bid, sel, next_set, err)) return false;
        pend[next_set] = true;
        next_set ^= 1;
        xw_linked_ += (uint64_t)bid.size();
        xw_prefix_ema_ = 0.75 * xw_prefix_ema_ + 0.25 * (double)bid.size();
        seq_ = cur;
        p = t;
        phase(g_pt.xw_link);
    }
    return finish(0) && finish(1);
}

// The link half of one exact-window prefix (items `bid` in order, their selections in sel.own): own lists and the
// back-edge appends grouped per (neighbour, layer) list in item order, enqueued on staging set `set` without waiting.
bool HnswIndex::link_prefix_begin(const std::vector<int> &bid, const Selection &sel, int set, std::string &err)
{
    const int n = (int)bid.size();
    const int M2 = 2 * p_.max_edges, row_stride = 3 + M2, list_stride = graph_.stride0;
    host_lists_stale_ = true;
    std::vector<int> &rows = lk_rows_, &g_node = lk_node_, &g_layer = lk_layer_, &g_cnt = lk_cnt_;
    std::vector<int> &g_off = lk_off_, &g_items = lk_items_, &fill = lk_fill_;
    std::vector<std::pair<int, int>> &seq = lk_seq_; // (group, item id) in append order
    if ((int)grp_of_node0_.size() < graph_.length) grp_of_node0_.resize((size_t)graph_.length, -1);
    {
        Tick t(g_pt.collect);
        rows.clear(); g_node.clear(); g_layer.clear(); g_cnt.clear(); seq.clear();
        std::unordered_map<uint64_t, int> upper_groups;
        for (int i = 0; i < n; ++i) {
            const int id = bid[(size_t)i];
            const std::vector<std::vector<int>> &own = sel.own[(size_t)i];
            for (int layer = (int)own.size() - 1; layer >= 0; --layer) {
                const std::vector<int> &sp = own[(size_t)layer];
                const int sc = (int)sp.size();
                const size_t r0 = rows.size(); // currNode.OutEdges[layer] = selected (GraphConnector.cs:192)
                rows.resize(r0 + (size_t)row_stride);
                rows[r0] = id; rows[r0 + 1] = layer; rows[r0 + 2] = sc;
                std::memcpy(rows.data() + r0 + 3, sp.data(), sizeof(int) * (size_t)sc);
                for (int e = 0; e < sc; ++e) { // neighbor.OutEdges[layer].Add(currNode.Id) (:207), grouped per list
                    const int nb = sp[(size_t)e];
                    int gi;
                    if (layer == 0) {
                        gi = grp_of_node0_[(size_t)nb];
                        if (gi < 0) { gi = (int)g_node.size(); grp_of_node0_[(size_t)nb] = gi; g_node.push_back(nb); g_layer.push_back(0); g_cnt.push_back(0); }
                    } else {
                        const uint64_t key = ((uint64_t)(uint32_t)nb << 8) | (uint64_t)(uint32_t)layer;
                        auto it = upper_groups.find(key);
                        if (it == upper_groups.end()) { gi = (int)g_node.size(); upper_groups.emplace(key, gi); g_node.push_back(nb); g_layer.push_back(layer); g_cnt.push_back(0); }
                        else gi = it->second;
                    }
                    g_cnt[(size_t)gi]++;
                    seq.emplace_back(gi, id);
                }
            }
        }
        for (size_t g = 0; g < g_node.size(); ++g) if (g_layer[g] == 0) grp_of_node0_[(size_t)g_node[g]] = -1;
        const int G = (int)g_node.size();
        g_off.resize((size_t)G + 1);
        g_items.resize(seq.size());
        fill.assign((size_t)G, 0);
        g_off[0] = 0;
        for (int g = 0; g < G; ++g) g_off[(size_t)g + 1] = g_off[(size_t)g] + g_cnt[(size_t)g];
        for (const auto &pr : seq) g_items[(size_t)(g_off[(size_t)pr.first] + fill[(size_t)pr.first]++)] = pr.second;
    }
    Tick t(g_pt.link_dev);
    g_pt.rounds++;
    if (!dev_->link_batch_begin(set, rows.data(), (int)(rows.size() / (size_t)row_stride), row_stride, g_node.data(), g_layer.data(),
                                g_off.data(), g_items.data(), (int)g_node.size(), M2, list_stride, false)) { err = get_dev_error(); return false; }
    return true;
}

int HnswIndex::add(const float *vectors, int count, int dim, int *out_ids, std::string &err)
{
    if (failed(err)) return -1;
    in_valid_ = false;
    ++graph_epoch_;
    if (!ensure_dim(dim, err)) return -1;
    Tick t_total(g_pt.add_total);
    double t_nodes0 = g_pt.on ? now_s() : 0;
    // GraphData.AddItem (src/HNSWIndex/GraphData.cs:79-118): one RNG draw per item, in order;
    // vacated slots are reused first when removals are allowed (:85-91).
    // Everything that can be refused is checked BEFORE the index changes: the levels are drawn on a
    // copy of the generator and validated, and the capacity for the new slots is reserved; only then
    // are the generator advanced and the nodes appended.  A device failure after that point leaves
    // nodes without rows or links, so it marks the index failed (every later call reports it).
    if (p_.allow_removals && !graph_.removed_stack.empty() && !refresh_host_lists(err)) return -1; // slot reuse rewrites host rows
    std::vector<int> lvls((size_t)count);
    DotnetRandom rng = rng_;
    int live = 0;
    for (int i = 0; i < count; ++i) {
        const int lvl = level_from_uniform(rng.next_single(), p_.distribution_rate);
        if (lvl > 200) { err = "level draw out of range"; return -1; }
        lvls[(size_t)i] = lvl;
        live += lvl >= 0;
    }
    const int reusable = p_.allow_removals ? (int)graph_.removed_stack.size() : 0;
    if (!ensure_capacity((long long)graph_.length + std::max(0, live - reusable), err)) return -1;
    rng_ = rng;
    std::vector<int> ids((size_t)count), fresh; // fresh: the new nodes, in insertion order
    fresh.reserve((size_t)count);
    bool any_reused = false;
    const int first_new = graph_.length;
    graph_.grow_for(live - std::min(live, reusable)); // one resize of the per-node arrays, not one per node
    for (int i = 0; i < count; ++i) {
        if (lvls[(size_t)i] < 0) { ids[(size_t)i] = -1; ++skipped_; continue; } // :82
        ids[(size_t)i] = graph_.add_node(lvls[(size_t)i], p_.allow_removals, &any_reused);
        fresh.push_back(ids[(size_t)i]);
    }
    if (g_pt.on) { g_pt.add_nodes += now_s() - t_nodes0; t_nodes0 = now_s(); }
    if (any_reused) { graph_dirty_ = true; replicas_.clear(); } // existing rows of the HBM mirror changed: full re-upload, replicas from scratch
    // rows -> HBM (id == row index)
    // A large Add starts linking as soon as its first rows are resident: the rest is uploaded by a helper
    // thread on its own stream while the first (small, latency-bound) batches run, and every batch waits
    // only for the rows it touches.
    constexpr int kUploadAhead = 65536;
    bool background = false;
    if (!any_reused && (int)fresh.size() == count) {
        if (count >= 4 * kUploadAhead && metric_ != HNSWDEV_SQ_EUCLID_I8) {
            if (!dev_->upload_rows(first_new, kUploadAhead, vectors)) return fail(get_dev_error(), err);
            if (!dev_->upload_rows_begin(first_new + kUploadAhead, count - kUploadAhead, vectors + (size_t)kUploadAhead * dim)) return fail(get_dev_error(), err);
            background = true;
        } else if (!dev_->upload_rows(first_new, count, vectors)) return fail(get_dev_error(), err);
    } else {
        for (int i = 0; i < count; ++i)
            if (ids[(size_t)i] >= 0 && !dev_->upload_rows(ids[(size_t)i], 1, vectors + (size_t)i * dim)) return fail(get_dev_error(), err);
    }
    if (g_pt.on) g_pt.add_upload += now_s() - t_nodes0;
    // GraphConnector.ConnectNewNode (:24-47), batched
    const int m = (int)fresh.size();
    const int bmax = std::max(1, insert_batch_cap()); // a negative cap selects the exact window below (1 where that cannot run)
    int p = 0;
    std::vector<int> bid;
    if (p_.insert_batch < 0 && m >= 2 && !any_reused && p_.device_traversal && dev_->traversal_fits(p_.max_candidates, true, p_.max_edges)) {
        if (!insert_exact_window(fresh, p, std::min(-p_.insert_batch, 4096), background, err)) { if (background) (void)dev_->upload_rows_wait(-1); return fail(err, err); }
    }
    while (p < m) {
        if (graph_.entry < 0) { graph_.entry = fresh[(size_t)p++]; continue; } // :28-33
        const int top = graph_.top_layer();
        bid.clear();
        bid.push_back(fresh[(size_t)p]);
        if (graph_.level[(size_t)fresh[(size_t)p]] > top) { // new entry point: alone, under the "entry point lock" (:36-41)
            if (background && !dev_->upload_rows_wait((long long)bid.back() + 1)) return fail(get_dev_error(), err);
            if (!insert_batch(bid, err)) { if (background) (void)dev_->upload_rows_wait(-1); return fail(err, err); }
            graph_.entry = fresh[(size_t)p++];
            continue;
        }
        const int linked = graph_.count - (m - p); // nodes already linked (== the id when nothing was ever removed)
        const int early_limit = std::min(kEarlyLinked, graph_.count / kBatchGrowthDiv); // graph_.count: nodes once this call is done
        const int div_now = linked < early_limit ? kEarlyGrowthDiv : kBatchGrowthDiv;
        const int b = std::min(bmax, std::max(1, linked / div_now));
        while ((int)bid.size() < b && p + (int)bid.size() < m && graph_.level[(size_t)fresh[(size_t)(p + (int)bid.size())]] <= top)
            bid.push_back(fresh[(size_t)(p + (int)bid.size())]);
        if (background && !dev_->upload_rows_wait((long long)bid.back() + 1)) return fail(get_dev_error(), err); // ids ascend within an Add
        if (!insert_batch(bid, err)) { if (background) (void)dev_->upload_rows_wait(-1); return fail(err, err); }
        p += (int)bid.size();
    }
    if (background && !dev_->upload_rows_wait(-1)) return fail(get_dev_error(), err); // `vectors` is borrowed only for this call
    if (out_ids) for (int i = 0; i < count; ++i) out_ids[i] = ids[(size_t)i];
    return count;
}

bool HnswIndex::sync_graph(std::string &err)
{
    if (!graph_dirty_) { // incremental: nodes appended on the host since the last sync (their lists are empty)
        bool full = false;
        // Device tracks how many nodes it mirrors; anything beyond is new
        if (!dev_->graph_append_nodes(dev_->graph_nodes(), graph_.length - dev_->graph_nodes(), graph_.level.data(), graph_.upper.data(),
                                      graph_.pool.data(), dev_pool_len_, (long long)graph_.pool.size(), &full)) {
            err = get_dev_error();
            return false;
        }
        if (!full) { dev_pool_len_ = (long long)graph_.pool.size(); return true; }
    }
    if (!refresh_host_lists(err)) return false; // a full upload sends the host copy: it must be current
    if (!dev_->set_graph(graph_.adj0.data(), graph_.length, graph_.stride0, graph_.level.data(), graph_.upper.data(),
                         graph_.pool.data(), (long long)graph_.pool.size(), graph_.strideU)) {
        err = get_dev_error();
        return false;
    }
    dev_pool_len_ = (long long)graph_.pool.size();
    graph_dirty_ = false;
    return true;
}

// Host lock-step traversal for the queries listed in `which` (nullptr: all `count` queries).
int HnswIndex::knn_query_lockstep(const int *which, int count, int k, int *out_ids, float *out_dists, std::string &err)
{
    if (!refresh_host_lists(err)) return -1;
    QuerySource src;
    src.jobs.resize((size_t)count);
    const int ef = std::max(p_.min_nn, k); // HNSWIndex.cs:115
    for (int i = 0; i < count; ++i) {
        const int qi = which ? which[i] : i;
        QueryJob &j = src.jobs[(size_t)i];
        j.g = &graph_;
        j.capacity = (int)capacity_;
        j.qi = qi;
        j.ef = ef;
        j.k = k;
        j.out_ids = out_ids + (size_t)qi * k;
        j.out_d = out_dists + (size_t)qi * k;
    }
    if (!engine()->run(src, count)) { err = get_dev_error(); return -1; }
    return 0;
}

// Graph-resident traversal: one kernel launch runs every query's FindEntryPointQuery +
// SearchLayerQuery + stable OrderBy/Take(k) (HNSWIndex.cs:116-123).
int HnswIndex::knn_query_device(const float *, int count, int k, int *out_ids, float *out_dists, std::string &err)
{
    if (!sync_graph(err)) return -1;
    const int ef = std::max(p_.min_nn, k);
    const int ep = graph_.entry, top = graph_.top_layer();
    std::vector<int> flag((size_t)count);
    { Tick t(g_pt.query_dev);
    if (!dev_->search_queries(count, ep, top, ef, k, out_ids, out_dists, flag.data())) { err = get_dev_error(); return -1; } }
    Tick t_post(g_pt.post);
    std::vector<int> redo;
    for (int i = 0; i < count; ++i) if (flag[(size_t)i]) redo.push_back(i);
    if (!redo.empty()) // candidate heap outgrew LDS: exact re-run on the lock-step path
        return knn_query_lockstep(redo.data(), (int)redo.size(), k, out_ids, out_dists, err);
    return 0;
}

int HnswIndex::knn_query(const float *queries, int count, int dim, int k, int *out_ids, float *out_dists, std::string &err)
{
    if (count <= 0) return 0;
    if (failed(err)) return -1;
    if (k < 1 || graph_.entry < 0 || graph_.count <= 0) { // HNSWIndex.cs:109: empty result lists, padded by the export
        for (long long j = 0; j < (long long)count * std::max(k, 0); ++j) { out_ids[j] = -1; out_dists[j] = std::numeric_limits<float>::quiet_NaN(); }
        return 0;
    }
    // the queries arrive as host buffers: where the traversal runs on the device, all but the first rows are uploaded
    // BEHIND the launch (Device::set_queries_streamed), which starts as soon as those first rows are resident
    const bool streamed = p_.device_traversal && dim == dim_ && dev_ && dev_->traversal_fits(std::max(p_.min_nn, k), false, p_.max_edges);
    if (set_resident_queries(queries, count, dim, err, streamed) < 0) return -1;
    const int rc = knn_query_resident(k, out_ids, out_dists, err);
    for (int g = 0; g < (sharded_resident_ ? p_.devices : 1); ++g) context(g)->cancel_streamed(); // (only after an error: `queries` is borrowed for this call)
    if (rc < 0) resident_queries_ = 0; // a failed call leaves no resident set: part of it may never have been uploaded
    return rc;
}

// The device contexts 1 .. devices - 1 (created on first use; more contexts than GPUs share them round robin --
// a rehearsal of the sharded path on a box with fewer GPUs), and with `clone` their rows + graph mirror brought up
// to date from the primary, every replica copying on its own stream at the same time.
bool HnswIndex::ensure_replicas(bool clone, std::string &err)
{
    const int n = p_.devices;
    if ((int)replicas_.size() != n - 1) {
        replicas_.clear();
        const int ndev = std::max(1, hnswdev_device_count());
        for (int g = 1; g < n; ++g) {
            Device *d = Device::create((device_ordinal_ + g) % ndev, dim_, metric_, capacity_);
            if (!d) { err = get_dev_error(); replicas_.clear(); return false; }
            d->set_profiling(profiling_);
            replicas_.emplace_back(d);
        }
        replica_epoch_.assign((size_t)std::max(0, n - 1), 0);
    }
    if (!clone) return true;
    std::vector<int> stale;
    for (int g = 1; g < n; ++g) if (replica_epoch_[(size_t)g - 1] != graph_epoch_) stale.push_back(g);
    if (stale.empty()) return true;
    std::vector<std::string> errs(stale.size());
    std::vector<std::thread> th;
    for (size_t i = 0; i < stale.size(); ++i)
        th.emplace_back([&, i] { if (!replicas_[(size_t)stale[i] - 1]->clone_from(dev_.get(), dev_pool_len_)) errs[i] = get_dev_error().empty() ? "replica copy failed" : get_dev_error(); });
    for (auto &t : th) t.join();
    for (size_t i = 0; i < stale.size(); ++i) {
        if (!errs[i].empty()) { err = errs[i]; return false; }
        replica_epoch_[(size_t)stale[i] - 1] = graph_epoch_;
    }
    return true;
}

// a call of at least kStreamMin queries starts on its first kStreamHead rows; see Device::set_queries_streamed
static constexpr int kStreamMin = 32768, kStreamHead = 4096;

// hnsw_knn_query from several host threads at once (lock held shared): one query lane per call.
int HnswIndex::knn_query_concurrent(const float *queries, int count, int dim, int k, int *out_ids, float *out_dists, int &rc, std::string &err)
{
    const bool enabled = diag("concurrent_queries", 1) != 0;
    if (!enabled || !failed_msg_.empty() || p_.devices > 1 || !p_.device_traversal || !dev_ || dim != dim_ || count <= 0 || k < 1 ||
        graph_.entry < 0 || graph_.count <= 0 || graph_dirty_ || dev_->graph_nodes() != graph_.length)
        return 0;
    const int ef = std::max(p_.min_nn, k);
    if (!dev_->traversal_fits(ef, false, p_.max_edges)) return 0;
    // A call large enough to fill the chip on its own takes the exclusive path, where its query rows are uploaded BEHIND the
    // launch (Device::set_queries_streamed): a kernel that sleeps on rows still arriving must have the GPU's queues to
    // itself -- two such launches on two lanes were measured to starve each other's copies -- and overlapping it with
    // another call would gain little.  The lanes serve the smaller calls, which leave the chip part-idle.
    if (count >= kStreamMin) return 0;
    int lane = -1;
    bool shares_chip = false;
    {
        std::unique_lock<std::mutex> lk(lane_mu_);
        lane_cv_.wait(lk, [&] { return !lane_busy_[0] || !lane_busy_[1]; });
        lane = lane_busy_[0] ? 1 : 0; // lane 0 is the primary context itself (a lone caller runs exactly where the exclusive path would)
        shares_chip = lane_busy_[1 - lane];
        lane_busy_[lane] = true;
        if (lane > 0 && !lanes_[lane]) {
            lanes_[lane].reset(Device::create_view(dev_.get()));
            if (lanes_[lane]) lanes_[lane]->set_profiling(profiling_);
        }
    }
    struct Release { HnswIndex *ix; int lane; ~Release() { { std::lock_guard<std::mutex> lk(ix->lane_mu_); ix->lane_busy_[lane] = false; } ix->lane_cv_.notify_one(); } } release{this, lane};
    Device *d = lane == 0 ? dev_.get() : lanes_[lane].get();
    if (!d) return 0;
    if (lane > 0) d->rebind(dev_.get()); // no writer is active: the primary's arrays are stable while this call runs
    rc = -1;
    if (lane == 0) { resident_queries_ = 0; sharded_resident_ = false; } // hnsw_knn_query leaves its own queries resident on the primary
    if (!d->set_queries(queries, count)) { err = get_dev_error(); return 1; }
    if (lane == 0) resident_queries_ = count;
    const int ep = graph_.entry, top = graph_.top_layer();
    std::vector<int> flag((size_t)count);
    // the waves a draining launch keeps behind as shadows (graph_search_kernel) are waves the other lane's launch is waiting
    // for: a call that starts while the other lane is busy leaves none
    d->set_shadows_allowed(!shares_chip);
    const bool ok = d->search_queries(count, ep, top, ef, k, out_ids, out_dists, flag.data());
    d->set_shadows_allowed(true);
    if (!ok) { err = get_dev_error(); return 1; }
    for (int i = 0; i < count; ++i) if (flag[(size_t)i]) return 0; // something was handed back: the exclusive path answers the whole call
    rc = 0;
    return 1;
}

void HnswIndex::collect_stats(hnswdev_stats *out)
{
    std::memset(out, 0, sizeof(*out));
    if (!dev_) return;
    dev_->get_stats(out);
    for (auto &l : lanes_) {
        if (!l) continue;
        hnswdev_stats s;
        l->get_stats(&s);
        out->launches += s.launches; out->evals += s.evals; out->timed_launches += s.timed_launches; out->timed_evals += s.timed_evals; out->kernel_ms += s.kernel_ms;
        out->search_launches += s.search_launches; out->search_evals += s.search_evals; out->search_timed_launches += s.search_timed_launches;
        out->search_timed_evals += s.search_timed_evals; out->search_kernel_ms += s.search_kernel_ms; out->search_overflows += s.search_overflows;
        out->search_repeats += s.search_repeats; out->visited_hash_launches += s.visited_hash_launches;
        out->tie_windows += s.tie_windows; out->lat_launches += s.lat_launches; out->lean_launches += s.lean_launches; out->peer_direct_copies += s.peer_direct_copies; out->peer_staged_copies += s.peer_staged_copies; out->insert_tie_reruns += s.insert_tie_reruns; out->range_device_ordered += s.range_device_ordered; out->range_host_ordered += s.range_host_ordered;
    }
}

void HnswIndex::reset_all_stats()
{
    if (dev_) dev_->reset_stats();
    for (auto &r : replicas_) r->reset_stats();
    for (auto &l : lanes_) if (l) l->reset_stats();
}

int HnswIndex::set_resident_queries(const float *queries, int count, int dim, std::string &err, bool streamed)
{
    const bool stream_on = diag("stream_queries", 1) != 0;
    streamed = streamed && stream_on;
    if (!ensure_dim(dim, err)) return -1;
    Tick t(g_pt.set_queries);
    sharded_resident_ = false;
    const int n = p_.devices;
    if (n > 1 && p_.device_traversal && count >= n) { // every context takes its shard (uploaded side by side)
        if (!ensure_replicas(false, err)) return -1;
        shard_lo_.assign((size_t)n + 1, 0);
        for (int g = 0; g <= n; ++g) shard_lo_[(size_t)g] = (long long)count * g / n;
        std::vector<std::string> errs((size_t)n);
        std::vector<std::thread> th;
        for (int g = 0; g < n; ++g)
            th.emplace_back([&, g] {
                const long long lo = shard_lo_[(size_t)g], hi = shard_lo_[(size_t)g + 1];
                const float *qs = queries + (size_t)lo * (size_t)dim;
                const int cnt = (int)(hi - lo);
                if (!(streamed && cnt >= kStreamMin ? context(g)->set_queries_streamed(qs, cnt, kStreamHead) : context(g)->set_queries(qs, cnt)))
                    errs[(size_t)g] = get_dev_error().empty() ? "set_queries failed" : get_dev_error();
            });
        for (auto &t2 : th) t2.join();
        for (const std::string &e : errs) if (!e.empty()) { err = e; resident_queries_ = 0; return -1; }
        sharded_resident_ = true;
        resident_queries_ = count;
        return 0;
    }
    if (!(streamed && count >= kStreamMin ? dev_->set_queries_streamed(queries, count, kStreamHead) : dev_->set_queries(queries, count))) { err = get_dev_error(); return -1; }
    resident_queries_ = count;
    return 0;
}

int HnswIndex::knn_query_resident(int k, int *out_ids, float *out_dists, std::string &err)
{
    const int count = resident_queries_;
    if (count <= 0) return 0;
    if (failed(err)) return -1;
    if (k < 1 || graph_.entry < 0 || graph_.count <= 0) {
        for (long long j = 0; j < (long long)count * std::max(k, 0); ++j) { out_ids[j] = -1; out_dists[j] = std::numeric_limits<float>::quiet_NaN(); }
        return 0;
    }
    const bool fits = p_.device_traversal && dev_->traversal_fits(std::max(p_.min_nn, k), false, p_.max_edges);
    if (sharded_resident_) {
        if (fits) return knn_query_sharded(k, out_ids, out_dists, err);
        // the host traversal runs on the primary alone: it needs the whole set there
        for (int g = 1; g < p_.devices; ++g)
            if (!dev_->adopt_queries(context(g), 0, shard_lo_[(size_t)g + 1] - shard_lo_[(size_t)g], shard_lo_[(size_t)g], count)) { err = get_dev_error(); return -1; }
        sharded_resident_ = false;
    }
    if (fits) return knn_query_device(nullptr, count, k, out_ids, out_dists, err);
    return knn_query_lockstep(nullptr, count, k, out_ids, out_dists, err);
}

// One traversal launch per context, side by side: context g answers its shard of the resident queries on its replica
// and writes rows [lo_g, hi_g) of the caller's arrays.  Each query's traversal is what the single-device path runs, so
// the answer is bit for bit the same.  Whatever a kernel hands back is answered by the exact host traversal on the primary.
int HnswIndex::knn_query_sharded(int k, int *out_ids, float *out_dists, std::string &err)
{
    if (!sync_graph(err)) return -1;
    if (!ensure_replicas(true, err)) return -1;
    const int n = p_.devices, count = resident_queries_;
    const int ef = std::max(p_.min_nn, k);
    const int ep = graph_.entry, top = graph_.top_layer();
    std::vector<int> flag((size_t)count);
    std::vector<std::string> errs((size_t)n);
    std::vector<std::thread> th;
    { Tick t(g_pt.query_dev);
    for (int g = 0; g < n; ++g)
        th.emplace_back([&, g] {
            const long long lo = shard_lo_[(size_t)g], hi = shard_lo_[(size_t)g + 1];
            const int cnt = (int)(hi - lo);
            if (cnt > 0 && !context(g)->search_queries(cnt, ep, top, ef, k, out_ids + (size_t)lo * k, out_dists + (size_t)lo * k, flag.data() + lo))
                errs[(size_t)g] = get_dev_error().empty() ? "search_batch failed" : get_dev_error();
        });
    for (auto &t2 : th) t2.join(); }
    for (const std::string &e : errs) if (!e.empty()) { err = e; return -1; }
    std::vector<int> redo;
    for (int i = 0; i < count; ++i) if (flag[(size_t)i]) redo.push_back(i);
    if (redo.empty()) return 0;
    for (int g = 1; g < n; ++g) // the exact host traversal names queries by their global index on the primary
        if (!dev_->adopt_queries(context(g), 0, shard_lo_[(size_t)g + 1] - shard_lo_[(size_t)g], shard_lo_[(size_t)g], count)) { err = get_dev_error(); return -1; }
    return knn_query_lockstep(redo.data(), (int)redo.size(), k, out_ids, out_dists, err);
}

// Host lock-step range search for the queries listed in `which` (nullptr: all `count` queries).
int HnswIndex::range_query_lockstep(const int *which, int count, float range, std::vector<std::vector<NodeDist>> &out, std::string &err)
{
    if (!refresh_host_lists(err)) return -1;
    RangeSource src;
    src.jobs.resize((size_t)count);
    for (int i = 0; i < count; ++i) {
        const int qi = which ? which[i] : i;
        RangeJob &j = src.jobs[(size_t)i];
        j.g = &graph_;
        j.capacity = (int)capacity_;
        j.qi = qi;
        j.range = range;
        j.out = &out[(size_t)qi];
    }
    if (!engine()->run(src, count)) { err = get_dev_error(); return -1; }
    return 0;
}

// Graph-resident range search (graph_range_kernel): the result SET of SearchLayerRange does not depend on the
// order its heaps pop in; the reference's stable OrderBy (HNSWIndex.cs:155) does only between results of equal
// distance, and for a query holding such a pair the heaps are replayed on the host from the known distances.
int HnswIndex::range_query_device(int count, float range, std::vector<std::vector<NodeDist>> &out, std::string &err)
{
    if (!sync_graph(err)) return -1;
    std::vector<SearchJob> jobs((size_t)count);
    const int ep = graph_.entry, top = graph_.top_layer();
    for (int i = 0; i < count; ++i) jobs[(size_t)i] = SearchJob{i, ep, top, 0, -1};
    Device::RangeResults r;
    g_pt.rq_queries += count;
    { Tick t(g_pt.rq_batch); if (!dev_->range_batch(jobs.data(), count, range, &r)) { err = get_dev_error(); return -1; } }
    // host threads over the queries: sort each result list; a list holding two equal distances is replayed instead
    auto parallel_for = [&](size_t n, size_t grain, const std::function<void(size_t, size_t)> &body) {
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t lo; (lo = next.fetch_add(grain, std::memory_order_relaxed)) < n;) body(lo, std::min(n, lo + grain));
        };
        const int nth = (int)std::min<size_t>((size_t)std::max(1, threads_), (n + grain - 1) / grain);
        std::vector<std::thread> pool;
        for (int t = 1; t < nth; ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
    };
    std::vector<unsigned char> state((size_t)count, 0); // 1: hand-back (lock-step), 2: equal distances (replay)
    double t_sort0 = g_pt.on ? now_s() : 0;
    parallel_for((size_t)count, 256, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            if (r.flag[i]) { state[i] = 1; continue; }
            SearchHit *b = r.found + r.off[i], *e = b + r.cnt[i];
            bool tie = r.state[i] == kRangeTied; // ascending, equal distances in it, and the device did not replay it
            if (r.state[i] == kRangeHostSort) {  // beyond the device ranking's reach (length, a -0 distance): as until round 5
                std::sort(b, e, [](const SearchHit &x, const SearchHit &y) { return x.dist < y.dist; }); // no NaN: d <= range held
                for (SearchHit *p = b; p + 1 < e; ++p) tie |= p[0].dist == p[1].dist; // also -0 next to +0
            }
            if (tie) { state[i] = 2; continue; }
            std::vector<NodeDist> &o = out[i];
            o.resize((size_t)(e - b));
            for (size_t a = 0; a < o.size(); ++a) o[a] = NodeDist{b[a].id, b[a].dist};
        }
    });
    if (g_pt.on) g_pt.rq_sort += now_s() - t_sort0;
    std::vector<int> redo, replay;
    for (int i = 0; i < count; ++i) {
        if (state[(size_t)i] == 1) redo.push_back(i);
        else if (state[(size_t)i] == 2) replay.push_back(i);
    }
    g_pt.rq_replayed += (long)replay.size();
    if (!replay.empty()) {
        { Tick t(g_pt.rq_refresh); if (!refresh_host_lists(err)) return -1; } // the replay walks the host's copy of the lists
        Tick t_rep(g_pt.rq_replay);
        parallel_for(replay.size(), 1, [&](size_t lo, size_t hi) {
            for (size_t t = lo; t < hi; ++t) {
                const int i = replay[t];
                replay_range_heaps([&](int id) { return graph_.list(id, 0); }, graph_.max_edges_at(0), r.entry[(size_t)i], range,
                                   r.found + r.off[(size_t)i], r.cnt[(size_t)i], out[(size_t)i]);
            }
        });
    }
    if (!redo.empty()) return range_query_lockstep(redo.data(), (int)redo.size(), range, out, err);
    return 0;
}

int HnswIndex::range_query(const float *queries, int count, int dim, float range, std::vector<std::vector<NodeDist>> &out, std::string &err)
{
    out.assign((size_t)std::max(count, 0), {});
    if (failed(err)) return -1;
    if (count <= 0 || graph_.entry < 0) return 0; // HNSWIndex.cs:146
    if (!ensure_dim(dim, err)) return -1;
    resident_queries_ = 0; // the resident set is replaced: a later knn_query_resident must not answer for these rows
    if (!dev_->set_queries(queries, count)) { err = get_dev_error(); return -1; }
    if (p_.device_traversal && dev_->traversal_fits(1, false, p_.max_edges)) return range_query_device(count, range, out, err);
    return range_query_lockstep(nullptr, count, range, out, err);
}

// ---- HNSWIndex.Remove (src/HNSWIndex/HNSWIndex.cs:83-102) ------------------------------------
// The reference keeps ordered in-edge lists (Node.InEdges); only their CONTENT matters for the
// out-graph (each affected node is re-linked from its own state and the shared candidate list,
// GraphConnector.cs:100-165), so the in-edge sets are rebuilt here by transposing the out-lists
// once per call and kept current with the deltas of each step.  Ids are removed in order.
int HnswIndex::remove(const int *ids, int count, std::string &err)
{
    if (!p_.allow_removals) { err = "System.InvalidOperationException: Removals are disabled in this index instance."; return -1; } // :85-86
    if (count <= 0) return 0;
    if (failed(err)) return -1;
    ++graph_epoch_;
    if (!refresh_host_lists(err)) return -1;
    {
        // duplicates: a bitmap over the slots only when the call is large enough to pay for zeroing it (hnsw_remove's usual
        // call names ONE id; a 10M-node index would allocate and clear 10 MB for it), a sorted copy otherwise
        const bool bitmap = (long long)count * 64 > (long long)graph_.length;
        std::vector<unsigned char> listed(bitmap ? (size_t)graph_.length : 0, 0);
        for (int t = 0; t < count; ++t) {
            const int id = ids[t];
            if (id < 0 || id >= graph_.length || graph_.removed[(size_t)id]) { err = "System.IndexOutOfRangeException: hnsw_remove: id " + std::to_string(id) + " is not in the index"; return -1; }
            if (bitmap) {
                if (listed[(size_t)id]) { err = "System.ArgumentException: hnsw_remove: duplicate id " + std::to_string(id); return -1; }
                listed[(size_t)id] = 1;
            }
        }
        if (!bitmap && count > 1) {
            std::vector<int> sorted(ids, ids + count);
            std::sort(sorted.begin(), sorted.end());
            const auto dup = std::adjacent_find(sorted.begin(), sorted.end());
            if (dup != sorted.end()) { err = "System.ArgumentException: hnsw_remove: duplicate id " + std::to_string(*dup); return -1; }
        }
    }
    Graph &g = graph_;
    // in-edge sets: layer 0 by node, upper layers by (node, layer)
    std::vector<std::vector<int>> &in0 = in0_;
    std::unordered_map<uint64_t, std::vector<int>> &inU = inU_;
    auto in_of = [&](int node, int layer) -> std::vector<int> & {
        return layer == 0 ? in0[(size_t)node] : inU[((uint64_t)(uint32_t)node << 8) | (uint64_t)(uint32_t)layer];
    };
    if (!in_valid_ || in0.size() != (size_t)g.length) { // (an Add, an import or a load since the last removal)
        in0.assign((size_t)g.length, {});
        inU.clear();
        for (int a = 0; a < g.count; ++a) {
            const int i = g.dense[(size_t)a];
            for (int layer = 0; layer <= g.level[(size_t)i]; ++layer) {
                const int *l = g.list(i, layer);
                for (int e = 1; e <= l[0]; ++e) in_of(l[e], layer).push_back(i);
            }
        }
    }
    in_valid_ = false; // until this call has gone through: an error return leaves the sets half updated
    auto erase_from = [](std::vector<int> &v, int x) {
        auto it = std::find(v.begin(), v.end(), x);
        if (it != v.end()) { *it = v.back(); v.pop_back(); }
    };
    // With device traversal the HBM mirror is brought up to date once and then kept in step list by list
    // (Device::patch_lists); otherwise it is rebuilt from the host lists on the next call that needs it.
    const bool on_device = p_.device_traversal && dim_ <= 2048 && dev_->traversal_fits(p_.remove_max_candidates + 1, false, p_.max_edges);
    if (on_device) { if (!sync_graph(err)) return -1; }
    else graph_dirty_ = true;
    // HNSWIndex.Remove(id) (:83-90) -> GraphConnector.RemoveNodeConnections (:53-66)
    auto remove_one = [&](const int id) -> bool {
        g.removed[(size_t)id] = 1; // item.IsRemoved = true, GraphConnector.cs:55-57
        for (int layer = g.level[(size_t)id]; layer >= 0; --layer) { // :59-66
            // ReplaceEntryPointIfNeeded :72-85
            if (id == g.entry) {
                const int *el = g.list(g.entry, layer);
                if (el[0] > 0) { // GraphData.TryReplaceEntryPoint :146-167
                    int repl = -1, maxc = -1;
                    for (int e = 1; e <= el[0]; ++e) {
                        const int c = g.list(el[e], layer)[0];
                        if (c > maxc) { maxc = c; repl = el[e]; }
                    }
                    g.entry = repl;
                } else if (layer == 0) {
                    if (g.count == 1) g.entry = -1;
                    else { // GraphData.ForceReplaceEntryPoint :173-190
                        int best_layer = -1, best_id = -1;
                        for (int a = 0; a < g.count; ++a) {
                            const int c = g.dense[(size_t)a];
                            if (g.level[(size_t)c] > best_layer) { best_layer = g.level[(size_t)c]; best_id = c; }
                        }
                        g.entry = best_id;
                    }
                }
            }
            // RemoveConnectionsAtLayer :90-167
            const int *rl = g.list(id, layer);
            for (int e = 1; e <= rl[0]; ++e) erase_from(in_of(rl[e], layer), id); // DetachOutgoingReferences :277-288
            const std::vector<int> affected = in_of(id, layer);                   // :95
            VecSource<AffectedJob> asrc;
            asrc.jobs.resize(affected.size());
            for (size_t a = 0; a < affected.size(); ++a) {
                AffectedJob &aj = asrc.jobs[a];
                aj.g = &g; aj.aid = affected[a]; aj.layer = layer; aj.removed = id;
            }
            // Graph-resident form: the search is one traversal on the device and the affected nodes are re-linked by
            // one launch (graph_relink_kernel); whatever depends on the heap-array order of the candidates (equal
            // distances, fewer candidates than MaxEdges) is repeated below on the exact lock-step path.
            bool done = false;
            if (on_device && !affected.empty()) {
                const int k = p_.remove_max_candidates, me = g.max_edges_at(0), n = (int)affected.size(); // me: stride of the selections
                std::vector<int> sid((size_t)k + 1), sflag(1), cids, sel((size_t)n * (size_t)me), scnt((size_t)n), sfl((size_t)n);
                std::vector<float> sd((size_t)k + 1);
                auto relink = [&](bool heap_order) -> int { // 1 applied, 0 flagged, -1 error
                    const std::vector<int> jl((size_t)n, layer), jr((size_t)n, id), js((size_t)n, 0);
                    const int c_off = 0, c_cnt = (int)cids.size();
                    if (!dev_->relink_batch(affected.data(), jl.data(), jr.data(), js.data(), n, cids.data(), &c_off, &c_cnt, 1, g.max_edges_at(0), sel.data(),
                                            scnt.data(), sfl.data(), me, heap_order)) {
                        err = get_dev_error();
                        return -1;
                    }
                    if (!std::all_of(sfl.begin(), sfl.end(), [](int f) { return f == 0; })) return 0;
                    for (int a = 0; a < n; ++a) {
                        AffectedJob &aj = asrc.jobs[(size_t)a];
                        aj.gather();
                        aj.apply(std::vector<int>(sel.begin() + (size_t)a * me, sel.begin() + (size_t)a * me + scnt[(size_t)a]));
                    }
                    return 1;
                };
                // (1) SearchLayer(removed, layer, k, its own vector, id != removed) (:96) on the sorted-list traversal: the
                // entry point is a candidate but not a result, so the list holds it as one extra entry (k + 1 slots, its own
                // distance keeps it in front).  Candidates come out ascending: fine unless something below is flagged.
                SearchJob sjob{~id, id, layer, layer, -1};
                if (!dev_->search_batch(&sjob, 1, k + 1, k + 1, sid.data(), sd.data(), sflag.data(), true)) { err = get_dev_error(); return false; }
                bool self_seen = false;
                for (int i = 0; i <= k && sid[(size_t)i] >= 0; ++i) {
                    if (sid[(size_t)i] == id) self_seen = true;
                    else cids.push_back(sid[(size_t)i]);
                }
                int r = 0;
                if (sflag[0] == 0 && self_seen) {
                    r = relink(false);
                    if (r < 0) return false;
                }
                // (2) equal distances somewhere, or fewer candidates than MaxEdges: the same step with the exact two-heap
                // search, whose result array IS topCandidates.ToArray() -- the candidate arrays are then the reference's
                // element for element and Span.Sort's answer follows
                if (r == 0) {
                    sjob.aux = -2;
                    if (!dev_->search_batch(&sjob, 1, k, k, sid.data(), sd.data(), sflag.data(), true, true)) { err = get_dev_error(); return false; }
                    if (sflag[0] == 0) {
                        cids.clear();
                        for (int i = 0; i < k && sid[(size_t)i] >= 0; ++i) cids.push_back(sid[(size_t)i]);
                        r = relink(true);
                        if (r < 0) return false;
                    }
                }
                done = r == 1;
            }
            if (!done) {
                VecSource<RemoveSearchJob> ssrc;
                ssrc.jobs.resize(1);
                RemoveSearchJob &sj = ssrc.jobs[0];
                sj.g = &g; sj.capacity = (int)capacity_; sj.removed = id; sj.layer = layer; sj.k = p_.remove_max_candidates;
                if (!engine()->run(ssrc, 1)) { err = get_dev_error(); return false; }
                if (!affected.empty()) {
                    for (AffectedJob &aj : asrc.jobs) aj.sc_cands = &sj.result;
                    if (!engine()->run(asrc, (long long)affected.size())) { err = get_dev_error(); return false; }
                }
            }
            for (AffectedJob &aj : asrc.jobs) {
                for (int o : aj.in_remove) erase_from(in_of(o, layer), aj.aid);
                for (int w : aj.in_add) in_of(w, layer).push_back(aj.aid);
            }
            if (on_device && !affected.empty()) { // the re-linked lists, into the HBM mirror
                const int stride = (layer == 0 ? g.stride0 : g.strideU) + 1;
                std::vector<int> recs(affected.size() * (size_t)stride, 0);
                for (size_t a = 0; a < affected.size(); ++a) {
                    const int *l = g.list(affected[a], layer);
                    int *r = recs.data() + a * (size_t)stride;
                    r[0] = affected[a]; r[1] = layer; r[2] = l[0];
                    for (int e = 0; e < l[0]; ++e) r[3 + e] = l[1 + e];
                }
                if (!dev_->patch_lists(recs.data(), (int)affected.size(), stride)) { err = get_dev_error(); return false; }
            }
            in_of(id, layer).clear();
            if (layer == 0) g.retire(id); // GraphData.RemoveItem :124-128
        }
        return true;
    };

    // Snapshot batches of removals with disjoint neighbourhoods (hnsw_mi355x_set_remove_batch(B), B > 1): the
    // deterministic counterpart of the reference's Remove(List) = Parallel.For under region locks (HNSWIndex.cs:95-101,
    // GraphLocker.cs:28-72).  Disjoint regions make the members' un-linking steps independent of each other, so
    // every (member, layer) search and every re-link of the batch is computed from the graph as it stands before
    // the batch -- two launches for the whole batch -- and the differences are applied member by member, in order.
    auto remove_in_batches = [&]() -> bool {
        const int B = p_.remove_batch, k = p_.remove_max_candidates, me = g.max_edges_at(0);
        std::vector<int> rem(ids, ids + count), next, batch;
        std::vector<unsigned char> marked((size_t)g.length + 1, 0);
        auto region = [&](int id, auto &&fn) { // itself, its out- and in-neighbours on every layer
            fn(id);
            for (int layer = 0; layer <= g.level[(size_t)id]; ++layer) {
                const int *l = g.list(id, layer);
                for (int e = 1; e <= l[0]; ++e) fn(l[e]);
                for (int v : in_of(id, layer)) fn(v);
            }
        };
        while (!rem.empty()) {
            if (rem[0] == g.entry) { // the entry point moves: alone, sequentially
                if (!remove_one(rem[0])) return false;
                rem.erase(rem.begin());
                continue;
            }
            const size_t window = std::min(rem.size(), (size_t)8 * (size_t)B);
            batch.clear(); next.clear();
            for (size_t t = 0; t < window; ++t) {
                const int id = rem[t];
                bool ok = id != g.entry && (int)batch.size() < B;
                if (ok) region(id, [&](int v) { ok = ok && !marked[(size_t)v]; });
                if (!ok) { next.push_back(id); continue; }
                batch.push_back(id);
                region(id, [&](int v) { marked[(size_t)v] = 1; });
            }
            next.insert(next.end(), rem.begin() + (long)window, rem.end());
            for (int id : batch) region(id, [&](int v) { marked[(size_t)v] = 0; });
            // (1) every (member, layer) search on the snapshot: the exact two-heap traversal, entry point filtered,
            // result = topCandidates.ToArray() (the candidate arrays below are then the reference's element for element)
            std::vector<SearchJob> sjobs;
            std::vector<int> step_id, step_layer;
            for (int id : batch)
                for (int layer = g.level[(size_t)id]; layer >= 0; --layer) {
                    sjobs.push_back(SearchJob{~id, id, layer, layer, -2});
                    step_id.push_back(id); step_layer.push_back(layer);
                }
            const int nsteps = (int)sjobs.size();
            std::vector<int> sid((size_t)nsteps * (size_t)k), sflag((size_t)nsteps);
            std::vector<float> sd((size_t)nsteps * (size_t)k);
            if (!dev_->search_batch(sjobs.data(), nsteps, k, k, sid.data(), sd.data(), sflag.data(), true, true)) { err = get_dev_error(); return false; }
            std::vector<std::vector<NodeDist>> sres((size_t)nsteps);
            for (int s2 = 0; s2 < nsteps; ++s2) {
                if (sflag[(size_t)s2] != 0) { // handed back (NaN / -0 distance, heap overflow): this search on the lock-step path, same snapshot
                    VecSource<RemoveSearchJob> ssrc;
                    ssrc.jobs.resize(1);
                    RemoveSearchJob &sj = ssrc.jobs[0];
                    sj.g = &g; sj.capacity = (int)capacity_; sj.removed = step_id[(size_t)s2]; sj.layer = step_layer[(size_t)s2]; sj.k = k;
                    if (!engine()->run(ssrc, 1)) { err = get_dev_error(); return false; }
                    sres[(size_t)s2] = sj.result;
                } else
                    for (int i = 0; i < k && sid[(size_t)s2 * k + i] >= 0; ++i) sres[(size_t)s2].push_back(NodeDist{sid[(size_t)s2 * k + i], sd[(size_t)s2 * k + i]});
            }
            // (2) every affected node of every step re-linked from the snapshot, one launch
            std::vector<int> j_aid, j_layer, j_rem, j_step, c_off((size_t)nsteps), c_cnt((size_t)nsteps), c_all;
            for (int s2 = 0; s2 < nsteps; ++s2) {
                c_off[(size_t)s2] = (int)c_all.size();
                c_cnt[(size_t)s2] = (int)sres[(size_t)s2].size();
                for (const NodeDist &nd : sres[(size_t)s2]) c_all.push_back(nd.id);
                for (int a : in_of(step_id[(size_t)s2], step_layer[(size_t)s2])) {
                    j_aid.push_back(a); j_layer.push_back(step_layer[(size_t)s2]); j_rem.push_back(step_id[(size_t)s2]); j_step.push_back(s2);
                }
            }
            const int nj = (int)j_aid.size();
            std::vector<int> sel((size_t)std::max(nj, 1) * (size_t)me), scnt((size_t)std::max(nj, 1)), sfl((size_t)std::max(nj, 1));
            if (nj > 0 && !dev_->relink_batch(j_aid.data(), j_layer.data(), j_rem.data(), j_step.data(), nj, c_all.data(), c_off.data(), c_cnt.data(), nsteps, me,
                                              sel.data(), scnt.data(), sfl.data(), me, true)) {
                err = get_dev_error();
                return false;
            }
            // (3) apply, member by member (GraphConnector.cs:55-66, :90-167)
            for (int id : batch) g.removed[(size_t)id] = 1;
            std::vector<int> recs;
            int nrecs = 0;
            const int rstride = g.stride0 + 1;
            int jpos = 0;
            for (int s2 = 0; s2 < nsteps; ++s2) {
                const int id = step_id[(size_t)s2], layer = step_layer[(size_t)s2];
                const int *rl = g.list(id, layer);
                for (int e = 1; e <= rl[0]; ++e) erase_from(in_of(rl[e], layer), id); // DetachOutgoingReferences :277-288
                const std::vector<int> affected = in_of(id, layer);
                VecSource<AffectedJob> asrc;
                asrc.jobs.resize(affected.size());
                bool flagged = false;
                for (size_t a = 0; a < affected.size(); ++a) {
                    AffectedJob &aj = asrc.jobs[a];
                    aj.g = &g; aj.aid = affected[a]; aj.layer = layer; aj.removed = id; aj.sc_cands = &sres[(size_t)s2];
                    flagged = flagged || sfl[(size_t)(jpos + (int)a)] != 0;
                }
                if (!flagged) {
                    for (size_t a = 0; a < affected.size(); ++a) {
                        const size_t j = (size_t)jpos + a;
                        asrc.jobs[a].gather();
                        asrc.jobs[a].apply(std::vector<int>(sel.begin() + j * (size_t)me, sel.begin() + j * (size_t)me + scnt[j]));
                    }
                } else if (!affected.empty()) { // (LDS capacity): these re-links on the lock-step path, same candidates
                    if (!engine()->run(asrc, (long long)affected.size())) { err = get_dev_error(); return false; }
                }
                jpos += (int)affected.size();
                for (AffectedJob &aj : asrc.jobs) {
                    for (int o : aj.in_remove) erase_from(in_of(o, layer), aj.aid);
                    for (int w : aj.in_add) in_of(w, layer).push_back(aj.aid);
                }
                for (int a : affected) {
                    const int *l = g.list(a, layer);
                    recs.resize((size_t)(nrecs + 1) * (size_t)rstride, 0);
                    int *r = recs.data() + (size_t)nrecs * (size_t)rstride;
                    r[0] = a; r[1] = layer; r[2] = l[0];
                    for (int e = 0; e < l[0]; ++e) r[3 + e] = l[1 + e];
                    ++nrecs;
                }
                in_of(id, layer).clear();
                if (layer == 0) g.retire(id); // GraphData.RemoveItem :124-128
            }
            if (nrecs > 0 && !dev_->patch_lists(recs.data(), nrecs, rstride)) { err = get_dev_error(); return false; }
            rem.swap(next);
        }
        return true;
    };
    // From here on the graph changes step by step (flags, host lists, then the HBM mirror): a device failure in the
    // middle leaves a node flagged removed but still linked and the mirror out of step with the host lists, so -- as
    // for a failed Add -- the index refuses every later call with the original message.
    if (p_.remove_batch > 1 && on_device) { if (!remove_in_batches()) { graph_dirty_ = true; return fail("Remove: " + err, err); } }
    else
        for (int t = 0; t < count; ++t)
            if (!remove_one(ids[t])) { graph_dirty_ = true; return fail("Remove: " + err, err); }
    in_valid_ = true;
    return 0;
}

// ---- Serialize / Deserialize ------------------------------------------------------------------
int HnswIndex::serialize(const char *path, std::string &err)
{
    if (!path) { err = "System.ArgumentNullException: filePath"; return -1; }
    if (failed(err)) return -1;
    if (metric_ == HNSWDEV_SQ_EUCLID_I8) { err = "System.NotSupportedException: Serialize: the snapshot format stores float items; an int8 index keeps quantised records only"; return -1; }
    if (!refresh_host_lists(err)) return -1;
    SnapshotParams sp;
    sp.max_edges = p_.max_edges;
    sp.distribution_rate = p_.distribution_rate;
    sp.min_nn = p_.min_nn;
    sp.max_candidates = p_.max_candidates;
    sp.remove_max_candidates = p_.remove_max_candidates;
    sp.collection_size = p_.collection_size;
    sp.random_seed = p_.random_seed;
    sp.allow_removals = p_.allow_removals;
    std::vector<float> rows((size_t)graph_.length * (size_t)dim_);
    if (graph_.length > 0 && !dev_->download_rows(0, graph_.length, rows.data())) { err = get_dev_error(); return -1; }
    return write_snapshot(path, sp, graph_, rows.data(), dim_, capacity_, err) ? 0 : -1;
}

HnswIndex *HnswIndex::deserialize(int metric, const Params &backend, const char *path, std::string &err)
{
    if (!path) { err = "System.ArgumentNullException: filePath"; return nullptr; }
    if (metric == HNSWDEV_SQ_EUCLID_I8) { err = "System.NotSupportedException: Deserialize: not available for the int8 metric"; return nullptr; }
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { err = std::string("System.IO.FileNotFoundException: Could not find file '") + path + "'"; return nullptr; }
    struct stat st;
    if (::fstat(fd, &st) != 0) { ::close(fd); err = "System.IO.IOException: fstat"; return nullptr; }
    const size_t len = (size_t)st.st_size;
    void *map = len ? ::mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    ::close(fd);
    if (len && map == MAP_FAILED) { err = "System.IO.IOException: mmap"; return nullptr; }
    SnapshotParams sp;
    Graph g;
    std::vector<float> rows;
    int dim = 0;
    long long capacity = 0;
    const bool ok = read_snapshot(static_cast<const uint8_t *>(map), len, sp, g, rows, dim, capacity, err);
    if (len) ::munmap(map, len);
    if (!ok) return nullptr;
    Params p = backend;
    p.max_edges = sp.max_edges;
    p.distribution_rate = sp.distribution_rate;
    p.min_nn = sp.min_nn;
    p.max_candidates = sp.max_candidates;
    p.remove_max_candidates = sp.remove_max_candidates;
    p.collection_size = sp.collection_size;
    p.random_seed = sp.random_seed;
    p.allow_removals = sp.allow_removals;
    // GraphData's snapshot constructor (GraphData.cs:58-74): the RNG restarts from RandomSeed
    // (create() does that), Capacity / Length / EntryPointId are taken from the snapshot
    HnswIndex *ix = create(metric, p, err);
    if (!ix) return nullptr;
    ix->capacity_ = std::max<long long>(1, capacity);
    if (g.length > 0) {
        if (!ix->ensure_dim(dim, err)) { delete ix; return nullptr; }
        ix->graph_ = std::move(g);
        if (!ix->dev_->upload_rows(0, ix->graph_.length, rows.data())) { err = get_dev_error(); delete ix; return nullptr; }
    } else {
        ix->graph_ = std::move(g);
    }
    ix->graph_dirty_ = true;
    return ix;
}

// ---- import of a graph built elsewhere --------------------------------------------------------
int HnswIndex::import_nodes(const float *rows, int n, int dim, const int *levels, int entry_point, std::string &err)
{
    if (failed(err)) return -1;
    in_valid_ = false;
    ++graph_epoch_;
    if (!rows || !levels || n <= 0 || dim <= 0) { err = "System.ArgumentException: hnsw_mi355x_import_nodes: bad argument"; return -1; }
    if (graph_.length != 0) { err = "System.InvalidOperationException: hnsw_mi355x_import_nodes: the index already holds items"; return -1; }
    if (entry_point < 0 || entry_point >= n) { err = "System.ArgumentException: hnsw_mi355x_import_nodes: entry point outside the nodes"; return -1; }
    for (int i = 0; i < n; ++i)
        if (levels[i] < 0 || levels[i] > 200) { err = "System.ArgumentException: hnsw_mi355x_import_nodes: level out of range"; return -1; }
    if (!ensure_dim(dim, err)) return -1;
    if (!ensure_capacity(n, err)) return -1;
    graph_.grow_for(n);
    for (int i = 0; i < n; ++i) (void)graph_.add_node(levels[i], false);
    for (int i = 0; i < n; ++i) (void)rng_.next_single(); // one level draw per item, as on the index the graph came from
    graph_.entry = entry_point;
    if (!dev_->upload_rows(0, n, rows)) return fail(get_dev_error(), err);
    graph_dirty_ = true;       // the HBM mirror is (re)built from the host lists on the next call
    host_lists_stale_ = false;
    return 0;
}

int HnswIndex::import_edges(int layer, const int *counts, const int *edges, int stride, std::string &err)
{
    if (failed(err)) return -1;
    in_valid_ = false;
    ++graph_epoch_;
    if (!counts || !edges || layer < 0 || stride < 1) { err = "System.ArgumentException: hnsw_mi355x_import_edges: bad argument"; return -1; }
    if (graph_.length <= 0) { err = "System.InvalidOperationException: hnsw_mi355x_import_edges: call hnsw_mi355x_import_nodes first"; return -1; }
    if (!refresh_host_lists(err)) return -1;
    const int n = graph_.length, cap = graph_.max_edges_at(layer);
    // validate everything before anything is written: a bad list must not leave a half-imported layer behind
    std::vector<int> stamp((size_t)n, -1);
    for (int i = 0; i < n; ++i) {
        if (graph_.level[(size_t)i] < layer) continue;
        const int c = counts[i];
        if (c < 0 || c > cap || c > stride) { err = "System.ArgumentException: hnsw_mi355x_import_edges: list longer than MaxEdges(layer) (or than the stride)"; return -1; }
        for (int j = 0; j < c; ++j) {
            const int t = edges[(size_t)i * stride + j];
            if (t < 0 || t >= n || graph_.level[(size_t)t] < layer) { err = "System.ArgumentException: hnsw_mi355x_import_edges: edge to a node outside the layer"; return -1; }
            if (stamp[(size_t)t] == i) { err = "System.ArgumentException: hnsw_mi355x_import_edges: duplicate id in a list"; return -1; }
            stamp[(size_t)t] = i;
        }
    }
    for (int i = 0; i < n; ++i) {
        if (graph_.level[(size_t)i] < layer) continue;
        int *l = graph_.list(i, layer);
        l[0] = counts[i];
        std::memcpy(l + 1, edges + (size_t)i * stride, sizeof(int) * (size_t)counts[i]);
    }
    graph_dirty_ = true;
    return 0;
}

uint64_t HnswIndex::graph_hash()
{
    std::string e;
    (void)refresh_host_lists(e);
    return graph_hash_of(graph_);
}

// Neighbour lists written by the device link half, back into the host copy.
bool HnswIndex::refresh_host_lists(std::string &err)
{
    if (!host_lists_stale_) return true;
    // nodes / pool blocks appended on the host since the last sync are not in the mirror yet (their lists are empty)
    const long long n = std::min<long long>(graph_.length, dev_ ? dev_->graph_nodes() : 0);
    const long long pl = std::min<long long>((long long)graph_.pool.size(), dev_pool_len_);
    if (!dev_ || !dev_->download_graph(graph_.adj0.data(), n, graph_.pool.data(), pl)) {
        err = get_dev_error();
        return false;
    }
    host_lists_stale_ = false;
    return true;
}

} // namespace hnsw
