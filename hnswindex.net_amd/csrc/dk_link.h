// dk_link.h -- device code, part of device_kernels.h: Add, link half and Remove's re-link: graph_relink_kernel, link_group, graph_link_kernel, the dry run, the device-side grouping.
#pragma once
#include "dk_heuristic.h"

namespace hnsw {

// Insert, link half, on the HBM mirror.  (a) new nodes' own lists.
#ifdef HNSW_HOST_TU // launched from one place: defined only in the unit that launches it
// Remove, second half (GraphConnector.RemoveConnectionsAtLayer :100-133): one wave per AFFECTED node (an in-edge
// neighbour of the removed node): drop the edge to the removed node (EdgeList.Remove: the last entry takes its
// place), candidates = the remaining neighbours followed by the search candidates that are neither the node itself
// nor among them (:115-129), Distance(candidate, node) for all of them, RelativeNeighborPruning (:131).  Nothing
// is written to the graph: the selection goes back to the host, which applies the difference (:135-164).
// `cands` arrive ascending by distance to the removed node, not in the reference's heap-array order; that order
// shows only if the heuristic returns its input unsorted (fewer candidates than MaxEdges) or sorts equal distances:
// both raise out_flag and the host repeats the step on the exact lock-step path.
template <int METRIC>
__global__ void __launch_bounds__(64)
graph_relink_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, const int *__restrict__ adj0, int stride0,
                    const int64_t *__restrict__ upper, const int *__restrict__ pool, int strideU, const int4 *__restrict__ jobs,
                    const int *__restrict__ cands_all, const int *__restrict__ cand_off, const int *__restrict__ cand_cnt, int max_edges0,
                    int kcap, int nbcap, int *__restrict__ out_sel, int *__restrict__ out_cnt, int *__restrict__ out_flag, int sel_stride,
                    unsigned long long *__restrict__ eval_counter, int heap_order)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x, job = blockIdx.x;
    const SearchLds L = carve_lds(smem, kcap, 0, dim, nbcap);
    const GraphView G{adj0, stride0, upper, pool, strideU};
    // jobs[]: (affected node, layer, removed node, step); the step's search candidates: cands_all[cand_off[step] ..][0 .. cand_cnt[step])
    const int4 jd = jobs[job];
    const int aid = jd.x, layer = jd.y, removed = jd.z;
    const int *cands = cands_all + cand_off[jd.w];
    const int ncand = cand_cnt[jd.w];
    const int max_edges = layer == 0 ? max_edges0 : (max_edges0 >> 1); // GraphData.MaxEdges :247-250
    const float *q = rows + (size_t)aid * dim;
    double sb = 0.0;
    if (METRIC == M_COS) sb = row_sn[aid];
    for (int i = lane; i < dim; i += 64) L.qs[i] = q[i];
    const int *l = G.list(aid, layer);
    int cnt = l[0];
    // RemoveOutEdge :104 (EdgeList.Remove, Node.cs:79-93: swap with the last)
    int pos = -1;
    for (int base = 0; base < cnt && pos < 0; base += 64) {
        const unsigned long long hit = __ballot(base + lane < cnt && l[1 + base + lane] == removed);
        if (hit) pos = base + (int)__builtin_ctzll(hit);
    }
    const int last = cnt - 1;
    if (pos >= 0) --cnt;
    for (int i = lane; i < cnt; i += 64) L.nbuf[i] = (i == pos) ? l[1 + last] : l[1 + i]; // :110-120 the existing neighbours
    wave_sync();
    int n = cnt;
    bool bad = false;
    for (int base = 0; base < ncand; base += 64) { // :123-129
        const int i = base + lane;
        const int c = i < ncand ? cands[i] : -1;
        bool keep = i < ncand && c != aid;
        for (int t = 0; keep && t < cnt; ++t) keep = L.nbuf[t] != c;
        const unsigned long long mask = __ballot(keep);
        const int posn = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        if (n + (int)__popcll(mask) > min(kcap, nbcap)) { bad = true; break; }
        if (keep) L.nbuf[n + posn] = c;
        n += (int)__popcll(mask);
    }
    wave_sync();
    unsigned long long evals = 0;
    int rc = 0;
    // heap_order: `cands` are SearchLayer's heap array itself (the exact two-heap search), so the candidate array is
    // the reference's, element for element, and nothing below depends on anything else
    if (!bad && !heap_order && n < max_edges) bad = true; // Heuristic.cs:13-18 returns the INPUT order: the heap array's
    if (!bad && n == 0) rc = 0;
    if (!bad && n > 0) {
        measure_all<METRIC>(rows, row_sn, dim, L.qs, sb, L.nbuf, L.dbuf, n, lane); // Distance(id, affectedNodeId) :118, :128
        wave_sync();
        evals += (unsigned long long)n;
        for (int i = lane; i < n; i += 64) L.top[i] = ND{L.nbuf[i], L.dbuf[i]};
        wave_sync();
        rc = relative_neighbor_pruning<METRIC>(rows, row_sn, dim, L.top, n, max_edges, L, lane, evals); // sorts L.top
        wave_sync();
        bool odd = false; // equal, NaN or -0 distances: Span.Sort's answer depends on the input order
        for (int i = lane; i < n; i += 64) {
            const float d = L.top[i].dist;
            odd |= key_unsafe(d) || (i + 1 < n && f2key(L.top[i + 1].dist) == f2key(d));
        }
        if (!heap_order && __ballot(odd) != 0ull) bad = true;
    }
    if (!bad) for (int i = lane; i < rc; i += 64) out_sel[(size_t)job * sel_stride + i] = L.acc[i];
    if (lane == 0) {
        out_cnt[job] = bad ? 0 : rc;
        out_flag[job] = bad ? 1 : 0;
        atomicAdd(eval_counter, evals);
    }
}
#endif

#ifdef HNSW_HOST_TU // non-template kernels: only the unit that launches them defines them
__global__ void __launch_bounds__(64)
graph_write_rows_kernel(int *__restrict__ adj0, int stride0, const int64_t *__restrict__ upper, int *__restrict__ pool,
                        int strideU, const int *__restrict__ recs, int row_stride, int *__restrict__ tested0,
                        int *__restrict__ testedU, int max_edges0)
{
    const int *r = recs + (size_t)blockIdx.x * row_stride;
    const int node = r[0], layer = r[1] & 0xffff, cnt = r[2];
    const bool untested = (r[1] >> 30) & 1; // the list is not a heuristic's ordered output (a removal's re-link)
    int *l = layer == 0 ? adj0 + (size_t)node * stride0 : pool + upper[node] + (size_t)(layer - 1) * strideU;
    if (threadIdx.x == 0) {
        l[0] = cnt;
        // a full list can only be the ordered output of the heuristic's greedy pass (fewer candidates
        // than MaxEdges come back unsorted, Heuristic.cs:13-18): its entries are mutually tested
        const int me = layer == 0 ? max_edges0 : (max_edges0 >> 1);
        int *t = layer == 0 ? tested0 + node : testedU + (upper[node] / strideU + (layer - 1));
        *t = (cnt == me && !untested) ? cnt : 0;
    }
    for (int i = threadIdx.x; i < cnt; i += 64) l[1 + i] = r[3 + i];
}
#endif

// (b) one wave per (neighbour, layer) list: every back-edge append of the batch, in item order
// (neighbor.OutEdges[layer].Add(currNode.Id), GraphConnector.cs:207), each overflow pruned in
// place (PruneOverflow :222-262: distances :230-234, sort + heuristic :235).  Lists are
// independent, so the outcome equals the reference's sequential loop.
// next_item(): the next node id to append to this list, in item order, or -1.  out_list (optional):
// [count, ids...] of the final list for the host.
template <int METRIC, class NextItem>
__device__ __forceinline__ void link_group(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, int *__restrict__ adj0,
                  int stride0, const int64_t *__restrict__ upper, int *__restrict__ pool, int strideU,
                  int node, int layer, NextItem next_item, int max_edges0, int k_cap, int *__restrict__ out_list,
                  unsigned long long *__restrict__ eval_counter, int nbcap, int *__restrict__ tested0, int *__restrict__ testedU,
                  unsigned char *smem, int *__restrict__ dry_changed = nullptr, int *__restrict__ dry_drop = nullptr, int dry_item = -1)
{
    const SearchLds L = carve_lds(smem, k_cap, 0, dim, nbcap);
    // shortcut scratch behind the common carve-up: distances of up to kNewMax new entries to every
    // entry of the list, and the sorted order as original positions
    float *Dm = reinterpret_cast<float *>(smem + ((search_lds_bytes(k_cap, 0, dim, true, nbcap) + 15) & ~(size_t)15));
    int *perm = reinterpret_cast<int *>(Dm + kNewMax * nbcap);
    const int lane = threadIdx.x;
    const int max_edges = layer == 0 ? max_edges0 : (max_edges0 >> 1);
    int *l = layer == 0 ? adj0 + (size_t)node * stride0 : pool + upper[node] + (size_t)(layer - 1) * strideU;
    const float *q = rows + (size_t)node * dim;
    double sb = 0.0;
    if (METRIC == M_COS) sb = row_sn[node];
    for (int i = lane; i < dim; i += 64) L.qs[i] = q[i];
    int cnt = l[0];
    for (int i = lane; i < cnt; i += 64) L.nbuf[i] = l[1 + i];
    int *tested_p = layer == 0 ? tested0 + node : testedU + (upper[node] / strideU + (layer - 1));
    int tested = min(max(*tested_p, 0), cnt); // leading entries that are an ordered, mutually tested heuristic output
    wave_sync();
    unsigned long long evals = 0;
    PH_DECL();
    PH(0);
    for (int item = next_item(); item >= 0; item = next_item()) {
        if (lane == 0) L.nbuf[cnt] = item; // :207
        cnt++;
        wave_sync();
        PH_COUNT(6, 1);
        if (cnt > max_edges) { // :209
            measure_all<METRIC>(rows, row_sn, dim, L.qs, sb, L.nbuf, L.dbuf, cnt, lane); // Distance(cand, node.Id) :233
            wave_sync();
            evals += (unsigned long long)cnt;
            int rc = -1;
            // Shortcut.  The first `tested` entries are the output of an earlier greedy pass over this
            // very list (same node, same distances): ascending, and every earlier one already passed
            // the test `dist(s, c) < c.Dist` against every later one (Heuristic.cs:31-35).  Those pairs
            // need not be measured again; only pairs with one of the entries appended since do.  With
            // few new entries (typically one: lists are full, every append overflows) that is one
            // batch of distances per new entry instead of one dependent batch per candidate.
            const int n = cnt, u = n - tested;
            if (tested > 0 && u <= kNewMax && n <= 128) { // entries i = lane and i = lane + 64 on each lane
                const int i1 = lane + 64;
                const float d0 = lane < n ? L.dbuf[lane] : 0.0f, d1 = i1 < n ? L.dbuf[i1] : 0.0f;
                const unsigned k0 = f2key(d0), k1 = f2key(d1);
                bool odd = (lane < n && key_unsafe(d0)) || (i1 < n && key_unsafe(d1));
                int rank0 = 0, rank1 = 0;
                for (int t2 = 0; t2 < n; ++t2) { // Span.Sort :22 -- distinct ordinary distances: rank by counting
                    const unsigned kt = t2 < 64 ? (unsigned)__builtin_amdgcn_readlane((int)k0, t2) : (unsigned)__builtin_amdgcn_readlane((int)k1, t2 - 64);
                    rank0 += kt < k0 ? 1 : 0;
                    rank1 += kt < k1 ? 1 : 0;
                    odd |= lane < n && t2 != lane && kt == k0;
                    odd |= i1 < n && t2 != i1 && kt == k1;
                    // the tested prefix must still be ascending (it is, by construction)
                    odd |= lane < tested && t2 < tested && ((t2 < lane && kt >= k0) || (t2 > lane && kt <= k0));
                    odd |= i1 < tested && t2 < tested && ((t2 < i1 && kt >= k1) || (t2 > i1 && kt <= k1));
                }
                if (__ballot(odd) == 0ull) {
                    if (lane < n) perm[rank0] = lane;
                    if (i1 < n) perm[rank1] = i1;
                    // distances of every new entry to all entries of the list
                    for (int jn = 0; jn < u; ++jn) {
                        const int xid = L.nbuf[tested + jn];
                        const float *xrow = rows + (size_t)xid * dim;
                        wave_sync();
                        for (int t2 = lane; t2 < dim; t2 += 64) L.qs2[t2] = xrow[t2];
                        double sbx = 0.0;
                        if (METRIC == M_COS) sbx = row_sn[xid];
                        wave_sync();
                        // a single new entry only meets the old ones (one pass of <= 32 rows instead of two)
                        const int mrows = u == 1 ? tested : n;
                        measure_all<METRIC>(rows, row_sn, dim, L.qs2, sbx, L.nbuf, Dm + jn * nbcap, mrows, lane);
                        evals += (unsigned long long)(u == 1 ? mrows : n - 1);
                    }
                    wave_sync();
                    // greedy pass :23-40 in sorted order, on the distances at hand
                    bool acc0 = false, acc1 = false; // entries lane / lane + 64 accepted
                    unsigned new_acc = 0u;           // bit j: new entry j accepted
                    rc = 0;
                    for (int p2 = 0; p2 < n && rc < max_edges; ++p2) {
                        const int i = perm[p2];
                        const float di = L.dbuf[i];
                        bool rej;
                        if (i < tested) {           // an old entry: only accepted new ones can object
                            rej = false;
                            for (int jn = 0; jn < u; ++jn)
                                if ((new_acc >> jn) & 1u) rej = rej || Dm[jn * nbcap + i] < di;
                        } else {                    // a new entry: everything accepted so far can object
                            const float *Dj = Dm + (i - tested) * nbcap;
                            const float e0 = lane < n ? Dj[lane] : 0.0f, e1 = i1 < n ? Dj[i1] : 0.0f;
                            rej = __ballot((acc0 && e0 < di) || (acc1 && e1 < di)) != 0ull;
                        }
                        if (!rej) {
                            if (lane == i) acc0 = true;
                            if (i1 == i) acc1 = true;
                            if (i >= tested) new_acc |= 1u << (i - tested);
                            if (lane == 0) L.acc[rc] = L.nbuf[i];
                            rc++;
                        }
                    }
                    wave_sync();
                }
            }
            if (rc < 0) {
                for (int i = lane; i < cnt; i += 64) L.top[i] = ND{L.nbuf[i], L.dbuf[i]};
                rc = relative_neighbor_pruning<METRIC>(rows, row_sn, dim, L.top, cnt, max_edges, L, lane, evals);
            }
            for (int i = lane; i < rc; i += 64) L.nbuf[i] = L.acc[i]; // node.OutEdges[layer] = newOut :236
            cnt = rc;
            tested = rc; // the whole list is a greedy output now
            wave_sync();
        }
    }
    if (dry_changed) { // dry run (exact-window Add): nothing is written; would the list read differently afterwards?
        // code 0: the same sequence of ids.  Otherwise bit 0 set, bit 1 = the appended item stays in the list, bits 8.. = how
        // many ids the list loses (their ids to dry_drop[0..3), at most three; 255 = more than that).
        const int oc = l[0];
        bool diff = cnt != oc;
        for (int i = lane; i < cnt && !diff; i += 64) diff = L.nbuf[i] != l[1 + i];
        int code = 0;
        if (__ballot(diff) != 0ull) {
            bool has = false;
            for (int i = lane; i < cnt; i += 64) has = has || L.nbuf[i] == dry_item;
            code = 1 | (__ballot(has) != 0ull ? 2 : 0);
            int nd = 0;
            for (int base = 0; base < oc; base += 64) {
                const int i = base + lane;
                bool gone = false;
                int x = 0;
                if (i < oc) {
                    x = l[1 + i];
                    gone = true;
                    for (int u = 0; u < cnt; ++u) gone = gone && L.nbuf[u] != x;
                }
                unsigned long long m = __ballot(gone);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const int gid = __builtin_amdgcn_readlane(x, src);
                    if (nd < 3 && dry_drop && lane == 0) dry_drop[nd] = gid;
                    nd++;
                }
            }
            code |= (nd > 3 ? 255 : nd) << 8;
        }
        if (lane == 0) { *dry_changed = code; atomicAdd(eval_counter, evals); }
        wave_sync();
        return;
    }
    if (lane == 0) { l[0] = cnt; *tested_p = tested; if (out_list) out_list[0] = cnt; }
    for (int i = lane; i < cnt; i += 64) { l[1 + i] = L.nbuf[i]; if (out_list) out_list[1 + i] = L.nbuf[i]; }
    if (lane == 0) atomicAdd(eval_counter, evals);
    wave_sync();
}

// groups prepared by the host: one block per group, items in CSR order
template <int METRIC>
__global__ void __launch_bounds__(64)
graph_link_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, int *__restrict__ adj0,
                  int stride0, const int64_t *__restrict__ upper, int *__restrict__ pool, int strideU,
                  const int *__restrict__ g_node, const int *__restrict__ g_layer, const int *__restrict__ g_off,
                  const int *__restrict__ g_count, const int *__restrict__ g_items, int max_edges0, int k_cap,
                  int *__restrict__ out_lists, int list_stride,
                  unsigned long long *__restrict__ eval_counter, int nbcap, int *__restrict__ tested0, int *__restrict__ testedU,
                  const int *__restrict__ n_groups)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int g = blockIdx.x;
    // launched with an upper bound of the group count (small batches: no read-back in between); n_groups = the link plan's
    // counters: [0] groups, [3] the guard word -- nothing is dereferenced once a guard has tripped
    if (n_groups && (g >= n_groups[0] || n_groups[3] != 0)) return;
    int t = g_off[g];
    const int t_end = g_count ? t + g_count[g] : g_off[g + 1]; // CSR offsets, or start + count per group
    link_group<METRIC>(rows, row_sn, dim, adj0, stride0, upper, pool, strideU, g_node[g], g_layer[g],
                       [&]() { return t < t_end ? g_items[t++] : -1; }, max_edges0, k_cap,
                       out_lists ? out_lists + (size_t)g * list_stride : (int *)nullptr, eval_counter, nbcap, tested0, testedU, smem);
}

// Dry run of single appends (exact-window Add): job g = (node, layer, item) -- would appending `item` to that list,
// with PruneOverflow if it overflows (GraphConnector.cs:207-212), leave a list that READS differently (another
// sequence of ids)?  A full list whose prune turns the new item away comes out as the very same sequence (the
// earlier entries are a greedy output: ascending, mutually tested), and three out of four appends into a grown
// graph end that way: for every search that read the list, such an append never happened.  Writes nothing.
template <int METRIC>
__global__ void __launch_bounds__(64)
graph_link_dry_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, int *__restrict__ adj0,
                      int stride0, const int64_t *__restrict__ upper, int *__restrict__ pool, int strideU,
                      const int *__restrict__ jobs3, int max_edges0, int k_cap, int *__restrict__ out_changed,
                      unsigned long long *__restrict__ eval_counter, int nbcap, int *__restrict__ tested0, int *__restrict__ testedU)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int g = blockIdx.x;
    int item = jobs3[3 * g + 2];
    const int the_item = item;
    link_group<METRIC>(rows, row_sn, dim, adj0, stride0, upper, pool, strideU, jobs3[3 * g], jobs3[3 * g + 1],
                       [&]() { const int r = item; item = -1; return r; }, max_edges0, k_cap, (int *)nullptr, eval_counter, nbcap,
                       tested0, testedU, smem, out_changed + g, (int *)nullptr, the_item);
}

// The same for the selections an insert search just left on the device (no host step in between): block b stands
// for entry b % sel_stride of selection row b / sel_stride -- rows [0, njobs) are the jobs' layer-0 selections,
// row njobs + u is upper slot u, whose job is upper_owner[u].  out0 / outU (same shape as the selections) are
// preset to 1 by the host; rows a job did not produce (stop_layer), handed-back jobs and entries beyond the
// count keep that.
template <int METRIC>
__global__ void __launch_bounds__(64)
graph_link_dry_sel_kernel(const float *__restrict__ rows, const double *__restrict__ row_sn, int dim, int *__restrict__ adj0,
                          int stride0, const int64_t *__restrict__ upper, int *__restrict__ pool, int strideU,
                          const SearchJob *__restrict__ jobs, const int *__restrict__ flag, const int *__restrict__ sel0,
                          const int *__restrict__ cnt0, const int *__restrict__ selU, const int *__restrict__ cntU, int sel_stride,
                          const int *__restrict__ upper_owner, int njobs, int max_edges0, int k_cap, int *__restrict__ out0,
                          int *__restrict__ outU, unsigned long long *__restrict__ eval_counter, int nbcap, int *__restrict__ tested0,
                          int *__restrict__ testedU, long long n_nodes, int *__restrict__ drop0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int r = blockIdx.x / sel_stride, e = blockIdx.x % sel_stride;
    int job, layer, cnt;
    const int *sel;
    int *out, *drop = nullptr; // the ids a list would lose are reported for layer 0 (three per entry)
    if (r < njobs) {
        job = r; layer = 0;
        if (jobs[job].stop_layer > 0) return;
        cnt = cnt0[r]; sel = sel0 + (size_t)r * sel_stride; out = out0 + (size_t)r * sel_stride;
        drop = drop0 + ((size_t)r * sel_stride + e) * 3;
    } else {
        const int u = r - njobs;
        job = upper_owner[u];
        if (job < 0 || job >= njobs) return;
        layer = u - jobs[job].aux + 1;
        if (layer < 1 || layer > jobs[job].search_layer || layer < jobs[job].stop_layer) return;
        cnt = cntU[u]; sel = selU + (size_t)u * sel_stride; out = outU + (size_t)u * sel_stride;
    }
    if (flag[job] == 1 || e >= cnt || cnt > (layer == 0 ? max_edges0 : (max_edges0 >> 1))) return;
    const int nb = sel[e];
    int item = ~jobs[job].qref;
    if (nb < 0 || nb >= n_nodes || item < 0 || item >= n_nodes) return;
    const int the_item = item;
    link_group<METRIC>(rows, row_sn, dim, adj0, stride0, upper, pool, strideU, nb, layer,
                       [&]() { const int x = item; item = -1; return x; }, max_edges0, k_cap, (int *)nullptr, eval_counter, nbcap,
                       tested0, testedU, smem, out + e, drop, the_item);
}

// ---- the same with the grouping done on the device (no host work between the insert search and
// the link half).  Per adjacency-list slot (layer 0: the node id; upper layers: cap_n + list index
// in the pool) three counters, all zero between batches: appends, fill cursor, start offset. ----
struct LinkPlan {
    int *cnt, *fill, *off;                      // per list slot
    int *g_node, *g_layer, *g_start, *g_count;  // per group (a list that receives appends), any order
    int *items;                                 // batch positions of the appending items, grouped
    int *counters;                              // [0] groups, [1] item cursor, [3] first guard that fired
    long long cap_n;
    long long n_slots, n_nodes; // capacities, for the guards below: an index outside them is reported, never used
    int g_cap, n_jobs;
};
#define LINK_GUARD(cond, code) if (!(cond)) { atomicCAS(&P.counters[3], 0, (code)); continue; }
__device__ __forceinline__ long long link_slot(const LinkPlan &P, const int64_t *upper, int strideU, int nb, int layer)
{
    return layer == 0 ? (long long)nb : P.cap_n + upper[nb] / strideU + (layer - 1);
}
// pass 1 (count = true): the new nodes' own lists go into the mirror (currNode.OutEdges[layer] =
// selected, GraphConnector.cs:192), appends are counted per target list and the lists that receive
// any are enumerated.  pass 2 (count = false): the appends are filed per list.
template <bool COUNT>
__global__ void __launch_bounds__(64)
link_plan_kernel(const SearchJob *__restrict__ jobs, const int *__restrict__ sel0, const int *__restrict__ cnt0,
                 const int *__restrict__ selU, const int *__restrict__ cntU, int sel_stride, int *__restrict__ adj0, int stride0,
                 const int64_t *__restrict__ upper, int *__restrict__ pool, int strideU, int *__restrict__ tested0,
                 int *__restrict__ testedU, int max_edges0, LinkPlan P)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    const SearchJob jb = jobs[t];
    const int id = ~jb.qref;
    for (int layer = jb.search_layer; layer >= 0; --layer) {
        const int *sel = layer == 0 ? sel0 + (size_t)t * sel_stride : selU + (size_t)(jb.aux + layer - 1) * sel_stride;
        const int sc = layer == 0 ? cnt0[t] : cntU[jb.aux + layer - 1];
        LINK_GUARD(id >= 0 && id < P.n_nodes && sc >= 0 && sc <= sel_stride && sc <= (layer == 0 ? max_edges0 : (max_edges0 >> 1)), 1);
        if (COUNT) {
            int *l = layer == 0 ? adj0 + (size_t)id * stride0 : pool + upper[id] + (size_t)(layer - 1) * strideU;
            if (lane == 0) {
                l[0] = sc;
                const int me = layer == 0 ? max_edges0 : (max_edges0 >> 1);
                int *tp = layer == 0 ? tested0 + id : testedU + (upper[id] / strideU + (layer - 1));
                *tp = sc == me ? sc : 0; // see graph_write_rows_kernel
            }
            for (int e = lane; e < sc; e += 64) l[1 + e] = sel[e];
        }
        for (int e = lane; e < sc; e += 64) {
            const int nb = sel[e];
            LINK_GUARD(nb >= 0 && nb < P.n_nodes, 2);
            const long long slot = link_slot(P, upper, strideU, nb, layer);
            LINK_GUARD(slot >= 0 && slot < P.n_slots, 3);
            if (COUNT) {
                if (atomicAdd(&P.cnt[slot], 1) == 0) {
                    const int g = atomicAdd(&P.counters[0], 1);
                    LINK_GUARD(g < P.g_cap, 4);
                    P.g_node[g] = nb;
                    P.g_layer[g] = layer;
                }
            } else {
                const int p = atomicAdd(&P.fill[slot], 1);
                const long long at = (long long)P.off[slot] + p;
                LINK_GUARD(at >= 0 && at < P.g_cap, 5);
                P.items[at] = t;
            }
        }
    }
}
#ifdef HNSW_HOST_TU // non-template kernels: only the unit that launches them defines them
__global__ void __launch_bounds__(256)
link_offsets_kernel(const int64_t *__restrict__ upper, int strideU, LinkPlan P)
{
    const int G = min(P.counters[0], P.g_cap);
    for (int g = blockIdx.x * 256 + threadIdx.x; g < G; g += gridDim.x * 256) {
        const long long slot = link_slot(P, upper, strideU, P.g_node[g], P.g_layer[g]);
        LINK_GUARD(slot >= 0 && slot < P.n_slots, 6);
        const int c = P.cnt[slot];
        const int start = atomicAdd(&P.counters[1], c);
        LINK_GUARD(c >= 0 && start >= 0 && (long long)start + c <= P.g_cap, 7);
        P.g_start[g] = start;
        P.g_count[g] = c;
        P.off[slot] = start;
    }
}
#endif
// one block per group: its items (batch positions, filed in arbitrary order) become node ids in batch
// order -- repeatedly the smallest position not yet taken; groups are tiny -- and the slot's
// counters return to zero for the next batch
#ifdef HNSW_HOST_TU // non-template kernels: only the unit that launches them defines them
__global__ void __launch_bounds__(64)
link_order_kernel(const SearchJob *__restrict__ jobs, const int64_t *__restrict__ upper, int strideU, int *__restrict__ items_out, LinkPlan P, int bounded)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    if (bounded && (g >= min(P.counters[0], P.g_cap) || P.counters[3] != 0)) return; // grid = an upper bound of the group count
    const int node = P.g_node[g], layer = P.g_layer[g], start = P.g_start[g], n_items = P.g_count[g];
    if (!(node >= 0 && node < P.n_nodes && layer >= 0 && start >= 0 && n_items >= 0 && (long long)start + n_items <= P.g_cap)) {
        atomicCAS(&P.counters[3], 0, 8);
        return;
    }
    int last = -1;
    for (int k = 0; k < n_items; ++k) {
        int best = 0x7fffffff;
        for (int i = lane; i < n_items; i += 64) {
            const int p = P.items[start + i];
            if (p > last && p < best) best = p;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
        if (best >= P.n_jobs) { atomicCAS(&P.counters[3], 0, 9); return; }
        last = best;
        if (lane == 0) items_out[start + k] = ~jobs[best].qref;
    }
    if (lane == 0) {
        const long long slot = link_slot(P, upper, strideU, node, layer);
        P.cnt[slot] = 0;
        P.fill[slot] = 0;
    }
}
#endif

} // namespace hnsw
