// device_kernels.h -- all gfx950 device code of the backend (included by device_backend.hip, which holds the host side, and by
// the traverse_*.hip units, which only instantiate the two big traversal kernel templates -- one metric each -- so that the
// build compiles them in parallel).  The code lives in the dk_*.h parts, in dependency order:
//   dk_base.h            includes, wave_sync / wave_lds_sync, metric ids
//   dk_metric.h          the reference's lane arithmetic (EuclideanMetric.cs / CosineMetric.cs), int8 records, slot_distance_kernel
//   dk_heaps.h           integer keys, the two BinaryHeaps in LDS with the reference's sift rules
//   dk_measure.h         measure passes (rows of one expansion in one memory round trip)
//   dk_search_common.h   LDS carve-up, graph view, visited set, read log, FindEntryAtLayer
//   dk_sorted_top.h      SearchLayer on one sorted register list (the loaded launches), tie rules, no visited set
//   dk_team.h            the latency variants' memory wave and its mailbox
//   dk_pool_top.h        SearchLayer on an unsorted register pool (the latency variants' logic wave)
//   dk_traverse_exact.h  the exact two-heap traversal
//   dk_heuristic.h       RelativeNeighborPruning (with its MFMA Gram-block prefilter)
//   dk_range_finish.h   RangeQuery's order on the device: ranking by counting, the heaps replayed on known distances
//   dk_search_kernels.h  graph_search_kernel, graph_range_kernel
//   dk_insert_kernels.h  graph_insert_search_kernel
//   dk_link.h            the link half of Add, Remove's re-link
//   dk_misc_kernels.h    small kernels
// See device_backend.hip's header comment for what the kernels replace and the numerical contract.
#pragma once
#include "dk_base.h"
#include "dk_metric.h"
#include "dk_heaps.h"
#include "dk_measure.h"
#include "dk_search_common.h"
#include "dk_sorted_top.h"
#include "dk_team.h"
#include "dk_pool_top.h"
#include "dk_traverse_exact.h"
#include "dk_heuristic.h"
#include "dk_search_kernels.h"
#include "dk_range_finish.h"
#include "dk_insert_kernels.h"
#include "dk_link.h"
#include "dk_misc_kernels.h"

namespace hnsw {

// Explicit instantiations of the two traversal kernels live in traverse_<metric>_<search|insert>.hip;
// every other unit only declares them.
// Ten forms per metric and kernel (eighteen until round 5) plus four lean ones: register sets NS in {2, 4, 8} (beams up to 128 / 256 / 512; a beam of
// up to 64 runs in the two-set form), the visited set as a bitset or (graphs above 4M nodes, NS <= 4) a per-wave hash table, and the
// latency variant of each.  What used to be forms of their own: NS = 1 (same code with one register less), NS = 0 (the exact
// two-heap traversal alone: now launch flag 0x200 of the two-set form), hash tables for NS = 8 (such launches keep bitsets).
#define HNSW_FOR_EACH_TRAVERSAL(X, M) \
    X(M, 2, false, kFormPlain) X(M, 4, false, kFormPlain) X(M, 8, false, kFormPlain) \
    X(M, 2, true, kFormPlain) X(M, 4, true, kFormPlain)
// the latency variants: units of their own
#define HNSW_FOR_EACH_TRAVERSAL_LAT(X, M) \
    X(M, 2, false, kFormLat) X(M, 4, false, kFormLat) X(M, 8, false, kFormLat) \
    X(M, 2, true, kFormLat) X(M, 4, true, kFormLat)
// the lean forms of graph_search_kernel (round 5; kFormLean in dk_base.h): beams up to 256 entries, launches without visited sets;
// units of their own.  (The insert kernel has none: measured, its f32 form loses 6 % that way.)
#define HNSW_FOR_EACH_TRAVERSAL_LEAN(X, M) \
    X(M, 2, false, kFormLean) X(M, 4, false, kFormLean) X(M, 2, true, kFormLean) X(M, 4, true, kFormLean)
#define HNSW_SEARCH_SIGNATURE(PREFIX, M, NS, H, LT)                                                                                  \
    PREFIX template __global__ void graph_search_kernel<M, NS, H, LT>(                                                              \
        const float *__restrict__, const double *__restrict__, const float *__restrict__, const double *__restrict__, int,      \
        const int *__restrict__, int, const int64_t *__restrict__, const int *__restrict__, int, const SearchJob *__restrict__,  \
        int, int, ND *__restrict__, int, unsigned *__restrict__, long long, int *__restrict__, int, int, int *__restrict__,      \
        float *__restrict__, int *__restrict__, int *__restrict__, unsigned long long *__restrict__, int, int, int *__restrict__, int, \
        const int *__restrict__);
#define HNSW_INSERT_SIGNATURE(PREFIX, M, NS, H, LT)                                                                                  \
    PREFIX template __global__ void graph_insert_search_kernel<M, NS, H, LT>(                                                       \
        const float *__restrict__, const double *__restrict__, int, const int *__restrict__, int, const int64_t *__restrict__,   \
        const int *__restrict__, int, const SearchJob *__restrict__, int, int, ND *__restrict__, int, int, unsigned *__restrict__, \
        long long, int *__restrict__, int, int *__restrict__, int *__restrict__, int *__restrict__, int *__restrict__, int,      \
        int *__restrict__, unsigned long long *__restrict__, int, int, int *__restrict__, int, const int *__restrict__, int *__restrict__, int);
#define HNSW_DECLARE_TRAVERSAL(M, NS, H, LT) HNSW_SEARCH_SIGNATURE(extern, M, NS, H, LT) HNSW_INSERT_SIGNATURE(extern, M, NS, H, LT)
#define HNSW_DECLARE_SEARCH(M, NS, H, LT) HNSW_SEARCH_SIGNATURE(extern, M, NS, H, LT)
#define HNSW_DEFINE_TRAVERSAL(M, NS, H, LT) HNSW_SEARCH_SIGNATURE(, M, NS, H, LT) HNSW_INSERT_SIGNATURE(, M, NS, H, LT)
#define HNSW_DEFINE_SEARCH(M, NS, H, LT) HNSW_SEARCH_SIGNATURE(, M, NS, H, LT)
#define HNSW_DEFINE_INSERT(M, NS, H, LT) HNSW_INSERT_SIGNATURE(, M, NS, H, LT)

} // namespace hnsw
