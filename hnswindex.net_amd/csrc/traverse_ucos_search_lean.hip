// traverse_ucos_search_lean.hip -- instantiates the lean forms of graph_search_kernel for M_UCOS (launches without visited sets:
// dk_base.h, kFormLean).  Device code: device_kernels.h; the split exists for build time.
#include "device_kernels.h"

namespace hnsw {
HNSW_FOR_EACH_TRAVERSAL_LEAN(HNSW_DEFINE_SEARCH, M_UCOS)
} // namespace hnsw
HNSW_PHASE_BIND(ucos_search_lean)
