// traverse_cos_search.hip -- instantiates graph_search_kernel for M_COS (every register-set count,
// both visited-set representations).  Device code: device_kernels.h; the split exists for build time.
#include "device_kernels.h"

namespace hnsw {
HNSW_FOR_EACH_TRAVERSAL(HNSW_DEFINE_SEARCH, M_COS)
} // namespace hnsw
HNSW_PHASE_BIND(cos_search)
