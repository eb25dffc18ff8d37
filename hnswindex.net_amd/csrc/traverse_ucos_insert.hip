// traverse_ucos_insert.hip -- instantiates graph_insert_search_kernel for M_UCOS (every register-set count,
// both visited-set representations).  Device code: device_kernels.h; the split exists for build time.
#include "device_kernels.h"

namespace hnsw {
HNSW_FOR_EACH_TRAVERSAL(HNSW_DEFINE_INSERT, M_UCOS)
} // namespace hnsw
HNSW_PHASE_BIND(ucos_insert)
