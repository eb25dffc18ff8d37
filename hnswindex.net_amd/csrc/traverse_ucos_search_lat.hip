// traverse_ucos_search_lat.hip -- instantiates the latency variants of graph_search_kernel for M_UCOS (launches that do
// not fill the chip: device_kernels.h, LAT).  Device code: device_kernels.h; the split exists for build time.
#include "device_kernels.h"

namespace hnsw {
HNSW_FOR_EACH_TRAVERSAL_LAT(HNSW_DEFINE_SEARCH, M_UCOS)
} // namespace hnsw
HNSW_PHASE_BIND(ucos_search_lat)
