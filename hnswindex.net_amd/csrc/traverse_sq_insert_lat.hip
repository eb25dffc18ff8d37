// traverse_sq_insert_lat.hip -- instantiates the latency variants of graph_insert_search_kernel for M_SQ (launches that do
// not fill the chip: device_kernels.h, LAT).  Device code: device_kernels.h; the split exists for build time.
#include "device_kernels.h"

namespace hnsw {
HNSW_FOR_EACH_TRAVERSAL_LAT(HNSW_DEFINE_INSERT, M_SQ)
} // namespace hnsw
HNSW_PHASE_BIND(sq_insert_lat)
