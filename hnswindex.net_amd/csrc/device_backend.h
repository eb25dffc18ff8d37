// device_backend.h -- internal C++ face of the gfx950 distance backend.
// The C ABI (include/hnsw_mi355x.h, hnswdev_*) and the host driver (search_engine.cpp,
// hnsw_index.cpp) both sit on this class.  Nothing here computes a distance on the CPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/hnsw_mi355x.h"

namespace hnsw {

void set_dev_error(const std::string &msg);
std::string get_dev_error();

// One lock-step "step" worth of work for up to nslots concurrent searches.
// All four arrays live in pinned, device-mapped host memory: the host driver writes
// cnt/qidx/ids, the kernel reads them over PCIe and writes dist back; nothing is staged.
//   slot s evaluates metric(row[ids[s*stride + c]], Q(s)) for c < cnt[s]
//   Q(s) = resident query  qidx[s]            if qidx[s] >= 0
//        = stored row      ~qidx[s]           if qidx[s] <  0   (id<->id distances)
struct StepBuffers {
    int nslots = 0, stride = 0;
    int *cnt = nullptr, *qidx = nullptr, *ids = nullptr;
    float *dist = nullptr;
    int *d_cnt = nullptr, *d_qidx = nullptr, *d_ids = nullptr;
    float *d_dist = nullptr;
    void *done = nullptr;      // hipEvent_t recorded after the kernel
    void *t0 = nullptr, *t1 = nullptr; // hipEvent_t pair when profiling
    bool timed = false;
    uint64_t evals = 0;
};

class Device {
public:
    static Device *create(int device, int dim, int metric, long long capacity);
    ~Device();

    int dim() const { return dim_; }
    int metric() const { return metric_; }
    long long capacity() const { return capacity_; }

    bool reserve(long long capacity);
    bool upload_rows(int first_id, int n, const float *rows);
    bool download_rows(int first_id, int n, float *rows);
    // Replaces the resident query set (nq x dim); norms for cosine computed on device.
    bool set_queries(const float *queries, int nq);

    StepBuffers *alloc_step(int nslots, int stride);
    void free_step(StepBuffers *sb);
    // Asynchronous: one kernel over slots [0, nslots_used).  `evals` = sum of cnt (for stats).
    bool launch_step(StepBuffers *sb, int nslots_used, uint64_t evals);
    bool wait_step(StepBuffers *sb);
    bool sync();
    // Makes this context's HIP device current on the calling thread.
    bool bind_thread() { return bind(); }

    // C-ABI conveniences (synchronous; validate ids on the host before launching).
    bool dist_query_batch(const float *queries, int nq, const int *offsets, const int *ids, float *out);
    bool dist_pair_batch(const int *a, const int *b, int n, float *out);

    void set_profiling(bool on) { profiling_ = on; }
    void get_stats(hnswdev_stats *out);
    void reset_stats();

private:
    Device() = default;
    bool bind();
    int device_ = 0, dim_ = 0, metric_ = 0;
    long long capacity_ = 0;
    long long n_rows_hw_ = 0; // high-water mark of uploaded rows (id validation)
    float *d_rows_ = nullptr;
    double *d_row_sn_ = nullptr; // cosine: sqrt((double)|row|^2_f32)
    float *d_queries_ = nullptr;
    double *d_q_sn_ = nullptr;
    long long q_capacity_ = 0, n_queries_ = 0;
    void *stream_ = nullptr;
    bool profiling_ = false;
    hnswdev_stats stats_{};
};

} // namespace hnsw
