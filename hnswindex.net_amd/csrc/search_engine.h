// search_engine.h -- lock-step host driver.
//
// The reference evaluates one distance at a time inside its traversal loops
// (GraphNavigator.SearchLayer / FindEntryAtLayer, Heuristic.RelativeNeighborPruning,
// GraphConnector.PruneOverflow).  Here many independent traversals ("jobs") advance
// together: each step, every live job states which candidate rows it needs measured
// against which vector (one *task*: <= stride ids + one query reference), one kernel
// launch evaluates every task of every job, and each job then consumes its distances in
// the reference's own order.  The traversal logic itself stays on the host, as in the
// reference; only its distance evaluations moved.
//
// Two slot groups ping-pong so that the GPU measures one group's tasks while the host
// threads consume/prepare the other group's.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "device_backend.h"
#include "host_structs.h"

namespace hnsw {

// What a job sees of its slot for one step.
struct SlotIO {
    int *ids;          // out: candidate row ids (capacity = stride)
    int *cnt;          // out: number of ids
    int *qidx;         // out: >= 0 resident query index, < 0: ~row id
    const float *dist; // in (consume): distances, aligned with ids
    int stride;
};

// Reusable per-slot scratch so that jobs allocate nothing per search.
struct SlotScratch {
    Visited visited;
    BinaryHeap<FartherFirst> top;
    BinaryHeap<CloserFirst> cand;
    std::vector<NodeDist> tmp;
    std::vector<int> accepted;
};

struct Job {
    virtual ~Job() = default;
    // Emit the next task into io; return false when the job has finished (no task emitted).
    virtual bool prepare(SlotIO &io, SlotScratch &sc) = 0;
    // Consume the distances of the task emitted by the last prepare().
    virtual void consume(const SlotIO &io, SlotScratch &sc) = 0;
};

struct JobSource {
    virtual ~JobSource() = default;
    // Thread-safe.  Returns nullptr when no work is left.
    virtual Job *acquire(SlotScratch &sc) = 0;
    // Called once, by the thread that ran the job, after prepare() returned false.
    virtual void release(Job *job, SlotScratch &sc) = 0;
};

class LockStepEngine {
public:
    LockStepEngine(Device *dev, int nslots, int stride, int nthreads);
    ~LockStepEngine();
    bool ok() const { return ok_; }
    int stride() const { return stride_; }
    // Runs until the source is exhausted and every job has finished.  njobs_hint sizes the
    // number of slots / threads actually used (a single sequential insert uses 1 + 1).
    bool run(JobSource &src, long long njobs_hint);

private:
    struct Slot {
        Job *job = nullptr;
        bool awaiting = false;
        SlotScratch scratch;
    };
    void worker_main(int t);
    void half_step(int t, int g);
    void barrier();

    // The engine reaches the GPU only through the inner C ABI (include/hnsw_mi355x.h:
    // hnswdev_step_buffers / _submit / _wait) -- the same three calls a C# host P/Invokes.
    void *ctx_;  // hnswdev context
    int half_;   // slots per group
    int stride_, nthreads_;
    bool ok_ = false;
    int *rec_[2] = {nullptr, nullptr};     // the context's pinned step records, set 0 / 1
    float *dist_[2] = {nullptr, nullptr};  // ... and distances
    int rec_stride_ = 0;
    std::vector<Slot> slots_[2];

    // per-run state; workers read the descriptor they were woken for under mu_ (run_gen_)
    JobSource *src_ = nullptr;
    int used_half_ = 0, used_threads_ = 1;
    std::vector<uint64_t> t_evals_;
    std::vector<int> t_active_;
    std::vector<int> t_maxslot_;

    // worker pool
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    uint64_t run_gen_ = 0;
    bool shutting_down_ = false;
    std::atomic<int> bar_count_{0};
    std::atomic<uint32_t> bar_gen_{0};
    std::atomic<int> left_count_{0};
    std::atomic<bool> done_{false};
    std::atomic<bool> failed_{false};
};

} // namespace hnsw
