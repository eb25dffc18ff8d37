// search_engine.cpp -- see search_engine.h.
#include "search_engine.h"

#include <algorithm>
#include <immintrin.h>

namespace hnsw {

LockStepEngine::LockStepEngine(Device *dev, int nslots, int stride, int nthreads)
    : ctx_(dev), stride_(stride), nthreads_(std::max(1, nthreads))
{
    half_ = std::max(1, nslots / 2);
    rec_stride_ = stride_ + 2; // [cnt, qidx, ids[stride]] per slot (include/hnsw_mi355x.h)
    for (int g = 0; g < 2; ++g)
        if (hnswdev_step_buffers(ctx_, g, half_, stride_, &rec_[g], &dist_[g]) != 0) return;
    slots_[0].resize((size_t)half_);
    slots_[1].resize((size_t)half_);
    t_evals_.assign((size_t)nthreads_, 0);
    t_active_.assign((size_t)nthreads_, 0);
    t_maxslot_.assign((size_t)nthreads_, 0);
    for (int t = 1; t < nthreads_; ++t) workers_.emplace_back([this, t] { worker_main(t); });
    ok_ = true;
}

LockStepEngine::~LockStepEngine()
{
    {
        std::lock_guard<std::mutex> lk(mu_);
        shutting_down_ = true;
        ++run_gen_;
    }
    cv_.notify_all();
    for (auto &w : workers_) w.join();
    // the step buffers belong to the context
}

void LockStepEngine::barrier()
{
    const int n = used_threads_;
    if (n == 1) return;
    const uint32_t gen = bar_gen_.load(std::memory_order_acquire);
    if (bar_count_.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
        bar_count_.store(0, std::memory_order_relaxed);
        bar_gen_.fetch_add(1, std::memory_order_release);
    } else {
        int spins = 0;
        while (bar_gen_.load(std::memory_order_acquire) == gen) {
            _mm_pause();
            if (++spins > 4000) { std::this_thread::yield(); spins = 0; }
        }
    }
}

// consume (if results are pending) then prepare every slot of group g owned by thread t
void LockStepEngine::half_step(int t, int g)
{
    const int lo = (int)((long long)t * used_half_ / used_threads_);
    const int hi = (int)((long long)(t + 1) * used_half_ / used_threads_);
    uint64_t evals = 0;
    int active = 0, maxslot = 0;
    for (int s = lo; s < hi; ++s) {
        Slot &sl = slots_[g][(size_t)s];
        int *rec = rec_[g] + (size_t)s * rec_stride_;
        SlotIO io{rec + 2, rec, rec + 1, dist_[g] + (size_t)s * stride_, stride_};
        if (sl.job && sl.awaiting) {
            sl.job->consume(io, sl.scratch);
            sl.awaiting = false;
        }
        *io.cnt = 0;
        for (;;) {
            if (!sl.job) {
                sl.job = src_->acquire(sl.scratch);
                if (!sl.job) break;
            }
            if (sl.job->prepare(io, sl.scratch)) {
                sl.awaiting = true;
                evals += (uint64_t)*io.cnt;
                ++active;
                maxslot = s + 1;
                break;
            }
            src_->release(sl.job, sl.scratch);
            sl.job = nullptr;
            *io.cnt = 0;
        }
    }
    t_evals_[(size_t)t] = evals;
    t_active_[(size_t)t] = active;
    t_maxslot_[(size_t)t] = maxslot;
}

void LockStepEngine::worker_main(int t)
{
    uint64_t seen = 0;
    for (;;) {
        bool take_part;
        {
            // run() publishes used_threads_ / used_half_ / src_ and every reset under mu_, together
            // with ++run_gen_; a worker acts on exactly the generation it observes here.  run() does
            // not return (and so cannot start the next generation) before every participating worker
            // has left the loop below, and a non-participant touches nothing outside this block.
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return run_gen_ != seen; });
            seen = run_gen_;
            if (shutting_down_) return;
            take_part = t < used_threads_;
        }
        if (!take_part) continue;
        int g = 0;
        for (;;) {
            half_step(t, g);
            barrier();
            barrier(); // thread 0 launches / waits between the two barriers
            if (done_.load(std::memory_order_acquire)) break;
            g ^= 1;
        }
        left_count_.fetch_add(1, std::memory_order_acq_rel); // this worker has left the loop
    }
}

bool LockStepEngine::run(JobSource &src, long long njobs_hint)
{
    if (!ok_) return false;
    if (njobs_hint <= 0) return true;
    // the sets may have been re-created since (another client of the same context grew them)
    for (int g = 0; g < 2; ++g)
        if (hnswdev_step_buffers(ctx_, g, half_, stride_, &rec_[g], &dist_[g]) != 0) return false;
    const int use_half = (int)std::min<long long>(half_, std::max<long long>(1, (njobs_hint + 1) / 2));
    int use_threads = (int)std::min<long long>(nthreads_, std::max<long long>(1, njobs_hint / 48));
    use_threads = std::min(use_threads, use_half);
    {
        // the whole run descriptor is published in one critical section with the generation bump
        std::lock_guard<std::mutex> lk(mu_);
        src_ = &src;
        used_half_ = use_half;
        used_threads_ = use_threads;
        done_.store(false, std::memory_order_release);
        failed_.store(false, std::memory_order_release);
        left_count_.store(0, std::memory_order_release);
        bar_count_.store(0, std::memory_order_release);
        for (int g = 0; g < 2; ++g) {
            // slots beyond used_half_ stay idle with cnt == 0 from allocation/previous runs
            for (int s = 0; s < used_half_; ++s) { slots_[g][(size_t)s].job = nullptr; slots_[g][(size_t)s].awaiting = false; }
        }
        ++run_gen_;
    }
    if (used_threads_ > 1) cv_.notify_all();
    bool pend[2] = {false, false};
    int active[2] = {0, 0};
    int g = 0;
    for (;;) {
        half_step(0, g);
        barrier();
        uint64_t evals = 0;
        int act = 0, maxslot = 0;
        for (int t = 0; t < used_threads_; ++t) {
            evals += t_evals_[(size_t)t];
            act += t_active_[(size_t)t];
            maxslot = std::max(maxslot, t_maxslot_[(size_t)t]);
        }
        active[g] = act;
        if (!failed_.load()) {
            if (evals > 0) {
                if (hnswdev_step_submit(ctx_, g, maxslot) != 0) failed_.store(true);
                else pend[g] = true;
            }
            if (pend[g ^ 1]) {
                if (hnswdev_step_wait(ctx_, g ^ 1) != 0) failed_.store(true);
                pend[g ^ 1] = false;
            }
        }
        bool fin = failed_.load() || (active[0] == 0 && active[1] == 0);
        done_.store(fin, std::memory_order_release);
        barrier();
        if (fin) break;
        g ^= 1;
    }
    if (used_threads_ > 1) {
        // wait until every worker has observed done_ and left its loop
        while (left_count_.load(std::memory_order_acquire) < used_threads_ - 1) _mm_pause();
    }
    if (pend[0]) (void)hnswdev_step_wait(ctx_, 0);
    if (pend[1]) (void)hnswdev_step_wait(ctx_, 1);
    return !failed_.load();
}

} // namespace hnsw
