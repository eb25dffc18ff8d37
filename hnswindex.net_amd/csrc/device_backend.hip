// device_backend.hip -- gfx950 (MI355X / CDNA4) candidate-distance kernels and the device
// context that owns the HBM-resident vector matrix.
//
// What the kernels replace (all citations relative to /root/reference/):
//   SquaredEuclideanMetric.Compute  src/HNSWIndex/Metrics/EuclideanMetric.cs:11-60
//   CosineMetric.UnitCompute        src/HNSWIndex/Metrics/CosineMetric.cs:95-142
//   CosineMetric.Compute            src/HNSWIndex/Metrics/CosineMetric.cs:10-92
// as invoked through GraphData.Distance (src/HNSWIndex/GraphData.cs:255-277) from the
// search / link loops.
//
// Numerical contract: the reference's AVX branch keeps EIGHT partial sums;
// element i feeds partial (i mod 8) in increasing i; L2 uses fma(d,d,acc) with d = a-b;
// dot/norm use acc + (a*b) (two roundings); the eight partials collapse through a fixed
// tree -- L2 ((p0+p4)+(p1+p5))+((p2+p6)+(p3+p7)), cosine family ((p0+p4)+(p2+p6))+((p1+p5)+(p3+p7))
// -- and a scalar mul-then-add tail handles dim % 8.  The kernels reproduce that order
// exactly, so distances are bit-identical to the CPU path and every float compare in the
// traversal branches the same way.  Built with -ffp-contract=off; every fused operation is
// written as __builtin_fmaf.
//
// Mapping to the wavefront: 8 lanes own the 8 partials of one candidate row, so a wave64
// evaluates 8 candidates at a time (up to 4 rows per lane group in one memory round trip); the
// collapse tree is three cross-lane adds.  Wider per-lane loads and lane-ring variants were
// measured no faster (tools/kbench.hip): this mapping gathers random 512-B rows at 4.2-4.9 TB/s.
//
// Contents: (1) slot_distance_kernel / pair_distance_kernel -- batched distances for a host that
// keeps the traversal (the hnswdev_dist_* entry points, the lock-step engine);  (2) the
// graph-resident traversals -- traverse_sorted (one sorted list in registers; the default) and
// traverse (the reference's two heaps in LDS; exact under equal distances), wrapped by the
// persistent graph_search_kernel (KnnQuery) and graph_insert_search_kernel (Add, search half +
// RelativeNeighborPruning);  (3) Add's link half -- link_plan / link_offsets / link_order
// (grouping of the back-edge appends on the device) and graph_link_kernel (appends and
// PruneOverflow with the tested-prefix shortcut);  (4) class Device: HBM matrix, graph mirror,
// per-wave scratch, launches.  DESIGN.md section 3 has the reasoning and the measurements.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "device_backend.h"
#include "diag.h"
#define HNSW_HOST_TU
#include "device_kernels.h"
#include "range_replay.h"

namespace hnsw {

// ------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------
static std::mutex g_err_mu;
static std::string g_err;
static thread_local Device *tl_ctx = nullptr;
ErrorScope::ErrorScope(Device *d) : prev(tl_ctx) { tl_ctx = d; }
ErrorScope::~ErrorScope() { tl_ctx = prev; }
void set_dev_error(const std::string &msg)
{
    if (tl_ctx) tl_ctx->set_error(msg);
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = msg;
}
std::string get_dev_error()
{
    std::lock_guard<std::mutex> lk(g_err_mu);
    return g_err;
}

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            set_dev_error(std::string(#expr) + ": " + hipGetErrorString(_e));                     \
            return false;                                                                         \
        }                                                                                         \
    } while (0)

template <class T>
static bool grow_dev(T **p, size_t *cap, size_t need)
{
    if (need <= *cap) return true;
    if (*p) HIP_OK(hipFree(*p));
    *p = nullptr;
    HIP_OK(hipMalloc(p, sizeof(T) * need));
    *cap = need;
    return true;
}

#ifdef EXP_PHASE_CLOCKS
HNSW_PHASE_BIND(backend)
extern "C" {
hipError_t hnsw_phase_bind_sq_insert(unsigned long long *); hipError_t hnsw_phase_bind_sq_search(unsigned long long *);
hipError_t hnsw_phase_bind_cos_insert(unsigned long long *); hipError_t hnsw_phase_bind_cos_search(unsigned long long *);
hipError_t hnsw_phase_bind_ucos_insert(unsigned long long *); hipError_t hnsw_phase_bind_ucos_search(unsigned long long *);
hipError_t hnsw_phase_bind_i8_insert(unsigned long long *); hipError_t hnsw_phase_bind_i8_search(unsigned long long *);
hipError_t hnsw_phase_bind_sq_insert_lat(unsigned long long *); hipError_t hnsw_phase_bind_sq_search_lat(unsigned long long *);
hipError_t hnsw_phase_bind_cos_insert_lat(unsigned long long *); hipError_t hnsw_phase_bind_cos_search_lat(unsigned long long *);
hipError_t hnsw_phase_bind_ucos_insert_lat(unsigned long long *); hipError_t hnsw_phase_bind_ucos_search_lat(unsigned long long *);
hipError_t hnsw_phase_bind_i8_insert_lat(unsigned long long *); hipError_t hnsw_phase_bind_i8_search_lat(unsigned long long *);
hipError_t hnsw_phase_bind_sq_search_lean(unsigned long long *);
hipError_t hnsw_phase_bind_cos_search_lean(unsigned long long *);
hipError_t hnsw_phase_bind_ucos_search_lean(unsigned long long *);
hipError_t hnsw_phase_bind_i8_search_lean(unsigned long long *);
}
static unsigned long long *g_phase_buf = nullptr; // one buffer per process (diagnostic builds run one index at a time)
static bool phase_bind_all()
{
    if (g_phase_buf) return true;
    if (hipMalloc(&g_phase_buf, sizeof(unsigned long long) * hnsw::kPhaseWords) != hipSuccess) return false;
    (void)hipMemset(g_phase_buf, 0, sizeof(unsigned long long) * hnsw::kPhaseWords);
    bool ok = hnsw_phase_bind_backend(g_phase_buf) == hipSuccess;
#ifndef HNSW_SINGLE_TU
    ok = ok && hnsw_phase_bind_sq_insert(g_phase_buf) == hipSuccess && hnsw_phase_bind_sq_search(g_phase_buf) == hipSuccess &&
         hnsw_phase_bind_cos_insert(g_phase_buf) == hipSuccess && hnsw_phase_bind_cos_search(g_phase_buf) == hipSuccess &&
         hnsw_phase_bind_ucos_insert(g_phase_buf) == hipSuccess && hnsw_phase_bind_ucos_search(g_phase_buf) == hipSuccess &&
         hnsw_phase_bind_i8_insert(g_phase_buf) == hipSuccess && hnsw_phase_bind_i8_search(g_phase_buf) == hipSuccess &&
         hnsw_phase_bind_sq_insert_lat(g_phase_buf) == hipSuccess && hnsw_phase_bind_sq_search_lat(g_phase_buf) == hipSuccess &&
         hnsw_phase_bind_cos_insert_lat(g_phase_buf) == hipSuccess && hnsw_phase_bind_cos_search_lat(g_phase_buf) == hipSuccess &&
         hnsw_phase_bind_ucos_insert_lat(g_phase_buf) == hipSuccess && hnsw_phase_bind_ucos_search_lat(g_phase_buf) == hipSuccess &&
         hnsw_phase_bind_i8_insert_lat(g_phase_buf) == hipSuccess && hnsw_phase_bind_i8_search_lat(g_phase_buf) == hipSuccess &&
         hnsw_phase_bind_sq_search_lean(g_phase_buf) == hipSuccess &&
         hnsw_phase_bind_cos_search_lean(g_phase_buf) == hipSuccess &&
         hnsw_phase_bind_ucos_search_lean(g_phase_buf) == hipSuccess &&
         hnsw_phase_bind_i8_search_lean(g_phase_buf) == hipSuccess;
#endif
    return ok;
}
static bool phase_read(unsigned long long *h) { return g_phase_buf && hipMemcpy(h, g_phase_buf, sizeof(unsigned long long) * hnsw::kPhaseWords, hipMemcpyDeviceToHost) == hipSuccess; }
static void phase_zero() { if (g_phase_buf) (void)hipMemset(g_phase_buf, 0, sizeof(unsigned long long) * hnsw::kPhaseWords); }
static void phase_report(const char *when)
{
    unsigned long long w[hnsw::kPhaseWords] = {0};
    if (!phase_read(w)) return;
    const unsigned long long *h = w, *hl = w + 12, *hx = w + 24;
    double tot = 0, tl = 0;
    for (int i = 0; i < 6; ++i) tot += (double)h[i];
    for (int i = 0; i < 5; ++i) tl += (double)hl[i];
    if (hl[7])
        fprintf(stderr, "[phase clocks, link] stage %.1f%% measure %.1f%% sort %.1f%% heuristic %.1f%% rest %.1f%% | appends %llu, prunes %llu, cycles/prune %.0f\n",
                100 * hl[0] / tl, 100 * hl[1] / tl, 100 * hl[2] / tl, 100 * hl[3] / tl, 100 * hl[4] / tl, hl[6], hl[7], tl / (double)std::max(1ull, hl[7]));
    if (h[9]) fprintf(stderr, "[phase clocks, insert] RelativeNeighborPruning %.1f%% of the insert jobs' cycles; heuristic cycles %.0f, job cycles %.0f\n", 100.0 * (double)h[8] / (double)h[9], (double)h[8], (double)h[9]);
    if (!h[7]) return;
    const double e = (double)h[7];
    fprintf(stderr, "[phase clocks, %s] descent %.1f%% pop %.1f%% list %.1f%% visited %.1f%% rows %.1f%% push %.1f%% | expansions %llu, prefetch hits %.1f%%, cycles/expansion %.0f\n", when,
            100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot, 100 * h[4] / tot, 100 * h[5] / tot, h[7], 100.0 * h[6] / e, tot / e);
    fprintf(stderr, "[phase clocks, per expansion] descent %.0f pop %.0f list %.0f visited %.0f rows %.0f (before %.0f, measure %.0f, after %.0f) push %.0f (checks %.0f, next-pop guess %.0f, merge %.0f, single inserts %.0f) | merges %.2f, single inserts %.2f, candidates passing %.2f\n",
            h[0] / e, h[1] / e, h[2] / e, h[3] / e, h[4] / e, hx[0] / e, hx[1] / e, hx[2] / e, h[5] / e, hx[8] / e, hx[9] / e, hx[10] / e, hx[11] / e, hx[3] / e, hx[4] / e, hx[5] / e);
    if (hx[13]) fprintf(stderr, "[phase clocks, memory wave, per expansion] waiting for a request %.0f, list %.0f, marks + rows + distances %.0f, keys + answer %.0f\n",
                        hx[12] / e, hx[13] / e, hx[14] / e, hx[15] / e);
}
#endif

#ifdef HNSW_SINGLE_TU
// one translation unit (diagnostic builds)
HNSW_FOR_EACH_TRAVERSAL(HNSW_DEFINE_TRAVERSAL, M_SQ)
HNSW_FOR_EACH_TRAVERSAL(HNSW_DEFINE_TRAVERSAL, M_COS)
HNSW_FOR_EACH_TRAVERSAL(HNSW_DEFINE_TRAVERSAL, M_UCOS)
HNSW_FOR_EACH_TRAVERSAL(HNSW_DEFINE_TRAVERSAL, M_I8)
HNSW_FOR_EACH_TRAVERSAL_LAT(HNSW_DEFINE_TRAVERSAL, M_SQ)
HNSW_FOR_EACH_TRAVERSAL_LAT(HNSW_DEFINE_TRAVERSAL, M_COS)
HNSW_FOR_EACH_TRAVERSAL_LAT(HNSW_DEFINE_TRAVERSAL, M_UCOS)
HNSW_FOR_EACH_TRAVERSAL_LAT(HNSW_DEFINE_TRAVERSAL, M_I8)
HNSW_FOR_EACH_TRAVERSAL_LEAN(HNSW_DEFINE_SEARCH, M_SQ)
HNSW_FOR_EACH_TRAVERSAL_LEAN(HNSW_DEFINE_SEARCH, M_COS)
HNSW_FOR_EACH_TRAVERSAL_LEAN(HNSW_DEFINE_SEARCH, M_UCOS)
HNSW_FOR_EACH_TRAVERSAL_LEAN(HNSW_DEFINE_SEARCH, M_I8)
#else
HNSW_FOR_EACH_TRAVERSAL(HNSW_DECLARE_TRAVERSAL, M_SQ)
HNSW_FOR_EACH_TRAVERSAL(HNSW_DECLARE_TRAVERSAL, M_COS)
HNSW_FOR_EACH_TRAVERSAL(HNSW_DECLARE_TRAVERSAL, M_UCOS)
HNSW_FOR_EACH_TRAVERSAL(HNSW_DECLARE_TRAVERSAL, M_I8)
HNSW_FOR_EACH_TRAVERSAL_LAT(HNSW_DECLARE_TRAVERSAL, M_SQ)
HNSW_FOR_EACH_TRAVERSAL_LAT(HNSW_DECLARE_TRAVERSAL, M_COS)
HNSW_FOR_EACH_TRAVERSAL_LAT(HNSW_DECLARE_TRAVERSAL, M_UCOS)
HNSW_FOR_EACH_TRAVERSAL_LAT(HNSW_DECLARE_TRAVERSAL, M_I8)
HNSW_FOR_EACH_TRAVERSAL_LEAN(HNSW_DECLARE_SEARCH, M_SQ)
HNSW_FOR_EACH_TRAVERSAL_LEAN(HNSW_DECLARE_SEARCH, M_COS)
HNSW_FOR_EACH_TRAVERSAL_LEAN(HNSW_DECLARE_SEARCH, M_UCOS)
HNSW_FOR_EACH_TRAVERSAL_LEAN(HNSW_DECLARE_SEARCH, M_I8)
#endif

// ------------------------------------------------------------------------------------
// host side of the context
// ------------------------------------------------------------------------------------
// C-ABI graph staging (hnswdev_graph_*): the host graph flattened layer by layer
struct Device::HostGraphStage {
    int n = 0, M = 0, stride0 = 0, strideU = 0, top = 0;
    std::vector<int> level, adj0, pool;
    std::vector<int64_t> upper;
};

static inline hipStream_t S(void *p) { return (hipStream_t)p; }

bool Device::bind()
{
    HIP_OK(hipSetDevice(device_));
    return true;
}

Device *Device::create(int device, int dim, int metric, long long capacity)
{
    if (dim <= 0 || metric < 0 || metric > 3 || capacity < 0) {
        set_dev_error("hnswdev_create: bad argument");
        return nullptr;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_dev_error(std::string("no HIP device available (") + hipGetErrorString(e) +
                      "): this library has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        set_dev_error("hnswdev_create: device ordinal out of range");
        return nullptr;
    }
    Device *d = new Device();
    d->device_ = device;
    d->dim_ = dim;
    // words per stored row: the floats themselves, or the int8 record (data words + scale + sumsq, a
    // multiple of 16 words = 64 B; device_kernels.h "int8 rows")
    d->pitch_ = metric == M_I8 ? (((dim + 3) / 4 + 2 + 15) & ~15) : dim;
    d->metric_ = metric;
    // algorithmic bytes per evaluation: the row's elements, plus the 4-byte scale for int8
    d->stats_.row_bytes = metric == M_I8 ? (uint64_t)dim + 4u : (uint64_t)dim * sizeof(float);
    auto fail = [&]() -> Device * { delete d; return nullptr; };
    if (hipSetDevice(device) != hipSuccess) { set_dev_error("hipSetDevice failed"); return fail(); }
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { set_dev_error("hipStreamCreate failed"); return fail(); }
    d->stream_ = st;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) d->num_cu_ = cus;
    }
    if (!d->reserve(capacity > 0 ? capacity : 1)) return fail();
#ifdef EXP_PHASE_CLOCKS
    if (!phase_bind_all()) { set_dev_error("phase-clock buffer: bind failed"); return fail(); }
#endif
    return d;
}

Device *Device::create_view(Device *primary)
{
    if (!primary) return nullptr;
    Device *d = new Device();
    d->is_view_ = true;
    d->device_ = primary->device_;
    d->dim_ = primary->dim_;
    d->pitch_ = primary->pitch_;
    d->metric_ = primary->metric_;
    d->stats_.row_bytes = primary->stats_.row_bytes;
    d->num_cu_ = primary->num_cu_;
    if (hipSetDevice(d->device_) != hipSuccess) { set_dev_error("hipSetDevice failed"); delete d; return nullptr; }
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { set_dev_error("hipStreamCreate failed"); delete d; return nullptr; }
    d->stream_ = st;
    d->rebind(primary);
    return d;
}

void Device::rebind(const Device *p)
{
    d_rows_ = p->d_rows_; d_row_sn_ = p->d_row_sn_; capacity_ = p->capacity_; n_rows_hw_ = p->n_rows_hw_;
    g_adj0_ = p->g_adj0_; g_level_ = p->g_level_; g_upper_ = p->g_upper_; g_pool_ = p->g_pool_;
    g_tested0_ = p->g_tested0_; g_testedU_ = p->g_testedU_;
    g_n_ = p->g_n_; g_cap_n_ = p->g_cap_n_; g_pool_cap_ = p->g_pool_cap_; g_stride0_ = p->g_stride0_; g_strideU_ = p->g_strideU_;
}

Device::~Device()
{
    if (hipSetDevice(device_) != hipSuccess) return;
    if (is_view_) { // borrowed: the primary frees them
        d_rows_ = nullptr; d_row_sn_ = nullptr;
        g_adj0_ = nullptr; g_level_ = nullptr; g_upper_ = nullptr; g_pool_ = nullptr; g_tested0_ = nullptr; g_testedU_ = nullptr;
    }
    if (stream_) { (void)hipStreamSynchronize(S(stream_)); (void)hipStreamDestroy(S(stream_)); }
    for (int i = 0; i < 8; ++i) { if (up_pin_[i]) (void)hipHostFree(up_pin_[i]); if (up_ev_[i]) (void)hipEventDestroy((hipEvent_t)up_ev_[i]); }
    if (h_ready_) (void)hipHostFree(h_ready_);
    if (copy_stream_) (void)hipStreamDestroy(S(copy_stream_));
    if (bg_.th.joinable()) bg_.th.join();
    if (bg_.stream) (void)hipStreamDestroy(S(bg_.stream));
    for (int i = 0; i < 2; ++i) { if (bg_.pin[i]) (void)hipHostFree(bg_.pin[i]); if (bg_.ev[i]) (void)hipEventDestroy((hipEvent_t)bg_.ev[i]); }
    for (StepBuffers *&sb : abi_sb_) { if (sb) free_step(sb); sb = nullptr; }
    if (d_guard_) (void)hipFree(d_guard_);
    if (q_stage_) (void)hipFree(q_stage_);
    if (pair_dev_) (void)hipFree(pair_dev_);
    if (d_rows_) (void)hipFree(d_rows_);
    if (d_row_sn_) (void)hipFree(d_row_sn_);
    if (d_queries_) (void)hipFree(d_queries_);
    if (d_q_sn_) (void)hipFree(d_q_sn_);
#ifdef EXP_PHASE_CLOCKS
    phase_report("teardown");
#endif
    for (void *p : {(void *)g_adj0_, (void *)g_level_, (void *)g_upper_, (void *)g_pool_, (void *)g_tested0_, (void *)g_testedU_, (void *)s_visited_, (void *)s_jobs_,
                    (void *)s_hits_, (void *)s_cnt_, (void *)s_flag_, (void *)s_jobctr_, (void *)s_vistab_, (void *)lp_slot_[0], (void *)lp_slot_[1], (void *)lp_slot_[2], (void *)lp_grp_[0], (void *)lp_grp_[1], (void *)lp_grp_[2], (void *)lp_grp_[3], (void *)lp_grp_[4], (void *)lp_grp_[5], (void *)lp_counters_, (void *)s_evals_, (void *)s_sel_, (void *)s_lcnt_, (void *)s_selU_, (void *)s_cntU_, (void *)s_iflag_, (void *)s_lk_[0], (void *)s_lk_[1], (void *)s_lk_[2], (void *)s_lk_[3], (void *)s_lk_[4], (void *)s_spill_, (void *)s_order_, (void *)s_rlog_, (void *)s_dry_, (void *)s_wdry_, (void *)s_win_, (void *)s_arena_, (void *)s_roff_, (void *)s_arena_used_, (void *)s_rentry_, (void *)s_rlists_, (void *)s_rl_, (void *)s_rstate_, (void *)s_rtied_, (void *)s_rfin_ctr_})
        if (p) (void)hipFree(p);
    if (ev0_) (void)hipEventDestroy((hipEvent_t)ev0_);
    if (ev1_) (void)hipEventDestroy((hipEvent_t)ev1_);
    if (ev2_) (void)hipEventDestroy((hipEvent_t)ev2_);
    if (h_stage_) (void)hipHostFree(h_stage_);
    if (h_res_) (void)hipHostFree(h_res_);
    if (h_range_) (void)hipHostFree(h_range_);
    for (LinkSet &ls : lset_) {
        if (ls.h_in) (void)hipHostFree(ls.h_in);
        if (ls.h_out) (void)hipHostFree(ls.h_out);
        if (ls.h_ev) (void)hipHostFree(ls.h_ev);
        for (void *e : {ls.ev_start, ls.ev_stop, ls.ev_done}) if (e) (void)hipEventDestroy((hipEvent_t)e);
    }
    delete hg_;
}

bool Device::reserve(long long capacity)
{
    if (capacity <= capacity_) return true;
    if (!bind()) return false;
    float *nr = nullptr;
    double *nsn = nullptr;
    HIP_OK(hipMalloc(&nr, (size_t)capacity * pitch_ * sizeof(float)));
    if (metric_ == M_COS) HIP_OK(hipMalloc(&nsn, (size_t)capacity * sizeof(double)));
    if (d_rows_) {
        HIP_OK(hipMemcpyAsync(nr, d_rows_, (size_t)capacity_ * pitch_ * sizeof(float), hipMemcpyDeviceToDevice, S(stream_)));
        if (nsn) HIP_OK(hipMemcpyAsync(nsn, d_row_sn_, (size_t)capacity_ * sizeof(double), hipMemcpyDeviceToDevice, S(stream_)));
        HIP_OK(hipStreamSynchronize(S(stream_)));
        HIP_OK(hipFree(d_rows_));
        if (d_row_sn_) HIP_OK(hipFree(d_row_sn_));
    }
    d_rows_ = nr;
    d_row_sn_ = nsn;
    capacity_ = capacity;
    return true;
}

bool Device::upload_rows(int first_id, int n, const float *rows)
{
    if (n <= 0) return true;
    if (first_id < 0 || (long long)first_id + n > capacity_ || !rows) {
        set_dev_error("upload_rows: range outside capacity");
        return false;
    }
    if (!bind()) return false;
    {   // pageable -> pinned bounce buffer -> HBM, 64 MiB at a time (int8: through a float staging area on the
        // device, quantised into records there)
        const size_t row_bytes = (size_t)dim_ * sizeof(float);
        const size_t chunk_rows = std::max<size_t>(1, (64u << 20) / row_bytes);
        char *hs = static_cast<char *>(pinned_stage(std::min<size_t>((size_t)n, chunk_rows) * row_bytes));
        if (!hs) return false;
        if (metric_ == M_I8 && !grow_dev(&q_stage_, &q_stage_cap_, std::min<size_t>((size_t)n, chunk_rows) * (size_t)dim_)) return false;
        for (size_t r0 = 0; r0 < (size_t)n; r0 += chunk_rows) {
            const size_t nr = std::min(chunk_rows, (size_t)n - r0);
            memcpy(hs, rows + r0 * dim_, nr * row_bytes);
            if (metric_ == M_I8) {
                HIP_OK(hipMemcpyAsync(q_stage_, hs, nr * row_bytes, hipMemcpyHostToDevice, S(stream_)));
                hipLaunchKernelGGL(quantize_rows_kernel, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, S(stream_), q_stage_, dim_, (int)nr, d_rows_,
                                   (long long)first_id + (long long)r0, pitch_);
                HIP_OK(hipGetLastError());
            } else {
                HIP_OK(hipMemcpyAsync(d_rows_ + ((size_t)first_id + r0) * pitch_, hs, nr * row_bytes, hipMemcpyHostToDevice, S(stream_)));
            }
            HIP_OK(hipStreamSynchronize(S(stream_))); // the bounce buffer is reused
        }
    }
    if (metric_ == M_COS) {
        int blocks = (int)(((long long)n * 8 + 255) / 256);
        hipLaunchKernelGGL(row_sqrtnorm_kernel, dim3(blocks), dim3(256), 0, S(stream_), d_rows_, pitch_, (long long)first_id, n, d_row_sn_);
        HIP_OK(hipGetLastError());
    }
    HIP_OK(hipStreamSynchronize(S(stream_))); // `rows` is borrowed only for this call
    n_rows_hw_ = std::max(n_rows_hw_, (long long)first_id + n);
    return true;
}

bool Device::upload_rows_begin(int first_id, int n, const float *rows)
{
    if (n <= 0) return true;
    if (metric_ == M_I8 || bg_.active.load()) { set_dev_error("upload_rows_begin: not available (int8 rows, or an upload already in flight)"); return false; }
    if (first_id < 0 || (long long)first_id + n > capacity_ || !rows) { set_dev_error("upload_rows: range outside capacity"); return false; }
    if (!bind()) return false;
    const size_t row_bytes = (size_t)dim_ * sizeof(float);
    const size_t chunk_rows = std::max<size_t>(1, (8u << 20) / row_bytes);
    if (!bg_.stream) {
        hipStream_t st;
        HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        bg_.stream = st;
        for (int i = 0; i < 2; ++i) { hipEvent_t e; HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); bg_.ev[i] = e; }
    }
    if (bg_.pin_bytes < chunk_rows * row_bytes) {
        for (int i = 0; i < 2; ++i) {
            if (bg_.pin[i]) (void)hipHostFree(bg_.pin[i]);
            bg_.pin[i] = nullptr;
            HIP_OK(hipHostMalloc(&bg_.pin[i], chunk_rows * row_bytes, hipHostMallocDefault));
        }
        bg_.pin_bytes = chunk_rows * row_bytes;
    }
    bg_.resident.store(first_id);
    bg_.failed.store(false);
    bg_.active.store(true);
    n_rows_hw_ = std::max(n_rows_hw_, (long long)first_id + n); // ids are valid from now on; their rows are awaited per batch
    bg_.th = std::thread([this, first_id, n, rows, row_bytes, chunk_rows] {
        auto fail = [&](const char *what, hipError_t e) { bg_.err = std::string(what) + ": " + hipGetErrorString(e); bg_.failed.store(true); };
        hipError_t e = hipSetDevice(device_);
        if (e != hipSuccess) { fail("hipSetDevice", e); return; }
        hipStream_t st = S(bg_.stream);
        size_t pending_rows[2] = {0, 0};
        bool in_flight[2] = {false, false};
        int b = 0;
        long long landed = first_id;
        for (size_t r0 = 0; r0 < (size_t)n; r0 += chunk_rows, b ^= 1) {
            const size_t nr = std::min(chunk_rows, (size_t)n - r0);
            if (in_flight[b]) { // this pinned buffer's previous copy must have left it
                if ((e = hipEventSynchronize((hipEvent_t)bg_.ev[b])) != hipSuccess) { fail("hipEventSynchronize", e); return; }
                landed += (long long)pending_rows[b];
                bg_.resident.store(landed, std::memory_order_release);
                in_flight[b] = false;
            }
            memcpy(bg_.pin[b], rows + r0 * dim_, nr * row_bytes);
            if ((e = hipMemcpyAsync(d_rows_ + ((size_t)first_id + r0) * pitch_, bg_.pin[b], nr * row_bytes, hipMemcpyHostToDevice, st)) != hipSuccess) { fail("hipMemcpyAsync", e); return; }
            if (metric_ == M_COS) {
                const int blocks = (int)(((long long)nr * 8 + 255) / 256);
                hipLaunchKernelGGL(row_sqrtnorm_kernel, dim3(blocks), dim3(256), 0, st, d_rows_, pitch_, (long long)first_id + (long long)r0, (int)nr, d_row_sn_);
            }
            if ((e = hipEventRecord((hipEvent_t)bg_.ev[b], st)) != hipSuccess) { fail("hipEventRecord", e); return; }
            pending_rows[b] = nr;
            in_flight[b] = true;
        }
        for (int k = 0; k < 2; ++k, b ^= 1) // the two copies still in flight, oldest first
            if (in_flight[b]) {
                if ((e = hipEventSynchronize((hipEvent_t)bg_.ev[b])) != hipSuccess) { fail("hipEventSynchronize", e); return; }
                landed += (long long)pending_rows[b];
                bg_.resident.store(landed, std::memory_order_release);
            }
    });
    return true;
}

bool Device::upload_rows_wait(long long upto)
{
    if (!bg_.active.load()) return true;
    if (upto >= 0) {
        while (bg_.resident.load(std::memory_order_acquire) < upto && !bg_.failed.load()) std::this_thread::yield();
        if (!bg_.failed.load()) return true;
    }
    if (bg_.th.joinable()) bg_.th.join();
    bg_.active.store(false);
    if (bg_.failed.load()) { set_dev_error("background row upload failed: " + bg_.err); return false; }
    return true;
}

bool Device::download_rows(int first_id, int n, float *rows)
{
    if (n <= 0) return true;
    if (first_id < 0 || (long long)first_id + n > capacity_ || !rows) {
        set_dev_error("download_rows: range outside capacity");
        return false;
    }
    if (!bind()) return false;
    if (metric_ == M_I8) { // the dequantised rows q_i * scale
        if (!grow_dev(&q_stage_, &q_stage_cap_, (size_t)n * (size_t)dim_)) return false;
        hipLaunchKernelGGL(dequantize_rows_kernel, dim3((unsigned)(((long long)n * dim_ + 255) / 256)), dim3(256), 0, S(stream_), d_rows_, pitch_,
                           (long long)first_id, n, dim_, q_stage_);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpyAsync(rows, q_stage_, (size_t)n * dim_ * sizeof(float), hipMemcpyDeviceToHost, S(stream_)));
    } else {
        HIP_OK(hipMemcpyAsync(rows, d_rows_ + (size_t)first_id * pitch_, (size_t)n * dim_ * sizeof(float), hipMemcpyDeviceToHost, S(stream_)));
    }
    HIP_OK(hipStreamSynchronize(S(stream_)));
    return true;
}

// Pageable host memory -> HBM for the large per-call inputs (a 65 536 x 128 query set is 33 MB): four host
// threads copy their quarter of the source through two pinned 2-MB buffers each while the DMA engine drains the
// buffers filled before -- the copy into pinned memory (3 ms on one thread) was most of what the boundary call
// hnsw_knn_query cost beyond the resident-query step.  Everything is enqueued on the context's stream; returns
// when the SOURCE has been read (the caller's buffer is borrowed only for the call), not when the DMA is done.
bool Device::staged_upload(float *dst, const float *src, size_t bytes)
{
    constexpr int T = 4;
    constexpr size_t kChunk = 2u << 20;
    for (int i = 0; i < 2 * T; ++i) {
        if (!up_pin_[i]) HIP_OK(hipHostMalloc(&up_pin_[i], kChunk, hipHostMallocDefault));
        if (!up_ev_[i]) { hipEvent_t e; HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); up_ev_[i] = e; }
    }
    std::atomic<int> failed{0};
    const size_t slice = (((bytes + T - 1) / T) + 255) & ~(size_t)255;
    auto work = [&](int t) {
        if (hipSetDevice(device_) != hipSuccess) { failed.store(1); return; }
        const size_t lo = std::min(bytes, slice * (size_t)t), hi = std::min(bytes, slice * (size_t)(t + 1));
        int b = 0;
        for (size_t off = lo; off < hi; off += kChunk, b ^= 1) {
            const int slot = 2 * t + b;
            const size_t nb = std::min(kChunk, hi - off);
            if (up_busy_[slot] && hipEventSynchronize((hipEvent_t)up_ev_[slot]) != hipSuccess) { failed.store(1); return; }
            memcpy(up_pin_[slot], reinterpret_cast<const char *>(src) + off, nb);
            if (hipMemcpyAsync(reinterpret_cast<char *>(dst) + off, up_pin_[slot], nb, hipMemcpyHostToDevice, S(stream_)) != hipSuccess ||
                hipEventRecord((hipEvent_t)up_ev_[slot], S(stream_)) != hipSuccess) { failed.store(1); return; }
            up_busy_[slot] = true;
        }
    };
    std::thread th[T - 1];
    for (int t = 1; t < T; ++t) th[t - 1] = std::thread(work, t);
    work(0);
    for (int t = 1; t < T; ++t) th[t - 1].join();
    if (failed.load()) { set_dev_error("staged_upload: a HIP call failed"); return false; }
    return true;
}

bool Device::set_queries(const float *queries, int nq)
{
    if (nq < 0 || (nq > 0 && !queries)) { set_dev_error("set_queries: bad argument"); return false; }
    if (!bind()) return false;
    if (nq > q_capacity_) {
        if (d_queries_) HIP_OK(hipFree(d_queries_));
        if (d_q_sn_) HIP_OK(hipFree(d_q_sn_));
        d_queries_ = nullptr; d_q_sn_ = nullptr;
        long long cap = std::max<long long>(nq, 1024);
        HIP_OK(hipMalloc(&d_queries_, (size_t)cap * pitch_ * sizeof(float)));
        if (metric_ == M_COS) HIP_OK(hipMalloc(&d_q_sn_, (size_t)cap * sizeof(double)));
        q_capacity_ = cap;
    }
    n_queries_ = nq;
    if (nq == 0) return true;
    {
        const size_t bytes = (size_t)nq * dim_ * sizeof(float);
        void *hs = bytes < (4u << 20) ? pinned_stage(bytes) : nullptr;
        float *dst = d_queries_;
        if (metric_ == M_I8) { // floats to the staging area, quantised into the resident records
            if (!grow_dev(&q_stage_, &q_stage_cap_, (size_t)nq * (size_t)dim_)) return false;
            dst = q_stage_;
        }
        if (bytes >= (4u << 20)) { if (!staged_upload(dst, queries, bytes)) return false; }
        else if (hs) { memcpy(hs, queries, bytes); HIP_OK(hipMemcpyAsync(dst, hs, bytes, hipMemcpyHostToDevice, S(stream_))); }
        else HIP_OK(hipMemcpyAsync(dst, queries, bytes, hipMemcpyHostToDevice, S(stream_)));
        if (metric_ == M_I8) {
            hipLaunchKernelGGL(quantize_rows_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, S(stream_), q_stage_, dim_, nq, d_queries_, 0LL, pitch_);
            HIP_OK(hipGetLastError());
        }
    }
    if (metric_ == M_COS) {
        int blocks = (int)(((long long)nq * 8 + 255) / 256);
        hipLaunchKernelGGL(row_sqrtnorm_kernel, dim3(blocks), dim3(256), 0, S(stream_), d_queries_, pitch_, 0LL, nq, d_q_sn_);
        HIP_OK(hipGetLastError());
    }
    HIP_OK(hipStreamSynchronize(S(stream_)));
    return true;
}

StepBuffers *Device::alloc_step(int nslots, int stride)
{
    if (nslots <= 0 || stride <= 0) { set_dev_error("alloc_step: bad argument"); return nullptr; }
    if (!bind()) return nullptr;
    StepBuffers *sb = new StepBuffers();
    sb->nslots = nslots;
    sb->stride = stride;
    sb->rec_stride = stride + 2;
    const size_t rec_bytes = sizeof(int) * (size_t)nslots * sb->rec_stride;
    const size_t dist_bytes = sizeof(float) * ((size_t)nslots * stride + StepBuffers::kHeader);
    float *h_dist = nullptr;
    bool ok = hipHostMalloc((void **)&sb->rec, rec_bytes, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void **)&h_dist, dist_bytes, hipHostMallocDefault) == hipSuccess &&
              hipMalloc((void **)&sb->d_rec, rec_bytes) == hipSuccess &&
              hipMalloc((void **)&sb->d_dist, dist_bytes) == hipSuccess;
    if (ok) {
        memset(sb->rec, 0, rec_bytes);
        memset(h_dist, 0, dist_bytes);
        ok = hipMemsetAsync(sb->d_dist, 0, dist_bytes, S(stream_)) == hipSuccess && hipStreamSynchronize(S(stream_)) == hipSuccess;
    }
    if (h_dist) sb->dist = h_dist + StepBuffers::kHeader;
    hipEvent_t ev = nullptr, t0 = nullptr, t1 = nullptr;
    ok = ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess &&
         hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess;
    sb->done = ev; sb->t0 = t0; sb->t1 = t1;
    if (!ok) {
        set_dev_error("alloc_step: allocation failed");
        free_step(sb);
        return nullptr;
    }
    return sb;
}

// hnsw_knn_query hands over host buffers every call, and a 65 536 x 128 query set is 33 MB: uploaded in front of
// the traversal it was 4 % of the call.  Here only the first `head` rows are uploaded before the launch; the rest
// follows on the copy stream WHILE the traversal kernel runs (search_batch -> upload_tail), chunk by chunk, and a word
// in host memory tells the kernel how many rows have landed (graph_search_kernel's `ready`).  Chunks are whole 128-byte
// lines of the query matrix, so no line of it is ever read half-arrived.  Plain float metrics only (nothing to
// compute per query row on arrival).
bool Device::set_queries_streamed(const float *queries, int nq, int head)
{
    if (metric_ == M_COS || metric_ == M_I8 || nq <= 0 || head <= 0 || head >= nq) return set_queries(queries, nq);
    head = std::min(nq, (head + 31) & ~31);
    if (head >= nq) return set_queries(queries, nq);
    if (!bind()) return false;
    if (!h_ready_) {
        HIP_OK(hipHostMalloc((void **)&h_ready_, 64, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_OK(hipHostGetDevicePointer((void **)&d_ready_, h_ready_, 0));
        // The copies a gated kernel waits for must never queue behind that kernel.  HIP multiplexes streams onto a few
        // hardware queues, and a copy stream that lands on the queue of a compute stream does exactly that (measured: two
        // lanes, every wave slept its full bound).  High-priority streams have hardware queues of their own, and only
        // copy streams are created with that priority here.
        int lo = 0, hi = 0;
        HIP_OK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        hipStream_t cs;
        HIP_OK(hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, hi));
        copy_stream_ = cs;
    }
    if (!set_queries(queries, head)) return false; // allocates for `head` rows at least ...
    if (nq > q_capacity_) {                         // ... and for the whole set, keeping the head
        float *nqbuf = nullptr;
        const long long cap = std::max<long long>(nq, 1024);
        HIP_OK(hipMalloc(&nqbuf, (size_t)cap * pitch_ * sizeof(float)));
        HIP_OK(hipMemcpyAsync(nqbuf, d_queries_, (size_t)head * pitch_ * sizeof(float), hipMemcpyDeviceToDevice, S(stream_)));
        HIP_OK(hipStreamSynchronize(S(stream_)));
        HIP_OK(hipFree(d_queries_));
        d_queries_ = nqbuf;
        q_capacity_ = cap;
    }
    n_queries_ = nq;
    __atomic_store_n(h_ready_, head, __ATOMIC_RELEASE);
    tail_.src = queries;
    tail_.first = head;
    tail_.n = nq - head;
    return true;
}

bool Device::upload_tail()
{
    const long long first = tail_.first, n = tail_.n;
    const float *src = tail_.src;
    tail_.n = 0;
    if (n <= 0) return true;
    constexpr size_t kChunk = 2u << 20;
    for (int i = 0; i < 2; ++i) {
        if (!up_pin_[i]) HIP_OK(hipHostMalloc(&up_pin_[i], kChunk, hipHostMallocDefault));
        if (!up_ev_[i]) { hipEvent_t e; HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); up_ev_[i] = e; }
    }
    const size_t row_bytes = (size_t)dim_ * sizeof(float);
    const long long rows_per_chunk = std::max<long long>(32, (long long)(kChunk / row_bytes) & ~31LL); // whole 128-B lines
    hipStream_t cs = S(copy_stream_);
    long long sent = 0, confirmed = 0;
    int b = 0;
    long long in_slot[2] = {0, 0};
    while (confirmed < n) {
        if (sent < n && in_slot[b] == 0) {
            const long long r = std::min(rows_per_chunk, n - sent);
            memcpy(up_pin_[b], src + (size_t)(first + sent) * dim_, (size_t)r * row_bytes);
            HIP_OK(hipMemcpyAsync(d_queries_ + (size_t)(first + sent) * pitch_, up_pin_[b], (size_t)r * row_bytes, hipMemcpyHostToDevice, cs));
            HIP_OK(hipEventRecord((hipEvent_t)up_ev_[b], cs));
            in_slot[b] = r;
            sent += r;
            b ^= 1;
            continue;
        }
        // the older of the two copies in flight
        const int o = in_slot[b] != 0 ? b : b ^ 1;
        HIP_OK(hipEventSynchronize((hipEvent_t)up_ev_[o]));
        confirmed += in_slot[o];
        in_slot[o] = 0;
        b = o;
        __atomic_store_n(h_ready_, (int)(first + confirmed), __ATOMIC_RELEASE);
    }
    up_busy_[0] = up_busy_[1] = false;
    return true;
}

// Peer access between two devices, asked for once per ordered pair.  hipMemcpyPeerAsync works either way -- without peer
// access the runtime stages the copy through host memory -- so a refusal is not an error; it is COUNTED, because "replicas
// are copied device to device over xGMI" is a claim about the machine, not about this code (hnswdev_stats.peer_direct_copies
// / .peer_staged_copies).  Same device on both sides: a plain device-to-device copy, counted as direct.
static bool peer_direct(int dst, int src)
{
    if (dst == src) return true;
    static std::mutex mu;
    static std::map<std::pair<int, int>, bool> known;
    std::lock_guard<std::mutex> lk(mu);
    auto it = known.find({dst, src});
    if (it != known.end()) return it->second;
    int can = 0;
    bool ok = hipDeviceCanAccessPeer(&can, dst, src) == hipSuccess && can != 0;
    if (ok) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        ok = hipSetDevice(dst) == hipSuccess;
        if (ok) {
            const hipError_t e = hipDeviceEnablePeerAccess(src, 0);
            ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            (void)hipGetLastError(); // (already enabled is not an error to keep)
        }
        (void)hipSetDevice(cur);
    }
    known[{dst, src}] = ok;
    return ok;
}

bool Device::clone_from(Device *src, long long pool_len)
{
    if (!src || src == this || src->dim_ != dim_ || src->metric_ != metric_ || src->pitch_ != pitch_) { set_dev_error("clone_from: contexts differ in shape"); return false; }
    if (pool_len < 0 || pool_len > src->g_pool_cap_) { set_dev_error("clone_from: bad pool length"); return false; }
    if (!src->sync()) return false; // what is copied must have landed
    if (!reserve(src->capacity_)) return false;
    const bool direct = peer_direct(device_, src->device_);
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    const long long have = std::min(n_rows_hw_, src->n_rows_hw_); // rows never change once uploaded (slot reuse re-clones: see HnswIndex)
    const long long more = src->n_rows_hw_ - have;
    if (more > 0) {
        HIP_OK(hipMemcpyPeerAsync(d_rows_ + (size_t)have * pitch_, device_, src->d_rows_ + (size_t)have * pitch_, src->device_, (size_t)more * pitch_ * sizeof(float), st));
        if (metric_ == M_COS) HIP_OK(hipMemcpyPeerAsync(d_row_sn_ + have, device_, src->d_row_sn_ + have, src->device_, (size_t)more * sizeof(double), st));
    }
    const long long n = src->g_n_;
    if (n > g_cap_n_ || src->g_stride0_ != g_stride0_) {
        for (void *p : {(void *)g_adj0_, (void *)g_level_, (void *)g_upper_, (void *)g_tested0_}) if (p) HIP_OK(hipFree(p));
        g_adj0_ = nullptr; g_level_ = nullptr; g_upper_ = nullptr; g_tested0_ = nullptr;
        const long long cap = std::max<long long>(src->g_cap_n_, std::max<long long>(n, 1024));
        HIP_OK(hipMalloc(&g_adj0_, sizeof(int) * (size_t)cap * src->g_stride0_));
        HIP_OK(hipMalloc(&g_level_, sizeof(int) * (size_t)cap));
        HIP_OK(hipMalloc(&g_upper_, sizeof(int64_t) * (size_t)cap));
        HIP_OK(hipMalloc(&g_tested0_, sizeof(int) * (size_t)cap));
        g_cap_n_ = cap;
    }
    if (pool_len > g_pool_cap_) {
        if (g_pool_) HIP_OK(hipFree(g_pool_));
        if (g_testedU_) HIP_OK(hipFree(g_testedU_));
        g_pool_ = nullptr; g_testedU_ = nullptr;
        const long long cap = std::max<long long>(pool_len * 2, 4096);
        HIP_OK(hipMalloc(&g_pool_, sizeof(int) * (size_t)cap));
        HIP_OK(hipMalloc(&g_testedU_, sizeof(int) * (size_t)cap));
        g_pool_cap_ = cap;
    }
    g_n_ = n; g_stride0_ = src->g_stride0_; g_strideU_ = src->g_strideU_;
    if (n > 0) {
        HIP_OK(hipMemcpyPeerAsync(g_adj0_, device_, src->g_adj0_, src->device_, sizeof(int) * (size_t)n * g_stride0_, st));
        HIP_OK(hipMemcpyPeerAsync(g_level_, device_, src->g_level_, src->device_, sizeof(int) * (size_t)n, st));
        HIP_OK(hipMemcpyPeerAsync(g_upper_, device_, src->g_upper_, src->device_, sizeof(int64_t) * (size_t)n, st));
    }
    if (pool_len > 0) HIP_OK(hipMemcpyPeerAsync(g_pool_, device_, src->g_pool_, src->device_, sizeof(int) * (size_t)pool_len, st));
    // a replica only answers queries: the link kernel's pruning history is not carried over
    if (g_tested0_) HIP_OK(hipMemsetAsync(g_tested0_, 0, sizeof(int) * (size_t)g_cap_n_, st));
    if (g_testedU_) HIP_OK(hipMemsetAsync(g_testedU_, 0, sizeof(int) * (size_t)g_pool_cap_, st));
    HIP_OK(hipStreamSynchronize(st));
    n_rows_hw_ = src->n_rows_hw_;
    stats_.replica_bytes += (uint64_t)more * pitch_ * sizeof(float) + sizeof(int) * ((uint64_t)n * g_stride0_ + (uint64_t)n * 3 + (uint64_t)pool_len);
    (direct ? stats_.peer_direct_copies : stats_.peer_staged_copies) += 1;
    return true;
}

bool Device::adopt_queries(Device *src, long long first, long long n, long long at, long long total)
{
    if (!src || src->dim_ != dim_ || src->metric_ != metric_ || first < 0 || n < 0 || first + n > src->n_queries_ || at < 0 || at + n > total) {
        set_dev_error("adopt_queries: bad argument");
        return false;
    }
    if (!src->sync()) return false;
    (peer_direct(device_, src->device_) ? stats_.peer_direct_copies : stats_.peer_staged_copies) += 1;
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    if (total > q_capacity_) { // grow, keeping what is resident
        float *nq = nullptr;
        double *nsn = nullptr;
        const long long cap = std::max<long long>(total, 1024);
        HIP_OK(hipMalloc(&nq, (size_t)cap * pitch_ * sizeof(float)));
        if (metric_ == M_COS) HIP_OK(hipMalloc(&nsn, (size_t)cap * sizeof(double)));
        if (d_queries_ && n_queries_ > 0) {
            HIP_OK(hipMemcpyAsync(nq, d_queries_, (size_t)n_queries_ * pitch_ * sizeof(float), hipMemcpyDeviceToDevice, st));
            if (nsn) HIP_OK(hipMemcpyAsync(nsn, d_q_sn_, (size_t)n_queries_ * sizeof(double), hipMemcpyDeviceToDevice, st));
            HIP_OK(hipStreamSynchronize(st));
        }
        if (d_queries_) HIP_OK(hipFree(d_queries_));
        if (d_q_sn_) HIP_OK(hipFree(d_q_sn_));
        d_queries_ = nq; d_q_sn_ = nsn; q_capacity_ = cap;
    }
    if (n > 0 && src != this) {
        HIP_OK(hipMemcpyPeerAsync(d_queries_ + (size_t)at * pitch_, device_, src->d_queries_ + (size_t)first * pitch_, src->device_, (size_t)n * pitch_ * sizeof(float), st));
        if (metric_ == M_COS) HIP_OK(hipMemcpyPeerAsync(d_q_sn_ + at, device_, src->d_q_sn_ + first, src->device_, (size_t)n * sizeof(double), st));
        HIP_OK(hipStreamSynchronize(st));
    }
    n_queries_ = total;
    return true;
}

void Device::free_step(StepBuffers *sb)
{
    if (!sb) return;
    (void)hipSetDevice(device_);
    if (sb->rec) (void)hipHostFree(sb->rec);
    if (sb->dist) (void)hipHostFree(sb->dist - StepBuffers::kHeader);
    if (sb->d_rec) (void)hipFree(sb->d_rec);
    if (sb->d_dist) (void)hipFree(sb->d_dist);
    if (sb->done) (void)hipEventDestroy((hipEvent_t)sb->done);
    if (sb->t0) (void)hipEventDestroy((hipEvent_t)sb->t0);
    if (sb->t1) (void)hipEventDestroy((hipEvent_t)sb->t1);
    delete sb;
}

bool Device::launch_step(StepBuffers *sb, int nslots_used, uint64_t evals)
{
    sb->in_flight = false;
    if (nslots_used <= 0 || evals == 0) { sb->evals = 0; sb->timed = false; return true; }
    if (nslots_used > sb->nslots) { set_dev_error("launch_step: too many slots"); return false; }
    hipStream_t st = S(stream_);
    sb->timed = profiling_;
    sb->evals = evals;
    HIP_OK(hipMemcpyAsync(sb->d_rec, sb->rec, sizeof(int) * (size_t)nslots_used * sb->rec_stride, hipMemcpyHostToDevice, st));
    if (sb->timed) HIP_OK(hipEventRecord((hipEvent_t)sb->t0, st));
    dim3 grid((nslots_used + 3) / 4), block(256);
#define LAUNCH(M)                                                                                          \
    hipLaunchKernelGGL(slot_distance_kernel<M>, grid, block, 0, st, d_rows_, d_row_sn_, d_queries_, d_q_sn_, pitch_, \
                       sb->d_rec, sb->d_dist + StepBuffers::kHeader, sb->stride, sb->rec_stride, nslots_used,  \
                       n_rows_hw_, n_queries_, reinterpret_cast<int *>(sb->d_dist))
    if (metric_ == M_SQ) LAUNCH(M_SQ);
    else if (metric_ == M_COS) LAUNCH(M_COS);
    else if (metric_ == M_I8) LAUNCH(M_I8);
    else LAUNCH(M_UCOS);
#undef LAUNCH
    HIP_OK(hipGetLastError());
    if (sb->timed) HIP_OK(hipEventRecord((hipEvent_t)sb->t1, st));
    // guard word + distances of the used slots: one copy
    HIP_OK(hipMemcpyAsync(sb->dist - StepBuffers::kHeader, sb->d_dist, sizeof(float) * ((size_t)nslots_used * sb->stride + StepBuffers::kHeader),
                          hipMemcpyDeviceToHost, st));
    HIP_OK(hipEventRecord((hipEvent_t)sb->done, st));
    sb->in_flight = true;
    stats_.launches++;
    stats_.evals += evals;
    return true;
}

bool Device::wait_step(StepBuffers *sb)
{
    if (!sb->in_flight) return true;
    HIP_OK(hipEventSynchronize((hipEvent_t)sb->done));
    sb->in_flight = false;
    if (sb->timed) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)sb->t0, (hipEvent_t)sb->t1));
        stats_.kernel_ms += ms;
        stats_.timed_launches++;
        stats_.timed_evals += sb->evals;
        sb->timed = false;
    }
    sb->evals = 0;
    int *guard = reinterpret_cast<int *>(sb->dist - StepBuffers::kHeader);
    if (*guard != 0) { // the kernel refused a record: clear the flag, report
        *guard = 0;
        HIP_OK(hipMemsetAsync(sb->d_dist, 0, sizeof(int), S(stream_)));
        set_dev_error("distance step: a record names a row / query outside the uploaded data, or more ids than a slot holds (those distances are NaN)");
        return false;
    }
    return true;
}

// ---- step buffers of the C ABI ---------------------------------------------------------------
bool Device::step_buffers(int set, int nslots, int stride, int **rec, float **dist, bool internal)
{
    if (set < 0 || set > (internal ? 3 : 1) || nslots <= 0 || stride <= 0 || !rec || !dist) { set_dev_error("step_buffers: bad argument"); return false; }
    StepBuffers *&sb = abi_sb_[set];
    if (sb && sb->in_flight) { set_dev_error("step_buffers: the set is in flight (call hnswdev_step_wait first)"); return false; }
    if (!sb || sb->nslots < nslots || sb->stride != stride) {
        if (sb) { if (!sync()) return false; free_step(sb); sb = nullptr; }
        sb = alloc_step(nslots, stride);
        if (!sb) return false;
    }
    *rec = sb->rec;
    *dist = sb->dist;
    return true;
}

bool Device::step_submit(int set, int nslots_used)
{
    if (set < 0 || set > 1 || !abi_sb_[set]) { set_dev_error("step_submit: no such buffer set (call hnswdev_step_buffers first)"); return false; }
    StepBuffers *sb = abi_sb_[set];
    if (sb->in_flight) { set_dev_error("step_submit: the set is already in flight"); return false; }
    if (nslots_used < 0 || nslots_used > sb->nslots) { set_dev_error("step_submit: more slots than the set holds"); return false; }
    uint64_t evals = 0; // ids are guarded on the device; the counts are summed here for the counters
    for (int s = 0; s < nslots_used; ++s) {
        const int c = sb->rec[(size_t)s * sb->rec_stride];
        if (c > 0) evals += (uint64_t)c;
    }
    if (!bind()) return false;
    return launch_step(sb, nslots_used, evals);
}

bool Device::step_wait(int set)
{
    if (set < 0 || set > 1 || !abi_sb_[set]) { set_dev_error("step_wait: no such buffer set"); return false; }
    if (!bind()) return false;
    return wait_step(abi_sb_[set]);
}

bool Device::sync()
{
    if (!bind()) return false;
    HIP_OK(hipStreamSynchronize(S(stream_)));
    return true;
}

void Device::get_stats(hnswdev_stats *out) { *out = stats_; }
void Device::reset_stats()
{
    uint64_t rb = stats_.row_bytes;
    stats_ = hnswdev_stats{};
    stats_.row_bytes = rb;
#ifdef EXP_PHASE_CLOCKS
    (void)hipDeviceSynchronize();
    phase_report("reset_stats");
    phase_zero();
#endif
}


// ---- graph mirror + graph-resident search -------------------------------------------------
bool Device::set_graph(const int *adj0, long long n, int stride0, const int *level, const int64_t *upper, const int *pool,
                       long long pool_len, int strideU)
{
    if (n < 0 || (n > 0 && (!adj0 || !level || !upper))) { set_dev_error("set_graph: bad argument"); return false; }
    if (stride0 - 1 > 128 || strideU - 1 > 128) { set_dev_error("set_graph: MaxEdges > 63 is not supported by the graph-resident kernels (use hnsw_mi355x_set_device_traversal(0))"); return false; }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    if (n > g_cap_n_ || stride0 != g_stride0_) {
        if (g_adj0_) HIP_OK(hipFree(g_adj0_));
        if (g_level_) HIP_OK(hipFree(g_level_));
        if (g_upper_) HIP_OK(hipFree(g_upper_));
        if (g_tested0_) HIP_OK(hipFree(g_tested0_));
        g_adj0_ = nullptr; g_level_ = nullptr; g_upper_ = nullptr; g_tested0_ = nullptr;
        long long cap = std::max<long long>(n, std::max<long long>(capacity_, 1024));
        HIP_OK(hipMalloc(&g_adj0_, sizeof(int) * (size_t)cap * stride0));
        HIP_OK(hipMalloc(&g_level_, sizeof(int) * (size_t)cap));
        HIP_OK(hipMalloc(&g_upper_, sizeof(int64_t) * (size_t)cap));
        HIP_OK(hipMalloc(&g_tested0_, sizeof(int) * (size_t)cap));
        g_cap_n_ = cap;
    }
    if (pool_len > g_pool_cap_) {
        if (g_pool_) HIP_OK(hipFree(g_pool_));
        if (g_testedU_) HIP_OK(hipFree(g_testedU_));
        g_pool_ = nullptr; g_testedU_ = nullptr;
        long long cap = std::max<long long>(pool_len * 2, 4096);
        HIP_OK(hipMalloc(&g_pool_, sizeof(int) * (size_t)cap));
        HIP_OK(hipMalloc(&g_testedU_, sizeof(int) * (size_t)cap)); // indexed by list offset / strideU: never more than cap
        g_pool_cap_ = cap;
    }
    g_n_ = n; g_stride0_ = stride0; g_strideU_ = strideU;
    if (n > 0) {
        HIP_OK(hipMemcpyAsync(g_adj0_, adj0, sizeof(int) * (size_t)n * stride0, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(g_level_, level, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(g_upper_, upper, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
    }
    if (pool_len > 0) HIP_OK(hipMemcpyAsync(g_pool_, pool, sizeof(int) * (size_t)pool_len, hipMemcpyHostToDevice, st));
    // lists that arrive from the host carry no pruning history
    if (g_tested0_) HIP_OK(hipMemsetAsync(g_tested0_, 0, sizeof(int) * (size_t)g_cap_n_, st));
    if (g_testedU_) HIP_OK(hipMemsetAsync(g_testedU_, 0, sizeof(int) * (size_t)g_pool_cap_, st));
    HIP_OK(hipStreamSynchronize(st)); // host arrays are borrowed only for this call
    return true;
}

// LDS part of the candidate heap: sized for the common case (4 x beam width; C2 queries peak
// near 500 entries at ef = 128), the rest spills to HBM (SpillHeap).  A smaller LDS footprint means
// more resident waves to hide memory latency: 7.6 -> 6.1 ms per 10k-query launch going from 1024
// to 512 entries.  Beyond LDS + spill capacity the traversal is flagged for the lock-step path.
// Register sets of the sorted-list traversal (SortedTop<NS>: k <= 64 * NS); 0 = two-heap traversal
// only.  HNSW_MI355X_SORTED_TOP=0 forces the latter (the tests run both).
static int sorted_top_sets(int k)
{
    if (diag("sorted_top", 1) == 0) return 0;
    return k <= 128 ? 2 : k <= 256 ? 4 : k <= 512 ? 8 : 0; // (a beam of up to 64 entries runs in the two-set form as well)
}
constexpr long long kSortedTopMaxNodes = 1LL << 30; // the sorted list keeps two mark bits in the id word

int Device::max_waves_per_cu()
{
    // 20 = five per SIMD: what the int8 kernels' 92 VGPRs allow (10M x 96 int8, 12 500-query calls: 2.19 M queries/s at
    // 16, 2.31 M at 20); the float kernels (168 VGPRs) keep 12 resident whatever this says
    return 20;
}

// Blocks (= waves) of a persistent traversal launch: what stays resident on the chip.
template <class K>
static int resident_blocks(K kernel, size_t lds, int num_cu, int threads = 64)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || per_cu < 1) per_cu = threads > 64 ? 4 : 8;
    return per_cu * std::max(1, num_cu);
}

static int cand_lds_cap(int k, int dim, bool heur, int nbcap)
{
    int cap = std::min(std::max(4 * k, 256), 4096);
    if (const int c = diag("cand_cap", 0)) cap = std::max(1, c); // tests: force spill / hand-back
    while (cap > 64 && search_lds_bytes(k, cap, dim, heur, nbcap) > 64 * 1024) cap /= 2;
    return cap;
}
static int spill_cap_for_tests()
{
    const int c = diag("spill_cap", -1);
    return c < 0 ? kSpillCap : std::min(kSpillCap, c);
}


// Row loads overlapped with the visited atomics in launches that do not fill the chip, and in every launch on a
// graph large enough for the visited hash tables: there the traversal is bound by rows in flight, not by bytes
// (10M x 96 int8: 2.19 -> 2.43 M queries/s, 1.03 -> 1.12 M adds/s; 10M x 128 f32: +3 % / +5 %), while a full
// launch at 1M nodes is bandwidth-bound and gains nothing.  HNSW_MI355X_OVERLAP=0 disables, =2 forces it for
// every launch (tests).
// Shadow traversals in the search launches (graph_search_kernel): idle waves of a draining launch start the exact
// traversal of the jobs still running.  HNSW_MI355X_SHADOW=0 disables (tests run both).
static bool shadow_mode()
{
    return diag("shadow", 1) != 0;
}
static int overlap_mode()
{
    return diag("overlap", 1);
}
// The latency variants of the traversal kernels (device_kernels.h, LAT) for launches that do not fill the chip -- B = 1
// Add, the exact window's rounds, small query calls: 0 never, 1 (default) when the jobs fit the variant's resident waves,
// 2 whenever the graph allows it (adjacency lists of at most 64 entries; tests).
// KnnQuery launches run WITHOUT a visited set (traverse_sorted, oflags bit 3): every listed neighbour's row is requested as
// soon as the list is known, and a neighbour seen before is recognised by what the set was standing in for -- it is still in
// the result list, or the push test turns it away again.  HNSW_MI355X_NOVIS=0 keeps the sets, =1 drops them only on the
// graphs whose sets are hash tables (A/B runs; same answers either way).  Measured, same box: C2 (1M x 128, bitsets) 2.46-2.56
// -> 3.07 M queries/s (25-26 -> 20.7 ms per 65 536-query launch, +2.5 % rows measured), 12 500-query calls 2.0 -> 2.39 M; C4-size
// 1.68 -> 2.03 M, C5-size 2.26 -> 2.70 M.
static bool lean_mode() { return diag("lean", 1) != 0; } // launches with flags 9 on the lean kernel forms (0: the plain forms read the flags)
static int novis_mode() // 0 never, 1 hash-table graphs only, 2 (default) every graph
{
    return diag("novis", 2);
}
static constexpr size_t kTeamLds = ((sizeof(TeamMail) + 15) & ~(size_t)15) + 16; // the latency variants' mailbox, behind the traversal's LDS
static int lat_mode()
{
    return diag("lat", 1);
}

// The MFMA Gram-block prefilter of RelativeNeighborPruning (device_kernels.h): on by default where it applies
// (cosine family, dim % 8 == 0); HNSW_MI355X_MFMA=0 keeps the exact-only forms (the tests run both).
static bool mfma_heuristic()
{
    return diag("mfma", 1) != 0;
}

// The per-wave visited-id hash tables (VisitedSet): capacity a power of two, >= 16384 and >= 64 per
// beam entry (a traversal visits roughly 35 ids per beam entry), all entries -1 between jobs.
// HNSW_MI355X_VIS_HASH=1/0 forces / forbids them; HNSW_MI355X_VIS_HASH_CAP overrides the capacity (tests).
bool Device::visited_table(size_t vis_bytes_per_job, int k, int **out, int *out_cap, int min_cap, bool allow_hash)
{
    *out = nullptr;
    *out_cap = 0;
    if (!allow_hash) return true; // (the eight-set kernels have no hash-table form: their launches keep bitsets whatever the graph's size)
    const int e = diag("vis_hash", -1);
    // measured: at 1M nodes (125-KB bitsets) the bitset is faster (2.5 M vs 1.9 M queries/s on C2); at
    // 10M (1.25 MB) the table wins (1.48 M vs 1.28 M with the log-cleared bitset, 0.98 M streaming it)
    const bool want = e >= 0 ? e != 0 : vis_bytes_per_job > (512u << 10);
    if (!want) return true;
    int cap = 16384;
    while (cap < 64 * k && cap < (1 << 22)) cap <<= 1;
    if (const int c = diag("vis_hash_cap", 0)) { cap = 64; while (cap < c && cap < (1 << 22)) cap <<= 1; }
    // a traversal step inserts up to min_cap / 4 ids between two looks at crowded() (limit: 3/4 of the table)
    while (cap < min_cap) cap <<= 1;
    const size_t need = (size_t)max_slots() * (size_t)cap;
    if (need > s_vistab_cap_ || cap != s_vistab_each_) {
        HIP_OK(hipStreamSynchronize(S(stream_)));
        if (s_vistab_) HIP_OK(hipFree(s_vistab_));
        s_vistab_ = nullptr; s_vistab_cap_ = 0;
        HIP_OK(hipMalloc(&s_vistab_, sizeof(int) * need));
        HIP_OK(hipMemsetAsync(s_vistab_, 0xff, sizeof(int) * need, S(stream_)));
        s_vistab_cap_ = need;
        s_vistab_each_ = cap;
    }
    *out = s_vistab_;
    *out_cap = cap;
    return true;
}

// chunk: jobs per launch (job / result buffers); slots: waves of a persistent launch (visited
// bitsets, spill areas).  The visited arena is all zero between launches: zeroed when allocated,
// and every wave clears its bitset after each job.
bool Device::ensure_search_scratch(long long chunk, long long slots, int k, size_t vis_bytes_per_job)
{
    if (vis_bytes_per_job * (size_t)slots > s_visited_bytes_) {
        if (s_visited_) HIP_OK(hipFree(s_visited_));
        s_visited_ = nullptr;
        s_visited_bytes_ = vis_bytes_per_job * (size_t)slots;
        HIP_OK(hipMalloc(&s_visited_, s_visited_bytes_));
        HIP_OK(hipMemsetAsync(s_visited_, 0, s_visited_bytes_, S(stream_)));
    }
    if (!s_jobctr_ || (size_t)chunk > s_jobs_cap_) { // [next job, next shadow, -, -, one word per job] (graph_search_kernel)
        if (s_jobctr_) HIP_OK(hipFree(s_jobctr_));
        s_jobctr_ = nullptr;
        HIP_OK(hipMalloc(&s_jobctr_, sizeof(int) * (4 + std::max<size_t>((size_t)chunk, s_jobs_cap_))));
    }
    if ((size_t)chunk > s_jobs_cap_) {
        if (s_jobs_) HIP_OK(hipFree(s_jobs_));
        uj_len_ = 0;
        if (s_cnt_) HIP_OK(hipFree(s_cnt_));
        if (s_flag_) HIP_OK(hipFree(s_flag_));
        s_jobs_cap_ = (size_t)chunk;
        HIP_OK(hipMalloc(&s_jobs_, sizeof(SearchJob) * s_jobs_cap_));
        HIP_OK(hipMalloc(&s_cnt_, sizeof(int) * s_jobs_cap_));
        HIP_OK(hipMalloc(&s_flag_, sizeof(int) * s_jobs_cap_));
    }
    if (k > 0 && !grow_dev(&s_hits_, &s_hits_cap_, (size_t)chunk * k + ((size_t)chunk + 1) / 2)) return false; // ids, distances, and (single-launch calls) the flags behind them
    if (!grow_dev(&s_spill_, &s_spill_cap_, (size_t)slots * kSpillCap + 8)) return false; // +8: get2 may read one entry past a heap
    if (!s_evals_) HIP_OK(hipMalloc(&s_evals_, sizeof(unsigned long long)));
    if (!ev0_) { hipEvent_t a, b; HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b)); ev0_ = a; ev1_ = b; }
    if (!ev2_) { hipEvent_t c; HIP_OK(hipEventCreateWithFlags(&c, hipEventDisableTiming)); ev2_ = c; }
    return true;
}

static bool jobs_valid(const SearchJob *jobs, int njobs, long long g_n, long long n_queries, long long n_rows)
{
    for (int i = 0; i < njobs; ++i) {
        const SearchJob &j = jobs[i];
        bool ok = j.entry >= 0 && j.entry < g_n && j.search_layer >= 0 && j.entry_layer >= j.search_layer &&
                  (j.qref >= 0 ? j.qref < n_queries : (~j.qref) < n_rows);
        if (!ok) return false;
    }
    return true;
}

bool Device::insert_search_batch(const SearchJob *jobs, int njobs, int k, int max_edges0, int n_upper, InsertResults *res, WindowExtras *win)
{
    if (njobs <= 0) return true;
    const bool windowed = win != nullptr; // exact-window Add: read logs, selections and dry-run flags come back with the job flags, one wait
    const int read_log_cap = windowed ? win->read_log_cap : 0;
    if (windowed && (read_log_cap < 8 || njobs > (1 << 16) || (n_upper > 0 && !win->upper_owner))) { set_dev_error("insert_search_batch: bad window request"); return false; }
    if (!jobs || !res || k < 1 || n_upper < 0 || max_edges0 < 2) { set_dev_error("insert_search_batch: bad argument"); return false; }
    if (g_n_ <= 0) { set_dev_error("insert_search_batch: no graph uploaded"); return false; }
    for (int i = 0; i < njobs; ++i) {
        const SearchJob &j = jobs[i];
        if (j.qref >= 0) { set_dev_error("insert_search_batch: qref must name a stored row"); return false; }
        if (j.search_layer > 0 && (j.aux < 0 || j.aux + j.search_layer > n_upper)) { set_dev_error("insert_search_batch: upper-layer slot out of range"); return false; }
        if (j.stop_layer < 0 || j.stop_layer > j.search_layer || (!windowed && j.stop_layer != 0)) { set_dev_error("insert_search_batch: bad stop layer"); return false; }
    }
    if (!jobs_valid(jobs, njobs, g_n_, n_queries_, n_rows_hw_)) { set_dev_error("insert_search_batch: job outside the uploaded graph / rows"); return false; }
    const int cand_cap = cand_lds_cap(k, pitch_, true, nbcap());
    const size_t lds = search_lds_bytes(k, cand_cap, pitch_, true, nbcap());
    if (lds > 64 * 1024) { set_dev_error("insert_search_batch: beam width / dimension exceed the LDS budget"); return false; }
    const int ns_req = g_n_ < kSortedTopMaxNodes ? sorted_top_sets(k) : 0;
    const bool exact_only = ns_req == 0;           // beams beyond 512 entries / HNSW_MI355X_SORTED_TOP=0: the two-set form with launch flag 0x200
    const int ns = exact_only ? 2 : ns_req;
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    const int sel_stride = max_edges0;
    long long vis_words = ((g_n_ + 31) / 32 + 3) & ~3LL;
    size_t vis_bytes_per_job = sizeof(unsigned) * (size_t)vis_words;
    const long long chunk = std::min<long long>(njobs, 1 << 20);
    int *vis_tab = nullptr;
    int vis_tab_cap = 0;
    if (!visited_table(vis_bytes_per_job, k, &vis_tab, &vis_tab_cap, 512, ns != 8)) return false;
    if (vis_tab) { vis_words = 0; vis_bytes_per_job = 16; } // the HASHED kernels never touch the bitset arena: do not allocate one
    if (!ensure_search_scratch(chunk, max_slots(), 0, vis_bytes_per_job)) return false;
    const size_t nU = (size_t)std::max(n_upper, 1);
    if (!grow_dev(&s_sel_, &s_sel_cap_, (size_t)njobs * sel_stride) || !grow_dev(&s_lcnt_, &s_lcnt_cap_, (size_t)njobs) ||
        !grow_dev(&s_selU_, &s_selU_cap_, nU * sel_stride) || !grow_dev(&s_cntU_, &s_cntU_cap_, nU) ||
        !grow_dev(&s_iflag_, &s_iflag_cap_, (size_t)njobs))
        return false;
    if (windowed && !grow_dev(&s_rlog_, &s_rlog_cap_, (size_t)njobs * (size_t)read_log_cap)) return false;
    // pinned results: [sel0 | cnt0 | selU | cntU | flag | evals | read logs | dry0 | dryU]
    const size_t b_sel0 = 4u * (size_t)njobs * sel_stride, b_cnt0 = 4u * (size_t)njobs, b_selU = 4u * nU * sel_stride, b_cntU = 4u * nU, b_flag = 4u * (size_t)njobs;
    const size_t b_log = windowed ? 4u * (size_t)njobs * (size_t)read_log_cap : 0;
    const size_t b_dry = windowed ? b_sel0 + b_selU : 0;
    const size_t b_drop = windowed ? 3u * b_sel0 : 0;          // three ids per layer-0 selection entry
    const size_t b_rep = windowed ? b_flag : 0;                // the jobs' "repeated" flags (the job flags themselves are folded to 0 / 1 below)
    if (windowed && !grow_dev(&s_wdry_, &s_wdry_cap_, nU)) return false; // (upper_owner)
    const size_t need = b_sel0 + b_cnt0 + b_selU + b_cntU + b_flag + 16 + b_log + b_dry + b_drop + b_rep;
    // windowed: everything the kernels write lives in ONE device block laid out like the pinned results below, so that it comes
    // back in one copy (a round of the exact window is ~2 ms: a dozen 4-microsecond copies and their launch overhead showed)
    if (windowed && !grow_dev(&s_win_, &s_win_cap_, (need - b_rep + 3) / 4u + 64)) return false;
    if (need > h_res_cap_) {
        if (h_res_) (void)hipHostFree(h_res_);
        h_res_ = nullptr; h_res_cap_ = 0;
        if (hipHostMalloc(&h_res_, need + need / 2, hipHostMallocDefault) != hipSuccess) { set_dev_error("insert_search_batch: pinned allocation failed"); return false; }
        h_res_cap_ = need + need / 2;
    }
    char *hb = static_cast<char *>(h_res_);
    int *h_sel0 = reinterpret_cast<int *>(hb), *h_cnt0 = reinterpret_cast<int *>(hb + b_sel0);
    int *h_selU = reinterpret_cast<int *>(hb + b_sel0 + b_cnt0), *h_cntU = reinterpret_cast<int *>(hb + b_sel0 + b_cnt0 + b_selU);
    int *h_flag = reinterpret_cast<int *>(hb + b_sel0 + b_cnt0 + b_selU + b_cntU);
    unsigned long long *h_ev = reinterpret_cast<unsigned long long *>(hb + ((b_sel0 + b_cnt0 + b_selU + b_cntU + b_flag + 7) & ~(size_t)7));
    int *h_log = reinterpret_cast<int *>(reinterpret_cast<char *>(h_ev) + 8);
    // where the kernels write: the members, or (windowed) the same offsets inside s_win_
    char *db = reinterpret_cast<char *>(s_win_);
    int *p_sel0 = windowed ? reinterpret_cast<int *>(db) : s_sel_, *p_cnt0 = windowed ? reinterpret_cast<int *>(db + b_sel0) : s_lcnt_;
    int *p_selU = windowed ? reinterpret_cast<int *>(db + b_sel0 + b_cnt0) : s_selU_, *p_cntU = windowed ? reinterpret_cast<int *>(db + b_sel0 + b_cnt0 + b_selU) : s_cntU_;
    int *p_flag = windowed ? reinterpret_cast<int *>(db + b_sel0 + b_cnt0 + b_selU + b_cntU) : s_iflag_;
    const size_t off_ev = (b_sel0 + b_cnt0 + b_selU + b_cntU + b_flag + 7) & ~(size_t)7;
    unsigned long long *p_evals = windowed ? reinterpret_cast<unsigned long long *>(db + off_ev) : s_evals_;
    int *p_log = windowed ? reinterpret_cast<int *>(db + off_ev + 8) : s_rlog_;
    // staging: [jobs | processing order]
    SearchJob *h_jobs = static_cast<SearchJob *>(pinned_stage((sizeof(SearchJob) + sizeof(int)) * (size_t)chunk));
    if (!h_jobs) return false;
    int *h_order = reinterpret_cast<int *>(h_jobs + chunk);
    if (!grow_dev(&s_order_, &s_order_cap_, (size_t)chunk)) return false;
    for (long long off = 0; off < njobs; off += chunk) {
        const int nj = (int)std::min<long long>(chunk, njobs - off);
        memcpy(h_jobs, jobs + off, sizeof(SearchJob) * (size_t)nj);
        HIP_OK(hipMemcpyAsync(s_jobs_, h_jobs, sizeof(SearchJob) * (size_t)nj, hipMemcpyHostToDevice, st));
        uj_len_ = 0; // search_queries' jobs are gone from s_jobs_
        // items that search several layers first (counting sort by first layer, descending; stable)
        const int *d_order = nullptr;
        {
            int hist[64] = {0}, maxl = 0;
            for (int i = 0; i < nj; ++i) { const int l = std::min(h_jobs[i].search_layer, 63); hist[l]++; maxl = std::max(maxl, l); }
            if (maxl > 0 && nj > 1) {
                int start[64], acc = 0;
                for (int l = maxl; l >= 0; --l) { start[l] = acc; acc += hist[l]; }
                for (int i = 0; i < nj; ++i) h_order[start[std::min(h_jobs[i].search_layer, 63)]++] = i;
                HIP_OK(hipMemcpyAsync(s_order_, h_order, sizeof(int) * (size_t)nj, hipMemcpyHostToDevice, st));
                d_order = s_order_;
            }
        }
        // (Add's searches without a visited set as well -- see novis_mode(); HNSW_MI355X_NOVIS_INSERT=0 keeps the sets there)
        const bool novis_ins_on = diag("novis_insert", 1) != 0; // (read per call, like every other switch)
        const bool novis_ins_ = novis_ins_on && g_stride0_ - 2 <= 64 && overlap_mode() != 0 && (size_t)pitch_ * sizeof(float) <= 1024 && (novis_mode() == 2 || (novis_mode() == 1 && vis_tab != nullptr));
        HIP_OK(hipMemsetAsync(s_jobctr_, 0, sizeof(int), st));
        HIP_OK(hipMemsetAsync(p_evals, 0, sizeof(unsigned long long), st));
        const bool timed = profiling_;
        if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev0_, st));
#define LAUNCH2L(M, NS_, H_, LAT_, SLOTS, GRID, LDS, CAP) \
        hipLaunchKernelGGL((graph_insert_search_kernel<M, NS_, H_, LAT_>), dim3(std::min<int>(GRID, SLOTS)), \
                       dim3(LAT_ == kFormLat ? 128 : 64), (LDS) + (LAT_ == kFormLat ? kTeamLds : 0), st, d_rows_, d_row_sn_, pitch_, g_adj0_, g_stride0_, \
                       g_upper_, g_pool_, g_strideU_, s_jobs_, k, CAP, reinterpret_cast<ND *>(s_spill_), spill_cap_for_tests(),  \
                       max_edges0, s_visited_, vis_words, vis_tab, vis_tab_cap, p_sel0 + (size_t)off * sel_stride, p_cnt0 + off, p_selU, p_cntU,        \
                       sel_stride, p_flag + off, p_evals, nbcap(), GRID, s_jobctr_, (exact_only ? 0x200 : 0) | (novis_ins_ ? 9 : (overlap_mode() == 2 || (overlap_mode() == 1 && (GRID <= SLOTS || vis_tab != nullptr))) ? 1 : 0) | (mfma_heuristic() ? 2 : 0), d_order, \
                       windowed ? p_log : (int *)nullptr, read_log_cap)
#define LAUNCH2(M, NS_, H_, GRID, LDS, CAP) \
    do { \
        const int lslots_ = !exact_only && lat_mode() != 0 && g_stride0_ - 2 <= 64 && (LDS) + kTeamLds <= 64 * 1024 ? std::min(max_slots(), resident_blocks(graph_insert_search_kernel<M, NS_, H_, kFormLat>, (LDS) + kTeamLds, num_cu_, 128)) : 0; \
        if (lslots_ > 0 && (lat_mode() == 2 || GRID <= lslots_)) { LAUNCH2L(M, NS_, H_, kFormLat, lslots_, GRID, LDS, CAP); stats_.lat_launches++; } \
        else { /* (no lean form of this kernel: measured, the f32 insert search LOSES 6 % with it -- profiles/r5_lean_ab.log) */ \
            const int slots_ = std::min(max_slots(), resident_blocks(graph_insert_search_kernel<M, NS_, H_, kFormPlain>, LDS, num_cu_)); \
            LAUNCH2L(M, NS_, H_, kFormPlain, slots_, GRID, LDS, CAP); \
        } \
    } while (0)
#define LAUNCH3(NS_, H_, GRID, LDS, CAP)                                                                              \
    do {                                                                                                                   \
        if (metric_ == M_SQ) LAUNCH2(M_SQ, NS_, H_, GRID, LDS, CAP);                                                  \
        else if (metric_ == M_COS) LAUNCH2(M_COS, NS_, H_, GRID, LDS, CAP);                                           \
        else if (metric_ == M_I8) LAUNCH2(M_I8, NS_, H_, GRID, LDS, CAP);                                             \
        else LAUNCH2(M_UCOS, NS_, H_, GRID, LDS, CAP);                                                                \
    } while (0)
#define LAUNCH(NS_, GRID, LDS, CAP)                                                                                   \
    do {                                                                                                                   \
        if (vis_tab) LAUNCH3(NS_, true, GRID, LDS, CAP);                                                              \
        else LAUNCH3(NS_, false, GRID, LDS, CAP);                                                                     \
    } while (0)
        switch (ns) {
        case 2: LAUNCH(2, nj, lds, cand_cap); break;
        case 4: LAUNCH(4, nj, lds, cand_cap); break;
        default: LAUNCH3(8, false, nj, lds, cand_cap); break; // (no hash-table form: visited_table() was told so)
        }
        HIP_OK(hipGetLastError());
#undef LAUNCH
#undef LAUNCH3
#undef LAUNCH2
#undef LAUNCH2L
        if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev1_, st));
        if (!windowed) HIP_OK(hipMemcpyAsync(h_ev, s_evals_, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        if (windowed) { // one launch (njobs <= chunk): everything the host validates with rides on the same wait
            int *d_owner = s_wdry_, *d_dry0 = p_log + b_log / 4u, *d_dryU = d_dry0 + (size_t)njobs * sel_stride, *d_drop0 = d_dry0 + b_dry / 4u;
            if (n_upper > 0) {
                memcpy(h_log + b_log / 4u, win->upper_owner, 4u * (size_t)n_upper); // staged behind the logs (the dry flags land there afterwards)
                HIP_OK(hipMemcpyAsync(d_owner, h_log + b_log / 4u, 4u * (size_t)n_upper, hipMemcpyHostToDevice, st));
            }
            HIP_OK(hipMemsetAsync(d_dry0, 0xff, b_dry, st)); // every code "changed, set of lost ids unknown" unless the kernel says otherwise
            const int k_cap = nbcap();
            const size_t lds_link = ((search_lds_bytes(k_cap, 0, pitch_, true, nbcap()) + 15) & ~(size_t)15) + 4u * (size_t)(kNewMax + 1) * nbcap();
            const int grid = (njobs + n_upper) * sel_stride;
#define LAUNCH_DRY(M)                                                                                                          \
    hipLaunchKernelGGL(graph_link_dry_sel_kernel<M>, dim3(grid), dim3(64), lds_link, st, d_rows_, d_row_sn_, pitch_, g_adj0_, g_stride0_, \
                       g_upper_, g_pool_, g_strideU_, s_jobs_, p_flag, p_sel0, p_cnt0, p_selU, p_cntU, sel_stride, d_owner, njobs, max_edges0, \
                       k_cap, d_dry0, d_dryU, p_evals, nbcap(), g_tested0_, g_testedU_, g_n_, d_drop0)
            if (metric_ == M_SQ) LAUNCH_DRY(M_SQ);
            else if (metric_ == M_COS) LAUNCH_DRY(M_COS);
            else if (metric_ == M_I8) LAUNCH_DRY(M_I8);
            else LAUNCH_DRY(M_UCOS);
#undef LAUNCH_DRY
            HIP_OK(hipGetLastError());
            HIP_OK(hipMemcpyAsync(hb, db, need - b_rep, hipMemcpyDeviceToHost, st)); // selections, counts, flags, evaluations, read logs, dry-run codes, lost ids
        }
        const bool last_chunk = off + nj >= njobs;
        if (!windowed && last_chunk) HIP_OK(hipMemcpyAsync(h_flag, s_iflag_, b_flag, hipMemcpyDeviceToHost, st)); // the flags ride on the same wait
        HIP_OK(hipStreamSynchronize(st)); // the job staging buffer is reused by the next chunk
        stats_.search_launches++;
        stats_.search_evals += *h_ev;
        stats_.insert_launches++;
        stats_.insert_evals += *h_ev;
        if (vis_tab) stats_.visited_hash_launches++;
        if (timed) {
            float ms = 0.f;
            HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)ev0_, (hipEvent_t)ev1_));
            stats_.search_kernel_ms += ms;
            stats_.search_timed_launches++;
            stats_.search_timed_evals += *h_ev;
            stats_.insert_kernel_ms += ms;
            stats_.insert_timed_launches++;
            stats_.insert_timed_evals += *h_ev;
        }
    }
    // Only the flags come back (with the last launch's wait): the selections stay on the device, where the link half reads
    // them (link_batch_planned); a caller that links on the host fetches them (fetch_insert_selections).
    int *h_rep = windowed ? h_log + (b_log + b_dry + b_drop) / 4u : nullptr;
    for (int i = 0; i < njobs; ++i) {
        if (h_rep) h_rep[i] = h_flag[i] == 2;
        if (h_flag[i] == 2) { stats_.search_repeats++; stats_.insert_tie_reruns++; h_flag[i] = 0; }
        stats_.search_overflows += (uint64_t)(h_flag[i] != 0);
    }
    last_insert_jobs_ = (njobs <= chunk && !windowed) ? njobs : 0; // a single launch left everything in place (windowed: in its own block)
    last_insert_upper_ = n_upper;
    last_insert_stride_ = sel_stride;
    fetch_njobs_ = windowed ? 0 : njobs;
    fetch_nupper_ = n_upper;
    *res = InsertResults{h_sel0, h_cnt0, h_selU, h_cntU, h_flag, sel_stride};
    if (windowed) {
        win->read_log = h_log; win->dry0 = h_log + b_log / 4u; win->dryU = win->dry0 + (size_t)njobs * sel_stride;
        win->drop0 = win->dry0 + b_dry / 4u; win->repeated = h_rep;
    }
    return true;
}

// The selected ids of the last insert_search_batch, into the pinned arrays its InsertResults name.
bool Device::fetch_insert_selections(const InsertResults *res)
{
    if (!res || fetch_njobs_ <= 0) return true;
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    const int njobs = fetch_njobs_, n_upper = fetch_nupper_, sel_stride = res->sel_stride;
    HIP_OK(hipMemcpyAsync(const_cast<int *>(res->sel0), s_sel_, 4u * (size_t)njobs * sel_stride, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(const_cast<int *>(res->cnt0), s_lcnt_, 4u * (size_t)njobs, hipMemcpyDeviceToHost, st));
    if (n_upper > 0) {
        HIP_OK(hipMemcpyAsync(const_cast<int *>(res->selU), s_selU_, 4u * (size_t)n_upper * sel_stride, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(const_cast<int *>(res->cntU), s_cntU_, 4u * (size_t)n_upper, hipMemcpyDeviceToHost, st));
    }
    HIP_OK(hipStreamSynchronize(st));
    fetch_njobs_ = 0;
    return true;
}

bool Device::traversal_fits(int k, bool with_heuristic, int max_edges) const
{
    if (k < 1 || 2 * max_edges + 1 > 128) return false;
    const int nb = std::max(8, (2 * max_edges + 1 + 7) & ~7);
    const int cap = cand_lds_cap(k, pitch_, with_heuristic, nb);
    return search_lds_bytes(k, cap, pitch_, with_heuristic, nb) <= 64 * 1024 && search_lds_bytes(nb, 0, pitch_, true, nb) <= 64 * 1024;
}

bool Device::graph_append_nodes(long long first, long long n, const int *level, const int64_t *upper, const int *pool,
                                long long pool_from, long long pool_len, bool *need_full_sync)
{
    *need_full_sync = false;
    if (n <= 0) return true;
    if (!g_adj0_ || first != g_n_ || first + n > g_cap_n_ || pool_len > g_pool_cap_) { *need_full_sync = true; return true; }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    HIP_OK(hipMemcpyAsync(g_level_ + first, level + first, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(g_upper_ + first, upper + first, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
    // new nodes start with empty lists (GraphData.NewNode :224-242)
    HIP_OK(hipMemsetAsync(g_adj0_ + (size_t)first * g_stride0_, 0, sizeof(int) * (size_t)n * g_stride0_, st));
    HIP_OK(hipMemsetAsync(g_tested0_ + first, 0, sizeof(int) * (size_t)n, st));
    if (pool_len > pool_from) HIP_OK(hipMemsetAsync(g_testedU_ + pool_from / std::max(1, g_strideU_), 0, sizeof(int) * (size_t)((pool_len - pool_from) / std::max(1, g_strideU_) + 1), st));
    if (pool_len > pool_from) HIP_OK(hipMemcpyAsync(g_pool_ + pool_from, pool + pool_from, sizeof(int) * (size_t)(pool_len - pool_from), hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    g_n_ = first + n;
    return true;
}

bool Device::link_batch(const int *rows, int nrows, int row_stride, const int *g_node, const int *g_layer, const int *g_off,
                        const int *g_items, int ngroups, int max_edges0, int *out_lists, int list_stride)
{
    if (ngroups > 0 && !out_lists) { set_dev_error("link_batch: bad argument"); return false; }
    const int *res = nullptr;
    if (!link_batch_begin(0, rows, nrows, row_stride, g_node, g_layer, g_off, g_items, ngroups, max_edges0, list_stride)) return false;
    if (!link_batch_finish(0, &res)) return false;
    if (ngroups > 0) memcpy(out_lists, res, sizeof(int) * (size_t)ngroups * list_stride);
    return true;
}

bool Device::link_batch_begin(int set, const int *rows, int nrows, int row_stride, const int *g_node, const int *g_layer, const int *g_off,
                              const int *g_items, int ngroups, int max_edges0, int list_stride, bool want_lists)
{
    if (set < 0 || set > 1 || nrows < 0 || ngroups < 0 || (nrows > 0 && !rows) || (ngroups > 0 && (!g_node || !g_layer || !g_off || !g_items))) {
        set_dev_error("link_batch: bad argument");
        return false;
    }
    LinkSet &ls = lset_[set];
    if (ls.busy) { set_dev_error("link_batch: staging set still in flight"); return false; }
    if (g_n_ <= 0) { set_dev_error("link_batch: no graph uploaded"); return false; }
    // host-side validation: a bad id must be an error return, never a GPU fault
    for (int r = 0; r < nrows; ++r) {
        const int *x = rows + (size_t)r * row_stride;
        bool ok = x[0] >= 0 && x[0] < g_n_ && x[1] >= 0 && x[2] >= 0 && x[2] <= row_stride - 3 &&
                  x[2] <= (x[1] == 0 ? max_edges0 : max_edges0 / 2);
        for (int i = 0; ok && i < x[2]; ++i) ok = x[3 + i] >= 0 && x[3 + i] < g_n_;
        if (!ok) { set_dev_error("link_batch: row record outside the graph"); return false; }
    }
    const int total = ngroups > 0 ? g_off[ngroups] : 0;
    for (int g = 0; g < ngroups; ++g)
        if (g_node[g] < 0 || g_node[g] >= g_n_ || g_layer[g] < 0 || g_off[g + 1] < g_off[g]) { set_dev_error("link_batch: group outside the graph"); return false; }
    for (int t = 0; t < total; ++t)
        if (g_items[t] < 0 || g_items[t] >= g_n_) { set_dev_error("link_batch: item outside the graph"); return false; }
    if (list_stride < max_edges0 + 1 || max_edges0 + 1 > nbcap()) { set_dev_error("link_batch: list stride too small"); return false; }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    if (!ensure_search_scratch(1, 1, 0, 16)) return false;
    const size_t need[5] = {(size_t)nrows * row_stride, (size_t)ngroups * 2, (size_t)ngroups + 1, (size_t)std::max(total, 1), (size_t)ngroups * list_stride};
    // device buffers are shared by both sets: the stream orders one sub-batch after the other.
    // Growing one frees the old allocation, which must not be in use any more.
    for (int i = 0; i < 5; ++i) {
        if (std::max<size_t>(need[i], 1) > s_lk_cap_[i]) {
            HIP_OK(hipStreamSynchronize(st));
            if (!grow_dev(&s_lk_[i], &s_lk_cap_[i], std::max<size_t>(need[i], 1) * 2)) return false;
        }
    }
    // pinned staging of this set: [rows | node | layer | off | items], results, evaluation count
    const size_t in_ints = need[0] + need[1] + need[2] + need[3];
    if (in_ints > ls.in_cap) {
        if (ls.h_in) (void)hipHostFree(ls.h_in);
        ls.h_in = nullptr; ls.in_cap = 0;
        if (hipHostMalloc((void **)&ls.h_in, sizeof(int) * in_ints * 2, hipHostMallocDefault) != hipSuccess) { set_dev_error("link_batch: pinned allocation failed"); return false; }
        ls.in_cap = in_ints * 2;
    }
    if (std::max<size_t>(need[4], 1) > ls.out_cap) {
        if (ls.h_out) (void)hipHostFree(ls.h_out);
        ls.h_out = nullptr; ls.out_cap = 0;
        if (hipHostMalloc((void **)&ls.h_out, sizeof(int) * std::max<size_t>(need[4], 1) * 2, hipHostMallocDefault) != hipSuccess) { set_dev_error("link_batch: pinned allocation failed"); return false; }
        ls.out_cap = std::max<size_t>(need[4], 1) * 2;
    }
    if (!ls.h_ev && hipHostMalloc((void **)&ls.h_ev, 16, hipHostMallocDefault) != hipSuccess) { set_dev_error("link_batch: pinned allocation failed"); return false; }
    if (!ls.ev_done) {
        hipEvent_t a, b, c;
        HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b)); HIP_OK(hipEventCreate(&c));
        ls.ev_start = a; ls.ev_stop = b; ls.ev_done = c;
    }
    int *h_rows = ls.h_in, *h_node = h_rows + need[0], *h_off = h_node + need[1], *h_items = h_off + need[2];
    if (nrows > 0) memcpy(h_rows, rows, sizeof(int) * need[0]);
    if (ngroups > 0) {
        memcpy(h_node, g_node, sizeof(int) * (size_t)ngroups);
        memcpy(h_node + ngroups, g_layer, sizeof(int) * (size_t)ngroups);
        memcpy(h_off, g_off, sizeof(int) * ((size_t)ngroups + 1));
        memcpy(h_items, g_items, sizeof(int) * (size_t)total);
    }
    *ls.h_ev = 0;
    ls.ngroups = ngroups;
    ls.timed = false;
    if (nrows > 0) {
        HIP_OK(hipMemcpyAsync(s_lk_[0], h_rows, sizeof(int) * need[0], hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(graph_write_rows_kernel, dim3(nrows), dim3(64), 0, st, g_adj0_, g_stride0_, g_upper_, g_pool_, g_strideU_, s_lk_[0], row_stride,
                           g_tested0_, g_testedU_, max_edges0);
        HIP_OK(hipGetLastError());
    }
    if (ngroups > 0) {
        HIP_OK(hipMemcpyAsync(s_lk_[1], h_node, sizeof(int) * (size_t)ngroups * 2, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(s_lk_[2], h_off, sizeof(int) * ((size_t)ngroups + 1), hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(s_lk_[3], h_items, sizeof(int) * (size_t)total, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemsetAsync(s_evals_, 0, sizeof(unsigned long long), st));
        ls.timed = profiling_;
        if (ls.timed) HIP_OK(hipEventRecord((hipEvent_t)ls.ev_start, st));
        const int k_cap = nbcap();
        const size_t lds = ((search_lds_bytes(k_cap, 0, pitch_, true, nbcap()) + 15) & ~(size_t)15) + 4u * (size_t)(kNewMax + 1) * nbcap();
#define LAUNCH(M)                                                                                                          \
    hipLaunchKernelGGL(graph_link_kernel<M>, dim3(ngroups), dim3(64), lds, st, d_rows_, d_row_sn_, pitch_, g_adj0_, g_stride0_,  \
                       g_upper_, g_pool_, g_strideU_, s_lk_[1], s_lk_[1] + ngroups, s_lk_[2], (const int *)nullptr, s_lk_[3], max_edges0, k_cap, \
                       s_lk_[4], list_stride, s_evals_, nbcap(), g_tested0_, g_testedU_, (const int *)nullptr)
        if (metric_ == M_SQ) LAUNCH(M_SQ);
        else if (metric_ == M_COS) LAUNCH(M_COS);
        else if (metric_ == M_I8) LAUNCH(M_I8);
        else LAUNCH(M_UCOS);
#undef LAUNCH
        HIP_OK(hipGetLastError());
        if (ls.timed) HIP_OK(hipEventRecord((hipEvent_t)ls.ev_stop, st));
        if (want_lists) HIP_OK(hipMemcpyAsync(ls.h_out, s_lk_[4], sizeof(int) * (size_t)ngroups * list_stride, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(ls.h_ev, s_evals_, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    }
    HIP_OK(hipEventRecord((hipEvent_t)ls.ev_done, st));
    ls.busy = true;
    return true;
}

bool Device::link_dry_run(const int *jobs3, int n, int max_edges0, int *changed)
{
    if (n <= 0) return true;
    if (!jobs3 || !changed || max_edges0 + 1 > nbcap()) { set_dev_error("link_dry_run: bad argument"); return false; }
    if (g_n_ <= 0) { set_dev_error("link_dry_run: no graph uploaded"); return false; }
    for (int i = 0; i < n; ++i) { // a bad id must be an error return, never a GPU fault
        const int nb = jobs3[3 * i], layer = jobs3[3 * i + 1], item = jobs3[3 * i + 2];
        if (nb < 0 || nb >= g_n_ || item < 0 || item >= g_n_ || layer < 0) { set_dev_error("link_dry_run: job outside the graph"); return false; }
    }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    if (!ensure_search_scratch(1, 1, 0, 16)) return false;
    if (!grow_dev(&s_dry_, &s_dry_cap_, (size_t)n * 4)) return false;
    int *h = static_cast<int *>(pinned_stage(sizeof(int) * (size_t)n * 4 + 16));
    if (!h) return false;
    unsigned long long *h_ev = reinterpret_cast<unsigned long long *>(h + (((size_t)n * 4 + 1) & ~(size_t)1));
    memcpy(h, jobs3, sizeof(int) * (size_t)n * 3);
    HIP_OK(hipMemcpyAsync(s_dry_, h, sizeof(int) * (size_t)n * 3, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(s_evals_, 0, sizeof(unsigned long long), st));
    const int k_cap = nbcap();
    const size_t lds = ((search_lds_bytes(k_cap, 0, pitch_, true, nbcap()) + 15) & ~(size_t)15) + 4u * (size_t)(kNewMax + 1) * nbcap();
#define LAUNCH(M)                                                                                                          \
    hipLaunchKernelGGL(graph_link_dry_kernel<M>, dim3(n), dim3(64), lds, st, d_rows_, d_row_sn_, pitch_, g_adj0_, g_stride0_,   \
                       g_upper_, g_pool_, g_strideU_, s_dry_, max_edges0, k_cap, s_dry_ + (size_t)n * 3, s_evals_, nbcap(), g_tested0_, g_testedU_)
    if (metric_ == M_SQ) LAUNCH(M_SQ);
    else if (metric_ == M_COS) LAUNCH(M_COS);
    else if (metric_ == M_I8) LAUNCH(M_I8);
    else LAUNCH(M_UCOS);
#undef LAUNCH
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(h + (size_t)n * 3, s_dry_ + (size_t)n * 3, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(h_ev, s_evals_, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    memcpy(changed, h + (size_t)n * 3, sizeof(int) * (size_t)n);
    stats_.search_launches++;
    stats_.search_evals += *h_ev;
    stats_.link_evals += *h_ev;
    return true;
}

bool Device::link_batch_planned(int njobs, int n_upper, int max_edges0)
{
    if (njobs <= 0) return true;
    if (njobs != last_insert_jobs_ || n_upper != last_insert_upper_ || max_edges0 != last_insert_stride_ || max_edges0 + 1 > nbcap()) {
        set_dev_error("link_batch_planned: no matching insert_search_batch results on the device");
        return false;
    }
    last_insert_jobs_ = 0;
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    if (!ensure_search_scratch(1, 1, 0, 16)) return false;
    // per-slot counters: all zero between batches (the link kernel resets what it used)
    const long long slots = g_cap_n_ + g_pool_cap_ / std::max(1, g_strideU_) + 2;
    if (slots != lp_slots_) {
        HIP_OK(hipStreamSynchronize(st));
        for (int i = 0; i < 3; ++i) {
            if (lp_slot_[i]) HIP_OK(hipFree(lp_slot_[i]));
            lp_slot_[i] = nullptr;
            HIP_OK(hipMalloc(&lp_slot_[i], sizeof(int) * (size_t)slots));
            HIP_OK(hipMemsetAsync(lp_slot_[i], 0, sizeof(int) * (size_t)slots, st));
        }
        lp_slots_ = slots;
    }
    const size_t max_appends = (size_t)(njobs + n_upper) * (size_t)max_edges0 + 1;
    for (int i = 0; i < 6; ++i) {
        if (max_appends > lp_grp_cap_[i]) {
            HIP_OK(hipStreamSynchronize(st));
            if (!grow_dev(&lp_grp_[i], &lp_grp_cap_[i], max_appends * 2)) return false;
        }
    }
    if (!lp_counters_) HIP_OK(hipMalloc(&lp_counters_, sizeof(int) * 4));
    HIP_OK(hipMemsetAsync(lp_counters_, 0, sizeof(int) * 4, st));
    HIP_OK(hipMemsetAsync(s_evals_, 0, sizeof(unsigned long long), st));
    size_t g_cap = lp_grp_cap_[0];
    for (int i = 1; i < 6; ++i) g_cap = std::min(g_cap, lp_grp_cap_[i]);
    LinkPlan P{lp_slot_[0], lp_slot_[1], lp_slot_[2], lp_grp_[0], lp_grp_[1], lp_grp_[2], lp_grp_[3], lp_grp_[4], lp_counters_, g_cap_n_,
               slots, g_n_, (int)std::min<size_t>(g_cap, 0x7fffffff), njobs};
    const bool timed = profiling_;
    if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev0_, st));
    hipLaunchKernelGGL(link_plan_kernel<true>, dim3(njobs), dim3(64), 0, st, s_jobs_, s_sel_, s_lcnt_, s_selU_, s_cntU_, last_insert_stride_,
                       g_adj0_, g_stride0_, g_upper_, g_pool_, g_strideU_, g_tested0_, g_testedU_, max_edges0, P);
    hipLaunchKernelGGL(link_offsets_kernel, dim3(512), dim3(256), 0, st, g_upper_, g_strideU_, P);
    hipLaunchKernelGGL(link_plan_kernel<false>, dim3(njobs), dim3(64), 0, st, s_jobs_, s_sel_, s_lcnt_, s_selU_, s_cntU_, last_insert_stride_,
                       g_adj0_, g_stride0_, g_upper_, g_pool_, g_strideU_, g_tested0_, g_testedU_, max_edges0, P);
    HIP_OK(hipGetLastError());
    // the number of groups comes back to size the last two launches (16 bytes, one short wait) -- except for small batches
    // (bounded-concurrency Add: a few hundred items per batch, a batch every 1-3 ms), which launch the upper bound
    // instead and let the surplus blocks leave at once
    unsigned long long *h_ev = static_cast<unsigned long long *>(pinned_stage(32));
    if (!h_ev) return false;
    int *h_ctr = reinterpret_cast<int *>(h_ev + 1);
    const bool bounded = njobs <= 4096;
    int G = (int)std::min<size_t>(max_appends - 1, g_cap);
    if (!bounded) {
        HIP_OK(hipMemcpyAsync(h_ctr, lp_counters_, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        G = h_ctr[0];
        if (h_ctr[3] != 0 || G < 0 || (size_t)G > g_cap) {
            set_dev_error("link_batch_planned: inconsistent selection data on the device (guard " + std::to_string(h_ctr[3]) + ")");
            return false;
        }
    }
    if (G > 0) {
        hipLaunchKernelGGL(link_order_kernel, dim3(G), dim3(64), 0, st, s_jobs_, g_upper_, g_strideU_, lp_grp_[5], P, bounded ? 1 : 0);
        HIP_OK(hipGetLastError());
        const int k_cap = nbcap();
        const size_t lds = ((search_lds_bytes(k_cap, 0, pitch_, true, nbcap()) + 15) & ~(size_t)15) + 4u * (size_t)(kNewMax + 1) * nbcap();
#define LAUNCH(M)                                                                                                          \
    hipLaunchKernelGGL(graph_link_kernel<M>, dim3(G), dim3(64), lds, st, d_rows_, d_row_sn_, pitch_, g_adj0_, g_stride0_,       \
                       g_upper_, g_pool_, g_strideU_, lp_grp_[0], lp_grp_[1], lp_grp_[2], lp_grp_[3], lp_grp_[5], max_edges0, k_cap, \
                       (int *)nullptr, 0, s_evals_, nbcap(), g_tested0_, g_testedU_, bounded ? (const int *)lp_counters_ : (const int *)nullptr)
        if (metric_ == M_SQ) LAUNCH(M_SQ);
        else if (metric_ == M_COS) LAUNCH(M_COS);
        else if (metric_ == M_I8) LAUNCH(M_I8);
        else LAUNCH(M_UCOS);
#undef LAUNCH
        HIP_OK(hipGetLastError());
    }
    if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev1_, st));
    HIP_OK(hipMemcpyAsync(h_ev, s_evals_, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(h_ctr, lp_counters_, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (h_ctr[3] != 0 || h_ctr[0] < 0 || (size_t)h_ctr[0] > g_cap) {
        set_dev_error("link_batch_planned: inconsistent selection data on the device (guard " + std::to_string(h_ctr[3]) + ")");
        return false;
    }
    stats_.search_launches++;
    stats_.search_evals += *h_ev;
    stats_.link_launches++;
    stats_.link_evals += *h_ev;
    if (timed) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)ev0_, (hipEvent_t)ev1_));
        stats_.search_kernel_ms += ms;
        stats_.search_timed_launches++;
        stats_.search_timed_evals += *h_ev;
        stats_.link_kernel_ms += ms;
        stats_.link_timed_launches++;
        stats_.link_timed_evals += *h_ev;
    }
    return true;
}

bool Device::download_graph(int *adj0, long long n, int *pool, long long pool_len)
{
    if (n < 0 || n > g_n_ || pool_len < 0 || pool_len > g_pool_cap_ || (n > 0 && !adj0) || (pool_len > 0 && !pool)) { set_dev_error("download_graph: bad argument"); return false; }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    if (n > 0) HIP_OK(hipMemcpyAsync(adj0, g_adj0_, sizeof(int) * (size_t)n * g_stride0_, hipMemcpyDeviceToHost, st));
    if (pool_len > 0) HIP_OK(hipMemcpyAsync(pool, g_pool_, sizeof(int) * (size_t)pool_len, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return true;
}

bool Device::link_batch_finish(int set, const int **out_lists)
{
    if (set < 0 || set > 1 || !lset_[set].busy) { set_dev_error("link_batch_finish: nothing in flight"); return false; }
    LinkSet &ls = lset_[set];
    if (!bind()) return false;
    HIP_OK(hipEventSynchronize((hipEvent_t)ls.ev_done));
    ls.busy = false;
    if (out_lists) *out_lists = ls.h_out;
    if (ls.ngroups > 0) {
        stats_.search_launches++;
        stats_.search_evals += *ls.h_ev;
        stats_.link_launches++;
        stats_.link_evals += *ls.h_ev;
        if (ls.timed) {
            float ms = 0.f;
            HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)ls.ev_start, (hipEvent_t)ls.ev_stop));
            stats_.search_kernel_ms += ms;
            stats_.search_timed_launches++;
            stats_.search_timed_evals += *ls.h_ev;
            stats_.link_kernel_ms += ms;
            stats_.link_timed_launches++;
            stats_.link_timed_evals += *ls.h_ev;
        }
    }
    return true;
}


// Pinned host staging (grown on demand): DMA to/from pageable user memory runs at ~2 GB/s,
// through a pinned bounce buffer at PCIe rate.
void *Device::pinned_stage(size_t bytes)
{
    if (bytes <= h_stage_cap_) return h_stage_;
    if (h_stage_) (void)hipHostFree(h_stage_);
    h_stage_ = nullptr;
    h_stage_cap_ = 0;
    size_t cap = std::max<size_t>(bytes, 1u << 20);
    if (hipHostMalloc(&h_stage_, cap, hipHostMallocDefault) != hipSuccess) { set_dev_error("pinned staging allocation failed"); return nullptr; }
    h_stage_cap_ = cap;
    return h_stage_;
}

// KnnQuery on the device: descent + layer-0 beam search (width k) + the stable top-k_out tail.
// out_ids / out_d: njobs x k_out, final (padded with -1 / NaN); out_flag: 1 = not run to
// completion (candidate heap beyond LDS + spill), caller re-runs that job on the lock-step path.
bool Device::search_batch(const SearchJob *jobs, int njobs, int k, int k_out, int *out_ids, float *out_d, int *out_flag, bool keep_repeat_flag,
                          bool two_heap)
{
    if (njobs <= 0) return true;
    if (!jobs) { set_dev_error("search_batch: bad argument"); return false; }
    return search_batch_impl(jobs, njobs, k, k_out, out_ids, out_d, out_flag, keep_repeat_flag, two_heap, -1, -1);
}

bool Device::search_queries(int nq, int entry, int entry_layer, int k, int k_out, int *out_ids, float *out_d, int *out_flag)
{
    if (nq <= 0) return true;
    if (nq > (1 << 20)) { // more than one launch: the plain path
        std::vector<SearchJob> jobs((size_t)nq);
        for (int i = 0; i < nq; ++i) jobs[(size_t)i] = SearchJob{i, entry, entry_layer, 0, -1, 0};
        return search_batch(jobs.data(), nq, k, k_out, out_ids, out_d, out_flag);
    }
    return search_batch_impl(nullptr, nq, k, k_out, out_ids, out_d, out_flag, false, false, entry, entry_layer);
}

// jobs == nullptr: search_queries' jobs {i, u_entry, u_layer, 0, -1, 0}, i < njobs <= 2^20
bool Device::search_batch_impl(const SearchJob *jobs, int njobs, int k, int k_out, int *out_ids, float *out_d, int *out_flag, bool keep_repeat_flag,
                               bool two_heap, int u_entry, int u_layer)
{
    if (!out_ids || !out_d || !out_flag || k < 1 || k_out < 1) { set_dev_error("search_batch: bad argument"); return false; }
    if (g_n_ <= 0) { set_dev_error("search_batch: no graph uploaded"); return false; }
    if (jobs ? !jobs_valid(jobs, njobs, g_n_, n_queries_, n_rows_hw_) : !(u_entry >= 0 && u_entry < g_n_ && u_layer >= 0 && njobs <= n_queries_)) {
        set_dev_error("search_batch: job outside the uploaded graph / rows / queries");
        return false;
    }
    const int cand_cap = cand_lds_cap(k, pitch_, false, nbcap());
    const size_t lds = search_lds_bytes(k, cand_cap, pitch_, false, nbcap());
    if (lds > 64 * 1024) { set_dev_error("search_batch: beam width / dimension exceed the LDS budget"); return false; }
    const int ns_req = (g_n_ < kSortedTopMaxNodes && !two_heap) ? sorted_top_sets(k) : 0;
    const bool exact_only = ns_req == 0;           // two_heap callers, beams beyond 512 entries, HNSW_MI355X_SORTED_TOP=0: launch flag 0x200
    const int ns = exact_only ? 2 : ns_req;
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    long long vis_words = ((g_n_ + 31) / 32 + 3) & ~3LL;
    size_t vis_bytes_per_job = sizeof(unsigned) * (size_t)vis_words;
    const long long chunk = std::min<long long>(njobs, 1 << 20);
    int *vis_tab = nullptr;
    int vis_tab_cap = 0;
    if (!visited_table(vis_bytes_per_job, k, &vis_tab, &vis_tab_cap, 512, ns != 8)) return false;
    if (vis_tab) { vis_words = 0; vis_bytes_per_job = 16; } // see insert_search_batch
    if (!ensure_search_scratch(chunk, max_slots(), k_out, vis_bytes_per_job)) return false;
    // pinned layout: [evals (16 B) | jobs | ids | dists | flags]
    const size_t b_jobs = sizeof(SearchJob) * (size_t)chunk, b_res = 4u * (size_t)chunk * k_out;
    char *hs = static_cast<char *>(pinned_stage(16 + b_jobs + 2 * b_res + 4u * (size_t)chunk));
    if (!hs) return false;
    unsigned long long *h_ev = reinterpret_cast<unsigned long long *>(hs);
    SearchJob *h_jobs = reinterpret_cast<SearchJob *>(hs + 16);
    int *h_ids = reinterpret_cast<int *>(hs + 16 + b_jobs);
    float *h_d = reinterpret_cast<float *>(hs + 16 + b_jobs + b_res);
    int *h_flag = reinterpret_cast<int *>(hs + 16 + b_jobs + 2 * b_res);
    int *d_ids = reinterpret_cast<int *>(s_hits_);
    float *d_d = reinterpret_cast<float *>(s_hits_) + (size_t)chunk * k_out;
    // A call answered by ONE launch of modest size is mostly API calls around a latency-bound kernel (a single query: 0.37 ms
    // of kernel in a 0.45-ms call, eleven HIP calls): ids, distances and flags then sit in one device slab and come back in
    // one copy into the pinned staging (laid out alike), the evaluation counter lives in the job counter's spare words and
    // is zeroed with it -- five HIP calls.  (A large call keeps the separate copies: its ids are handed to the caller while
    // the distances are still crossing the link.)
    const bool compact = njobs == chunk && 2 * b_res + 4u * (size_t)chunk <= (size_t)1 << 20;
    int *d_flag = compact ? reinterpret_cast<int *>(d_d + (size_t)chunk * k_out) : s_flag_;
    unsigned long long *d_ev = compact ? reinterpret_cast<unsigned long long *>(s_jobctr_ + 2) : s_evals_;
    // (rows of more than 1 KB: the set's traffic is small beside the rows', and the rows of re-seen neighbours are what costs --
    //  C3's 3-KB rows: 282.7 k queries/s with the sets against 273.6 k without, build 69.5 k against 61.3 k adds/s)
    const bool novis_ = g_stride0_ - 2 <= 64 && overlap_mode() != 0 && (size_t)pitch_ * sizeof(float) <= 1024 && (novis_mode() == 2 || (novis_mode() == 1 && vis_tab != nullptr));
    // a query set whose tail is still on the host (set_queries_streamed): the launch is gated on the rows' arrival
    const int *gate = tail_.n > 0 ? d_ready_ : nullptr;
    // whatever happens below, nothing stays pending -- and a tail that never went up (an error between the launch and
    // upload_tail) leaves no resident query set behind: hnsw_mi355x_knn_query_resident must not answer from rows that
    // were never uploaded
    struct TailGuard { Device *d; ~TailGuard() { if (d->tail_.n > 0) d->n_queries_ = 0; d->tail_.n = 0; } } tail_guard{this};
    for (long long off = 0; off < njobs; off += chunk) {
        const int nj = (int)std::min<long long>(chunk, njobs - off);
        if (jobs) {
            memcpy(h_jobs, jobs + off, sizeof(SearchJob) * (size_t)nj);
            HIP_OK(hipMemcpyAsync(s_jobs_, h_jobs, sizeof(SearchJob) * (size_t)nj, hipMemcpyHostToDevice, st));
            uj_len_ = 0;
        } else if (!(uj_len_ >= nj && uj_entry_ == u_entry && uj_layer_ == u_layer)) {
            for (int i = 0; i < nj; ++i) h_jobs[i] = SearchJob{i, u_entry, u_layer, 0, -1, 0};
            HIP_OK(hipMemcpyAsync(s_jobs_, h_jobs, sizeof(SearchJob) * (size_t)nj, hipMemcpyHostToDevice, st));
            uj_len_ = nj; uj_entry_ = u_entry; uj_layer_ = u_layer;
        }
        HIP_OK(hipMemsetAsync(s_jobctr_, 0, sizeof(int) * (4 + (size_t)nj), st));
        if (!compact) HIP_OK(hipMemsetAsync(s_evals_, 0, sizeof(unsigned long long), st));
        const bool timed = profiling_;
        if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev0_, st));
#define LAUNCH2L(M, NS_, H_, LAT_, SLOTS, GRID, LDS, CAP) \
        hipLaunchKernelGGL((graph_search_kernel<M, NS_, H_, LAT_>), dim3(std::min<int>(GRID, SLOTS)), \
                       dim3(LAT_ == kFormLat ? 128 : 64), (LDS) + (LAT_ == kFormLat ? kTeamLds : 0), st, d_rows_, d_row_sn_, d_queries_, d_q_sn_, pitch_, \
                       g_adj0_, g_stride0_, g_upper_, g_pool_, g_strideU_, s_jobs_, k, CAP, reinterpret_cast<ND *>(s_spill_), \
                       spill_cap_for_tests(), s_visited_, vis_words, vis_tab, vis_tab_cap, k_out, d_ids, d_d, s_cnt_, d_flag, d_ev, nbcap(), GRID, s_jobctr_, (exact_only ? 0x200 : 0) | (novis_ ? 9 : (overlap_mode() == 2 || (overlap_mode() == 1 && (GRID <= SLOTS || vis_tab != nullptr))) ? 1 : 0) | (shadow_mode() && shadows_allowed_ ? 0x100 : 0), \
                       gate)
#define LAUNCH2(M, NS_, H_, GRID, LDS, CAP) \
    do { \
        const int lslots_ = !exact_only && lat_mode() != 0 && g_stride0_ - 2 <= 64 && (LDS) + kTeamLds <= 64 * 1024 ? std::min(max_slots(), resident_blocks(graph_search_kernel<M, NS_, H_, kFormLat>, (LDS) + kTeamLds, num_cu_, 128)) : 0; \
        if (lslots_ > 0 && (lat_mode() == 2 || GRID <= lslots_)) { LAUNCH2L(M, NS_, H_, kFormLat, lslots_, GRID, LDS, CAP); stats_.lat_launches++; } \
        else if (novis_ && !exact_only && lean_mode() && NS_ <= 4) { /* flags 9: the lean form (kFormLean; NS = 8 has none) */ \
            const int slots_ = std::min(max_slots(), resident_blocks(graph_search_kernel<M, NS_, H_, (NS_ <= 4 ? kFormLean : kFormPlain)>, LDS, num_cu_)); \
            LAUNCH2L(M, NS_, H_, (NS_ <= 4 ? kFormLean : kFormPlain), slots_, GRID, LDS, CAP); stats_.lean_launches++; \
        } else { \
            const int slots_ = std::min(max_slots(), resident_blocks(graph_search_kernel<M, NS_, H_, kFormPlain>, LDS, num_cu_)); \
            LAUNCH2L(M, NS_, H_, kFormPlain, slots_, GRID, LDS, CAP); \
        } \
    } while (0)
#define LAUNCH3(NS_, H_, GRID, LDS, CAP)                                                                              \
    do {                                                                                                                   \
        if (metric_ == M_SQ) LAUNCH2(M_SQ, NS_, H_, GRID, LDS, CAP);                                                  \
        else if (metric_ == M_COS) LAUNCH2(M_COS, NS_, H_, GRID, LDS, CAP);                                           \
        else if (metric_ == M_I8) LAUNCH2(M_I8, NS_, H_, GRID, LDS, CAP);                                             \
        else LAUNCH2(M_UCOS, NS_, H_, GRID, LDS, CAP);                                                                \
    } while (0)
#define LAUNCH(NS_, GRID, LDS, CAP)                                                                                   \
    do {                                                                                                                   \
        if (vis_tab) LAUNCH3(NS_, true, GRID, LDS, CAP);                                                              \
        else LAUNCH3(NS_, false, GRID, LDS, CAP);                                                                     \
    } while (0)
        switch (ns) {
        case 2: LAUNCH(2, nj, lds, cand_cap); break;
        case 4: LAUNCH(4, nj, lds, cand_cap); break;
        default: LAUNCH3(8, false, nj, lds, cand_cap); break; // (no hash-table form: visited_table() was told so)
        }
        HIP_OK(hipGetLastError());
#undef LAUNCH
#undef LAUNCH3
#undef LAUNCH2
#undef LAUNCH2L
        if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev1_, st));
        if (tail_.n > 0 && !upload_tail()) return false; // the rest of the query set, while the launch above is running
        if (compact) { // [ids | distances | flags] in one piece (nj == chunk: the device slab and the staging are laid out alike)
            HIP_OK(hipMemcpyAsync(h_ids, d_ids, 2 * b_res + sizeof(int) * (size_t)nj, hipMemcpyDeviceToHost, st));
            HIP_OK(hipMemcpyAsync(h_ev, d_ev, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            memcpy(out_ids + (size_t)off * k_out, h_ids, 4u * (size_t)nj * k_out);
        } else {
        // the ids are copied out to the caller's array while the distances are still crossing the link
        HIP_OK(hipMemcpyAsync(h_ids, d_ids, 4u * (size_t)nj * k_out, hipMemcpyDeviceToHost, st));
        HIP_OK(hipEventRecord((hipEvent_t)ev2_, st));
        HIP_OK(hipMemcpyAsync(h_d, d_d, 4u * (size_t)nj * k_out, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(h_flag, s_flag_, sizeof(int) * (size_t)nj, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(h_ev, s_evals_, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_OK(hipEventSynchronize((hipEvent_t)ev2_));
        memcpy(out_ids + (size_t)off * k_out, h_ids, 4u * (size_t)nj * k_out);
        HIP_OK(hipStreamSynchronize(st));
        }
        memcpy(out_d + (size_t)off * k_out, h_d, 4u * (size_t)nj * k_out);
        for (int i = 0; i < nj; ++i) {
            if (h_flag[i] == 2) { stats_.search_repeats++; if (!keep_repeat_flag) h_flag[i] = 0; }
            else if (h_flag[i] == 4) { stats_.tie_windows++; h_flag[i] = 0; } // informational: a group window closed cleanly
        }
        memcpy(out_flag + off, h_flag, sizeof(int) * (size_t)nj);
        const unsigned long long ev = *h_ev;
        if (vis_tab) stats_.visited_hash_launches++;
        stats_.search_launches++;
        stats_.search_evals += ev;
        if (timed) {
            float ms = 0.f;
            HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)ev0_, (hipEvent_t)ev1_));
            stats_.search_kernel_ms += ms;
            stats_.search_timed_launches++;
            stats_.search_timed_evals += ev;
        }
    }
    for (int i = 0; i < njobs; ++i) stats_.search_overflows += (uint64_t)(out_flag[i] == 1);
    return true;
}

bool Device::relink_batch(const int *affected, const int *layer, const int *removed, const int *step, int n, const int *cands, const int *cand_off,
                          const int *cand_cnt, int nsteps, int max_edges0, int *out_sel, int *out_cnt, int *out_flag, int sel_stride, bool heap_order)
{
    if (n <= 0) return true;
    if (!affected || !layer || !removed || !step || !cand_off || !cand_cnt || !out_sel || !out_cnt || !out_flag || nsteps < 1 || max_edges0 < 2 ||
        sel_stride < max_edges0) {
        set_dev_error("relink_batch: bad argument");
        return false;
    }
    if (g_n_ <= 0) { set_dev_error("relink_batch: no graph uploaded"); return false; }
    int total_c = 0, max_c = 0;
    for (int s = 0; s < nsteps; ++s) {
        if (cand_cnt[s] < 0 || cand_off[s] != total_c) { set_dev_error("relink_batch: candidate ranges must be packed in step order"); return false; }
        total_c += cand_cnt[s];
        max_c = std::max(max_c, cand_cnt[s]);
    }
    if (total_c > 0 && !cands) { set_dev_error("relink_batch: bad argument"); return false; }
    for (int i = 0; i < total_c; ++i)
        if (cands[i] < 0 || cands[i] >= g_n_ || cands[i] >= n_rows_hw_) { set_dev_error("relink_batch: candidate outside the graph / rows"); return false; }
    for (int i = 0; i < n; ++i)
        if (affected[i] < 0 || affected[i] >= g_n_ || affected[i] >= n_rows_hw_ || removed[i] < 0 || removed[i] >= g_n_ || layer[i] < 0 || layer[i] > 200 ||
            step[i] < 0 || step[i] >= nsteps) {
            set_dev_error("relink_batch: job outside the graph / rows");
            return false;
        }
    const int kcap = (g_stride0_ - 2) + max_c + 1, nb = (kcap + 7) & ~7;
    const size_t lds = search_lds_bytes(kcap, 0, pitch_, true, nb);
    if (lds > 64 * 1024) { set_dev_error("relink_batch: candidate count / dimension exceed the LDS budget"); return false; }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    if (!ensure_search_scratch(1, 1, 0, 16)) return false;
    // device staging: [jobs (4 n) | cand_off | cand_cnt | cands | sel | cnt | flag]
    const size_t o_off = 4u * (size_t)n, o_cnt = o_off + (size_t)nsteps, o_c = o_cnt + (size_t)nsteps, o_s = o_c + (size_t)std::max(total_c, 1),
                 o_n = o_s + (size_t)n * sel_stride, o_f = o_n + (size_t)n, total = o_f + (size_t)n;
    if (!grow_dev(&s_rl_, &s_rl_cap_, total + 4)) return false; // (+4: the int4 view of the jobs starts aligned, hipMalloc is)
    int *hs = static_cast<int *>(pinned_stage(sizeof(int) * total + 16));
    if (!hs) return false;
    for (int i = 0; i < n; ++i) { hs[4 * i] = affected[i]; hs[4 * i + 1] = layer[i]; hs[4 * i + 2] = removed[i]; hs[4 * i + 3] = step[i]; }
    memcpy(hs + o_off, cand_off, sizeof(int) * (size_t)nsteps);
    memcpy(hs + o_cnt, cand_cnt, sizeof(int) * (size_t)nsteps);
    if (total_c > 0) memcpy(hs + o_c, cands, sizeof(int) * (size_t)total_c);
    HIP_OK(hipMemcpyAsync(s_rl_, hs, sizeof(int) * o_s, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(s_evals_, 0, sizeof(unsigned long long), st));
#define LAUNCH_RL(M)                                                                                                                  \
    hipLaunchKernelGGL(graph_relink_kernel<M>, dim3(n), dim3(64), lds, st, d_rows_, d_row_sn_, pitch_, g_adj0_, g_stride0_, g_upper_,   \
                       g_pool_, g_strideU_, reinterpret_cast<const int4 *>(s_rl_), s_rl_ + o_c, s_rl_ + o_off, s_rl_ + o_cnt, max_edges0,  \
                       kcap, nb, s_rl_ + o_s, s_rl_ + o_n, s_rl_ + o_f, sel_stride, s_evals_, heap_order ? 1 : 0)
    if (metric_ == M_SQ) LAUNCH_RL(M_SQ);
    else if (metric_ == M_COS) LAUNCH_RL(M_COS);
    else if (metric_ == M_I8) LAUNCH_RL(M_I8);
    else LAUNCH_RL(M_UCOS);
#undef LAUNCH_RL
    HIP_OK(hipGetLastError());
    unsigned long long *h_ev = reinterpret_cast<unsigned long long *>(hs + ((total + 1) & ~(size_t)1));
    HIP_OK(hipMemcpyAsync(hs + o_s, s_rl_ + o_s, sizeof(int) * (total - o_s), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(h_ev, s_evals_, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    memcpy(out_sel, hs + o_s, sizeof(int) * (size_t)n * sel_stride);
    memcpy(out_cnt, hs + o_n, sizeof(int) * (size_t)n);
    memcpy(out_flag, hs + o_f, sizeof(int) * (size_t)n);
    stats_.search_evals += *h_ev;
    return true;
}

bool Device::patch_lists(const int *recs, int nrows, int row_stride)
{
    if (nrows <= 0) return true;
    if (!recs || row_stride < 4) { set_dev_error("patch_lists: bad argument"); return false; }
    if (g_n_ <= 0) { set_dev_error("patch_lists: no graph uploaded"); return false; }
    std::vector<int> tagged(recs, recs + (size_t)nrows * row_stride);
    for (int r = 0; r < nrows; ++r) { // a bad record must be an error return, never a GPU fault
        int *x = tagged.data() + (size_t)r * row_stride;
        const int cap = (x[1] == 0 ? g_stride0_ : g_strideU_) - 2;
        bool ok = x[0] >= 0 && x[0] < g_n_ && x[1] >= 0 && x[1] < 0x4000 && x[2] >= 0 && x[2] <= row_stride - 3 && x[2] <= cap &&
                  (x[1] == 0 || (hg_ ? hg_->level[(size_t)x[0]] >= x[1] : true));
        for (int i = 0; ok && i < x[2]; ++i) ok = x[3 + i] >= 0 && x[3 + i] < g_n_;
        if (!ok) { set_dev_error("patch_lists: record outside the graph"); return false; }
        x[1] |= 1 << 30; // not a heuristic's ordered output
    }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    const size_t total = (size_t)nrows * row_stride;
    if (!grow_dev(&s_rl_, &s_rl_cap_, total)) return false;
    int *hs = static_cast<int *>(pinned_stage(sizeof(int) * total));
    if (!hs) return false;
    memcpy(hs, tagged.data(), sizeof(int) * total);
    HIP_OK(hipMemcpyAsync(s_rl_, hs, sizeof(int) * total, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(graph_write_rows_kernel, dim3(nrows), dim3(64), 0, st, g_adj0_, g_stride0_, g_upper_, g_pool_, g_strideU_, s_rl_, row_stride,
                       g_tested0_, g_testedU_, g_stride0_ - 2);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(st)); // the staging buffers are reused
    return true;
}

// RangeQuery on the device (graph_range_kernel).  A launch packs its results into an arena sized from the last
// call's results per query (at least 1M entries); the jobs that did not fit run once more in an arena of exactly
// the size they asked for.  Jobs whose result set outgrew a wave's list (kSpillCap entries) run again, few at a
// time, with lists as long as the graph.
// Room for `entries` results in the pinned host buffer, the first `keep` of which survive a reallocation.
bool Device::range_host_room(size_t entries, size_t keep)
{
    if (entries <= h_range_cap_) return true;
    const size_t cap = std::max(entries + entries / 2, (size_t)1 << 20);
    SearchHit *p = nullptr;
    if (hipHostMalloc((void **)&p, sizeof(SearchHit) * cap, hipHostMallocDefault) != hipSuccess) { set_dev_error("range_batch: pinned allocation failed"); return false; }
    if (h_range_) {
        if (keep) memcpy(p, h_range_, sizeof(SearchHit) * keep);
        (void)hipHostFree(h_range_);
    }
    h_range_ = p;
    h_range_cap_ = cap;
    return true;
}

bool Device::range_batch(const SearchJob *jobs, int njobs, float range, RangeResults *res)
{
    res->off.assign((size_t)std::max(njobs, 0), 0ull);
    res->cnt.assign((size_t)std::max(njobs, 0), 0);
    res->flag.assign((size_t)std::max(njobs, 0), 0);
    res->entry.assign((size_t)std::max(njobs, 0), -1);
    res->state.assign((size_t)std::max(njobs, 0), kRangeHostSort);
    res->found = nullptr;
    res->found_n = 0;
    if (njobs <= 0) return true;
    if (!jobs) { set_dev_error("range_batch: bad argument"); return false; }
    if (g_n_ <= 0) { set_dev_error("range_batch: no graph uploaded"); return false; }
    if (!jobs_valid(jobs, njobs, g_n_, n_queries_, n_rows_hw_)) { set_dev_error("range_batch: job outside the uploaded graph / rows / queries"); return false; }
    for (int i = 0; i < njobs; ++i)
        if (jobs[i].qref < 0 || jobs[i].search_layer != 0) { set_dev_error("range_batch: jobs must name a resident query and layer 0"); return false; }
    const int nbcap_r = kRangeFan * nbcap(); // the kernel expands kRangeFan lists per step
    const size_t lds = search_lds_bytes(0, 0, pitch_, false, nbcap_r);
    if (lds > 64 * 1024) { set_dev_error("range_batch: dimension exceeds the LDS budget"); return false; }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    long long vis_words = ((g_n_ + 31) / 32 + 3) & ~3LL;
    size_t vis_bytes_per_job = sizeof(unsigned) * (size_t)vis_words;
    int *vis_tab = nullptr;
    int vis_tab_cap = 0;
    // 32 768 slots: 24 576 visited ids before a hand-back; a step inserts up to kRangeFan lists of 128 ids
    if (!visited_table(vis_bytes_per_job, 512, &vis_tab, &vis_tab_cap, 4 * kRangeFan * 128)) return false;
    if (vis_tab) { vis_words = 0; vis_bytes_per_job = 16; }
    const long long chunk = std::min<long long>(njobs, 1 << 20);
    if (!ensure_search_scratch(chunk, max_slots(), 0, vis_bytes_per_job)) return false; // also the per-wave result lists (s_spill_)
    if ((size_t)chunk > s_roff_cap_) {
        if (s_roff_) HIP_OK(hipFree(s_roff_));
        if (s_rentry_) HIP_OK(hipFree(s_rentry_));
        s_roff_ = nullptr; s_rentry_ = nullptr; s_roff_cap_ = 0;
        HIP_OK(hipMalloc(&s_roff_, sizeof(unsigned long long) * (size_t)chunk));
        HIP_OK(hipMalloc(&s_rentry_, sizeof(int) * (size_t)chunk));
        if (s_rstate_) HIP_OK(hipFree(s_rstate_));
        if (s_rtied_) HIP_OK(hipFree(s_rtied_));
        s_rstate_ = nullptr; s_rtied_ = nullptr;
        HIP_OK(hipMalloc(&s_rstate_, sizeof(int) * (size_t)chunk));
        HIP_OK(hipMalloc(&s_rtied_, sizeof(int) * ((size_t)chunk + 1)));
        s_roff_cap_ = (size_t)chunk;
    }
    if (!s_arena_used_) HIP_OK(hipMalloc(&s_arena_used_, sizeof(unsigned long long)));
    if (!s_rfin_ctr_) HIP_OK(hipMalloc(&s_rfin_ctr_, sizeof(int) * 2));
    constexpr size_t kArenaMax = (size_t)1 << 27; // 1 GB of results per launch; what does not fit then is handed back

    // One launch over the jobs listed in `todo` (at most `chunk`): results appended to res->found; jobs that found
    // the arena full are listed in `again` and *need = the entries they asked for; handed-back jobs in `handed`.
    auto launch = [&](const int *todo, int nj, ND *lists, int list_cap, int grid_cap, size_t arena_cap, std::vector<int> &again,
                      unsigned long long *need, std::vector<int> &handed) -> bool {
        if (!grow_dev(&s_arena_, &s_arena_cap_, arena_cap)) return false;
        // pinned layout: [evals, used (16 B) | jobs | offsets | counts | flags | entries]; then reused for the results
        const size_t b_jobs = sizeof(SearchJob) * (size_t)nj, b_off = 8u * (size_t)nj, b_i = 4u * (size_t)nj;
        char *hs = static_cast<char *>(pinned_stage(16 + b_jobs + b_off + 4 * b_i));
        if (!hs) return false;
        SearchJob *h_jobs = reinterpret_cast<SearchJob *>(hs + 16);
        for (int i = 0; i < nj; ++i) h_jobs[i] = jobs[todo[i]];
        HIP_OK(hipMemcpyAsync(s_jobs_, h_jobs, b_jobs, hipMemcpyHostToDevice, st));
        uj_len_ = 0;
        HIP_OK(hipMemsetAsync(s_jobctr_, 0, sizeof(int), st));
        HIP_OK(hipMemsetAsync(s_evals_, 0, sizeof(unsigned long long), st));
        HIP_OK(hipMemsetAsync(s_arena_used_, 0, sizeof(unsigned long long), st));
        const bool timed = profiling_;
        if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev0_, st));
#define LAUNCH_RANGE2(M, H_)                                                                                                          \
    do {                                                                                                                               \
        const int slots_ = std::min(std::min(max_slots(), grid_cap), resident_blocks(graph_range_kernel<M, H_>, lds, num_cu_));        \
        hipLaunchKernelGGL((graph_range_kernel<M, H_>), dim3(std::min<int>(nj, slots_)), dim3(64), lds, st, d_rows_, d_row_sn_,        \
                           d_queries_, d_q_sn_, pitch_, g_adj0_, g_stride0_, g_upper_, g_pool_, g_strideU_, s_jobs_, range,            \
                           lists, list_cap, s_visited_, vis_words, vis_tab, vis_tab_cap,                                               \
                           reinterpret_cast<ND *>(s_arena_), (unsigned long long)arena_cap, s_arena_used_, s_roff_, s_cnt_, s_flag_,   \
                           s_rentry_, s_evals_, nbcap_r, nj, s_jobctr_);                                                               \
    } while (0)
#define LAUNCH_RANGE(M) do { if (vis_tab) LAUNCH_RANGE2(M, true); else LAUNCH_RANGE2(M, false); } while (0)
        if (metric_ == M_SQ) LAUNCH_RANGE(M_SQ);
        else if (metric_ == M_COS) LAUNCH_RANGE(M_COS);
        else if (metric_ == M_I8) LAUNCH_RANGE(M_I8);
        else LAUNCH_RANGE(M_UCOS);
#undef LAUNCH_RANGE
#undef LAUNCH_RANGE2
        HIP_OK(hipGetLastError());
        // the ORDER, still on the device (dk_range_finish.h): every finished list ranked ascending in place; the lists that hold equal
        // distances replayed -- the reference's two heaps on the distances just found -- and ranked in heap-array order.  What these
        // hand back (lists beyond kRangeSortMax entries, -0 distances) the callers sort and replay on the host as before.
        const int finish = diag("range_finish", 2); // 0: order left to the host (as until round 5), 1: ranking only, 2: ranking and replays
        HIP_OK(hipMemsetAsync(s_rfin_ctr_, 0, sizeof(int) * 2, st));
        HIP_OK(hipMemsetAsync(s_rtied_, 0, sizeof(int), st));
        HIP_OK(hipMemsetAsync(s_rstate_, 0, sizeof(int) * (size_t)nj, st));
        if (finish >= 1) {
            hipLaunchKernelGGL(range_sort_kernel, dim3(std::min(nj, 8 * std::max(1, num_cu_))), dim3(64), 0, st, reinterpret_cast<ND *>(s_arena_), s_roff_, s_cnt_, s_flag_, nj,
                               s_rstate_, s_rtied_, s_rfin_ctr_);
            HIP_OK(hipGetLastError());
        }
        if (finish >= 2) {
            hipLaunchKernelGGL(range_replay_kernel, dim3(std::min(nj, 2 * std::max(1, num_cu_))), dim3(64), sizeof(RangeReplayLds) + 64, st, reinterpret_cast<ND *>(s_arena_), s_roff_,
                               s_cnt_, s_rentry_, g_adj0_, g_stride0_, g_n_, range, s_rstate_, s_rtied_, s_rfin_ctr_ + 1);
            HIP_OK(hipGetLastError());
        }
        if (timed) HIP_OK(hipEventRecord((hipEvent_t)ev1_, st));
        unsigned long long *h_hdr = reinterpret_cast<unsigned long long *>(hs);
        unsigned long long *h_off = reinterpret_cast<unsigned long long *>(hs + 16 + b_jobs);
        int *h_cnt = reinterpret_cast<int *>(hs + 16 + b_jobs + b_off);
        int *h_flag = h_cnt + nj, *h_entry = h_flag + nj, *h_state = h_entry + nj;
        HIP_OK(hipMemcpyAsync(h_state, s_rstate_, b_i, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(h_hdr, s_evals_, 8, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(h_hdr + 1, s_arena_used_, 8, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(h_off, s_roff_, b_off, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(h_cnt, s_cnt_, b_i, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(h_flag, s_flag_, b_i, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(h_entry, s_rentry_, b_i, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        const unsigned long long ev = h_hdr[0], used = h_hdr[1];
        // entries claimed beyond arena_cap belong to the jobs flagged 3 and were never written; the claims of the
        // finished jobs all lie below it, though not contiguously: the span they cover is copied back
        unsigned long long span = 0, finished = 0;
        const size_t base = res->found_n;
        for (int i = 0; i < nj; ++i) {
            const int j = todo[i];
            res->entry[(size_t)j] = h_entry[i];
            res->flag[(size_t)j] = 0;
            if (h_flag[i] == 0) {
                res->off[(size_t)j] = base + h_off[i];
                res->cnt[(size_t)j] = h_cnt[i];
                res->state[(size_t)j] = finish >= 1 ? h_state[i] : kRangeHostSort;
                if (h_cnt[i] >= 2) { if (res->state[(size_t)j] == kRangeFinal) stats_.range_device_ordered++; else stats_.range_host_ordered++; }
                finished += (unsigned long long)h_cnt[i];
                if (h_cnt[i] > 0) span = std::max(span, h_off[i] + (unsigned long long)h_cnt[i]);
            } else if (h_flag[i] == 3) again.push_back(j);
            else { res->flag[(size_t)j] = 1; handed.push_back(j); }
        }
        *need = used - finished;
        if (span > 0) { // one copy, straight into the context's pinned result buffer (no staging hop, nothing zero-filled first)
            if (!range_host_room(base + (size_t)span, base)) return false;
            HIP_OK(hipMemcpyAsync(h_range_ + base, s_arena_, sizeof(SearchHit) * (size_t)span, hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            res->found_n = base + (size_t)span;
        }
        res->found = h_range_;
        if (vis_tab) stats_.visited_hash_launches++;
        stats_.search_launches++;
        stats_.search_evals += ev;
        stats_.range_launches++;
        stats_.range_evals += ev;
        if (timed) {
            float ms = 0.f;
            HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)ev0_, (hipEvent_t)ev1_));
            stats_.search_kernel_ms += ms;
            stats_.search_timed_launches++;
            stats_.search_timed_evals += ev;
            stats_.range_kernel_ms += ms;
            stats_.range_timed_launches++;
            stats_.range_timed_evals += ev;
        }
        return true;
    };
    // `todo` through `launch`, with one more pass for the jobs that found the arena full
    auto run = [&](std::vector<int> todo, ND *lists, int list_cap, int grid_cap, size_t arena_cap, std::vector<int> &handed) -> bool {
        for (int pass = 0; pass < 2 && !todo.empty(); ++pass) {
            std::vector<int> again;
            unsigned long long need_total = 0;
            for (size_t off = 0; off < todo.size(); off += (size_t)chunk) {
                const int nj = (int)std::min<size_t>((size_t)chunk, todo.size() - off);
                unsigned long long need = 0;
                if (!launch(todo.data() + off, nj, lists, list_cap, grid_cap, arena_cap, again, &need, handed)) return false;
                need_total = std::max(need_total, need);
            }
            todo.swap(again);
            arena_cap = (size_t)std::min<unsigned long long>(std::max<unsigned long long>(need_total, 1ull << 20), kArenaMax);
        }
        for (int j : todo) { res->flag[(size_t)j] = 1; handed.push_back(j); } // more than kArenaMax results in one launch
        return true;
    };

    std::vector<int> all((size_t)njobs), handed, still;
    for (int i = 0; i < njobs; ++i) all[(size_t)i] = i;
    const size_t guess = (size_t)((range_hint_ * 1.25 + 16.0) * (double)std::min<long long>(chunk, njobs));
    if (!run(std::move(all), reinterpret_cast<ND *>(s_spill_), kSpillCap, max_slots(), std::min(std::max<size_t>((size_t)1 << 20, guess), kArenaMax), handed)) return false;
    // result sets beyond a wave's list: again, with lists as long as the graph (at most 1 GB of them at a time);
    // a visited table filling up (graphs above 4M nodes) is not helped by that and stays handed back
    if (!handed.empty() && !vis_tab && g_n_ > kSpillCap) {
        const size_t list_cap = (size_t)std::min<long long>(g_n_, 1 << 24);
        const int waves = (int)std::max<size_t>(1, std::min<size_t>(handed.size(), ((size_t)1 << 27) / list_cap));
        if (!grow_dev(&s_rlists_, &s_rlists_cap_, list_cap * (size_t)waves)) return false;
        if (!run(handed, reinterpret_cast<ND *>(s_rlists_), (int)list_cap, waves, std::min(std::max<size_t>((size_t)1 << 20, list_cap), kArenaMax), still)) return false;
        handed.swap(still);
    }
    stats_.range_handbacks += handed.size();
    unsigned long long total = 0;
    for (int i = 0; i < njobs; ++i) total += (unsigned long long)res->cnt[(size_t)i];
    range_hint_ = (double)total / (double)njobs;
    return true;
}

// float.CompareTo order on distances that are never NaN here (d <= range held)
static inline bool range_hit_less(const SearchHit &a, const SearchHit &b) { return a.dist < b.dist; }

bool Device::range_search(const float *queries, int nq, int entry_point, float range, int *out_counts, int *out_flags)
{
    abi_range_.clear();
    if (nq <= 0) return true;
    if (!out_counts || !out_flags) { set_dev_error("range_search: null argument"); return false; }
    if (!hg_ || g_n_ <= 0) { set_dev_error("range_search: no graph committed"); return false; }
    if (entry_point < 0 || entry_point >= hg_->n) { set_dev_error("range_search: bad argument"); return false; }
    if (!set_queries(queries, nq)) return false;
    std::vector<SearchJob> jobs((size_t)nq);
    const int top = hg_->level[(size_t)entry_point];
    for (int i = 0; i < nq; ++i) jobs[(size_t)i] = SearchJob{i, entry_point, top, 0, -1};
    RangeResults r;
    if (!range_batch(jobs.data(), nq, range, &r)) return false;
    for (int i = 0; i < nq; ++i) {
        out_counts[i] = 0;
        out_flags[i] = r.flag[(size_t)i];
        if (out_flags[i]) continue;
        SearchHit *b = r.found + r.off[(size_t)i], *e = b + r.cnt[(size_t)i];
        bool tie = r.state[(size_t)i] == kRangeTied; // (the device replays what it can: this is what it handed back)
        if (r.state[(size_t)i] == kRangeHostSort) {
            std::sort(b, e, range_hit_less);
            for (SearchHit *p = b; p + 1 < e; ++p) tie |= p[0].dist == p[1].dist; // also -0 next to +0
        }
        out_counts[i] = r.cnt[(size_t)i];
        if (tie) { // OrderBy keeps the heap array's order there (HNSWIndex.cs:155): replay the heaps on the committed graph
            std::vector<NodeDist> ordered;
            replay_range_heaps([&](int id) { return hg_->adj0.data() + (size_t)id * (size_t)hg_->stride0; }, 2 * hg_->M, r.entry[(size_t)i], range, b,
                               r.cnt[(size_t)i], ordered);
            for (const NodeDist &nd : ordered) abi_range_.push_back(SearchHit{nd.id, nd.dist});
            continue;
        }
        abi_range_.insert(abi_range_.end(), b, e);
    }
    return true;
}

bool Device::range_results(int *out_ids, float *out_d)
{
    if (abi_range_.empty()) return true;
    if (!out_ids || !out_d) { set_dev_error("range_results: null argument"); return false; }
    for (size_t i = 0; i < abi_range_.size(); ++i) { out_ids[i] = abi_range_[i].id; out_d[i] = abi_range_[i].dist; }
    return true;
}

// ---- C-ABI graph staging (layer by layer) -------------------------------------------------
bool Device::graph_begin(int n, int max_edges, const int *levels)
{
    if (n <= 0 || max_edges < 1 || !levels) { set_dev_error("graph_begin: bad argument"); return false; }
    delete hg_;
    hg_ = new HostGraphStage();
    HostGraphStage &g = *hg_;
    g.n = n; g.M = max_edges; g.stride0 = 2 * max_edges + 2; g.strideU = max_edges + 2;
    g.level.assign(levels, levels + n);
    g.adj0.assign((size_t)n * g.stride0, 0);
    g.upper.assign((size_t)n, -1);
    size_t pool_len = 0;
    for (int i = 0; i < n; ++i) {
        if (levels[i] < 0 || levels[i] > 200) { set_dev_error("graph_begin: level out of range"); return false; }
        g.top = std::max(g.top, levels[i]);
        if (levels[i] > 0) { g.upper[(size_t)i] = (int64_t)pool_len; pool_len += (size_t)levels[i] * g.strideU; }
    }
    g.pool.assign(pool_len, 0);
    return true;
}

bool Device::graph_set_layer(int layer, const int *counts, const int *edges, int stride)
{
    if (!hg_) { set_dev_error("graph_set_layer: call hnswdev_graph_begin first"); return false; }
    HostGraphStage &g = *hg_;
    if (layer < 0 || !counts || !edges || stride < 1) { set_dev_error("graph_set_layer: bad argument"); return false; }
    const int cap = layer == 0 ? 2 * g.M + 1 : g.M + 1;
    for (int i = 0; i < g.n; ++i) {
        if (g.level[(size_t)i] < layer) continue;
        const int c = counts[i];
        if (c < 0 || c > cap || c > stride) { set_dev_error("graph_set_layer: edge count exceeds MaxEdges(layer) + 1"); return false; }
        int *l = layer == 0 ? g.adj0.data() + (size_t)i * g.stride0 : g.pool.data() + g.upper[(size_t)i] + (size_t)(layer - 1) * g.strideU;
        l[0] = c;
        for (int j = 0; j < c; ++j) {
            const int e = edges[(size_t)i * stride + j];
            if (e < 0 || e >= g.n || g.level[(size_t)e] < layer) { set_dev_error("graph_set_layer: edge to a node outside the layer"); return false; }
            l[1 + j] = e;
        }
    }
    return true;
}

bool Device::graph_commit()
{
    if (!hg_) { set_dev_error("graph_commit: nothing staged"); return false; }
    HostGraphStage &g = *hg_;
    if (g.n > n_rows_hw_) { set_dev_error("graph_commit: graph has more nodes than uploaded rows"); return false; }
    return set_graph(g.adj0.data(), g.n, g.stride0, g.level.data(), g.upper.data(), g.pool.data(), (long long)g.pool.size(), g.strideU);
}

bool Device::knn_search(const float *queries, int nq, int entry_point, int k_beam, int k_out, int *out_ids, float *out_d, int *out_flag)
{
    if (nq <= 0) return true;
    if (!hg_ || g_n_ <= 0) { set_dev_error("knn_search: no graph committed"); return false; }
    if (entry_point < 0 || entry_point >= hg_->n || k_out < 1 || k_beam < k_out) { set_dev_error("knn_search: bad argument"); return false; }
    if (!set_queries(queries, nq)) return false;
    std::vector<SearchJob> jobs((size_t)nq);
    const int top = hg_->level[(size_t)entry_point];
    for (int i = 0; i < nq; ++i) jobs[(size_t)i] = SearchJob{i, entry_point, top, 0, -1};
    return search_batch(jobs.data(), nq, k_beam, k_out, out_ids, out_d, out_flag);
}

// ---- synchronous conveniences behind the C ABI ---------------------------------------
// Distance(int, TVector) for nq (query, candidate list) pairs.  Runs on the context's two step-
// buffer sets, ping-pong: while the GPU measures one set the host packs the next and unpacks the
// previous -- nothing is allocated per call once the sets exist.
bool Device::dist_query_batch(const float *queries, int nq, const int *offsets, const int *ids, float *out)
{
    if (nq <= 0) return true;
    if (!offsets || !out) { set_dev_error("dist_query_batch: null argument"); return false; }
    if (offsets[0] != 0) { set_dev_error("dist_query_batch: cand_offsets[0] must be 0"); return false; }
    for (int i = 0; i < nq; ++i)
        if (offsets[i + 1] < offsets[i]) { set_dev_error("dist_query_batch: cand_offsets must be non-decreasing"); return false; }
    const int total = offsets[nq];
    if (total > 0 && !ids) { set_dev_error("dist_query_batch: null cand_ids"); return false; }
    if (queries) { if (!set_queries(queries, nq)) return false; }
    else if (nq > n_queries_) { set_dev_error("dist_query_batch: queries == NULL needs a resident query set of at least nq rows (hnswdev_set_queries)"); return false; }
    const int stride = 64, NS = 8192;
    int *rec[2]; float *dist[2];
    StepBuffers **sets = abi_sb_ + 2; // private sets: a caller's hnswdev_step_buffers pointers stay valid
    for (int g = 0; g < 2; ++g)
        if (!step_buffers(2 + g, NS, stride, &rec[g], &dist[g], true)) return false;
    const int rec_stride = sets[0]->rec_stride;
    std::vector<std::pair<int, int>> where[2]; // (global offset, count) per slot of each set
    bool pend[2] = {false, false};
    auto collect = [&](int g) -> bool {
        if (!pend[g]) return true;
        pend[g] = false;
        if (!wait_step(sets[g])) return false;
        for (size_t sidx = 0; sidx < where[g].size(); ++sidx)
            memcpy(out + where[g][sidx].first, dist[g] + sidx * (size_t)stride, sizeof(float) * (size_t)where[g][sidx].second);
        return true;
    };
    int qi = 0, pos = 0, g = 0; // next (query, offset within its list) to schedule
    bool ok = true;
    while (ok && qi < nq) {
        ok = collect(g); // this set's previous step
        if (!ok) break;
        where[g].clear();
        uint64_t ev = 0;
        int used = 0;
        while (qi < nq && used < NS) {
            const int m = offsets[qi + 1] - offsets[qi] - pos;
            if (m <= 0) { ++qi; pos = 0; continue; }
            const int take = std::min(m, stride);
            const int at = offsets[qi] + pos;
            int *r = rec[g] + (size_t)used * rec_stride;
            r[0] = take;
            r[1] = qi;
            memcpy(r + 2, ids + at, sizeof(int) * (size_t)take);
            where[g].emplace_back(at, take);
            ev += (uint64_t)take;
            ++used;
            pos += take;
        }
        if (used == 0) break;
        ok = launch_step(sets[g], used, ev);
        pend[g] = ok;
        g ^= 1;
    }
    // drain in submission order; on failure still wait for what is in flight
    const bool a = collect(g), b = collect(g ^ 1);
    return ok && a && b;
}

bool Device::dist_pair_batch(const int *a, const int *b, int n, float *out)
{
    if (n <= 0) return true;
    if (!a || !b || !out) { set_dev_error("dist_pair_batch: null argument"); return false; }
    if (!bind()) return false;
    hipStream_t st = S(stream_);
    if (!d_guard_) { HIP_OK(hipMalloc(&d_guard_, sizeof(int))); HIP_OK(hipMemsetAsync(d_guard_, 0, sizeof(int), st)); }
    if (!grow_dev(&pair_dev_, &pair_dev_cap_, 3 * (size_t)n)) return false; // [a | b | out], grown on demand, kept
    int *hs = static_cast<int *>(pinned_stage(sizeof(int) * (3 * (size_t)n + 1)));
    if (!hs) return false;
    memcpy(hs, a, sizeof(int) * (size_t)n);
    memcpy(hs + n, b, sizeof(int) * (size_t)n);
    int *da = pair_dev_, *db = pair_dev_ + n;
    float *dout = reinterpret_cast<float *>(pair_dev_ + 2 * (size_t)n);
    HIP_OK(hipMemcpyAsync(da, hs, sizeof(int) * 2 * (size_t)n, hipMemcpyHostToDevice, st));
    dim3 grid((unsigned)(((long long)n * 8 + 255) / 256)), block(256);
#define LAUNCH(M) hipLaunchKernelGGL(pair_distance_kernel<M>, grid, block, 0, st, d_rows_, d_row_sn_, pitch_, da, db, dout, n, n_rows_hw_, d_guard_)
    if (metric_ == M_SQ) LAUNCH(M_SQ);
    else if (metric_ == M_COS) LAUNCH(M_COS);
    else if (metric_ == M_I8) LAUNCH(M_I8);
    else LAUNCH(M_UCOS);
#undef LAUNCH
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(hs + 2 * (size_t)n, dout, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(hs + 3 * (size_t)n, d_guard_, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    memcpy(out, hs + 2 * (size_t)n, sizeof(float) * (size_t)n);
    stats_.launches++;
    stats_.evals += (uint64_t)n;
    if (hs[3 * (size_t)n] != 0) {
        HIP_OK(hipMemsetAsync(d_guard_, 0, sizeof(int), st));
        set_dev_error("dist_pair_batch: id outside uploaded rows (those distances are NaN)");
        return false;
    }
    return true;
}

// test hook (exported through the C ABI as hnswdev_test_sqrt_rn)
bool device_sqrt_rn(int device, const double *in, double *out, int n)
{
    HIP_OK(hipSetDevice(device));
    double *di = nullptr, *dout = nullptr;
    HIP_OK(hipMalloc(&di, sizeof(double) * (size_t)n));
    HIP_OK(hipMalloc(&dout, sizeof(double) * (size_t)n));
    HIP_OK(hipMemcpy(di, in, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sqrt_rn_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, di, dout, n);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpy(out, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    (void)hipFree(di); (void)hipFree(dout);
    return true;
}

} // namespace hnsw

// ------------------------------------------------------------------------------------
// C ABI (B): hnswdev_*
// ------------------------------------------------------------------------------------
using hnsw::Device;

extern "C" {

#define DEV_API __attribute__((visibility("default")))

DEV_API int hnswdev_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { hnsw::set_dev_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); return -1; }
    return n;
}

DEV_API int hnswdev_create(int device, int dim, int metric, long long capacity, void **ctx)
{
    if (!ctx) { hnsw::set_dev_error("hnswdev_create: null ctx"); return -1; }
    *ctx = nullptr;
    Device *d = Device::create(device, dim, metric, capacity);
    if (!d) return -1;
    *ctx = d;
    return 0;
}
DEV_API int hnswdev_destroy(void *ctx)
{
    delete (Device *)ctx;
    return 0;
}
// every call on a context: serialised, and its errors are filed under that context
#define CTX_OR_FAIL()                                                     \
    Device *d = (Device *)ctx;                                            \
    if (!d) { hnsw::set_dev_error("null hnswdev context"); return -1; }   \
    std::lock_guard<std::mutex> ctx_lock_(d->mutex());                    \
    hnsw::ErrorScope ctx_scope_(d)

DEV_API int hnswdev_reserve(void *ctx, long long capacity) { CTX_OR_FAIL(); return d->reserve(capacity) ? 0 : -1; }
DEV_API int hnswdev_upload_rows(void *ctx, int first_id, int n, const float *rows) { CTX_OR_FAIL(); return d->upload_rows(first_id, n, rows) ? 0 : -1; }
DEV_API int hnswdev_download_rows(void *ctx, int first_id, int n, float *rows) { CTX_OR_FAIL(); return d->download_rows(first_id, n, rows) ? 0 : -1; }
DEV_API int hnswdev_dist_query_batch(void *ctx, const float *queries, int nq, const int *cand_offsets, const int *cand_ids, float *out)
{
    CTX_OR_FAIL();
    return d->dist_query_batch(queries, nq, cand_offsets, cand_ids, out) ? 0 : -1;
}
DEV_API int hnswdev_dist_pair_batch(void *ctx, const int *a_ids, const int *b_ids, int n, float *out)
{
    CTX_OR_FAIL();
    return d->dist_pair_batch(a_ids, b_ids, n, out) ? 0 : -1;
}
DEV_API int hnswdev_set_queries(void *ctx, const float *queries, int nq) { CTX_OR_FAIL(); return d->set_queries(queries, nq) ? 0 : -1; }
DEV_API int hnswdev_step_buffers(void *ctx, int set, int nslots, int stride, int **rec, float **dist) { CTX_OR_FAIL(); return d->step_buffers(set, nslots, stride, rec, dist) ? 0 : -1; }
DEV_API int hnswdev_step_submit(void *ctx, int set, int nslots_used) { CTX_OR_FAIL(); return d->step_submit(set, nslots_used) ? 0 : -1; }
DEV_API int hnswdev_step_wait(void *ctx, int set) { CTX_OR_FAIL(); return d->step_wait(set) ? 0 : -1; }
DEV_API int hnswdev_graph_begin(void *ctx, int n, int max_edges, const int *levels) { CTX_OR_FAIL(); return d->graph_begin(n, max_edges, levels) ? 0 : -1; }
DEV_API int hnswdev_graph_set_layer(void *ctx, int layer, const int *counts, const int *edges, int stride) { CTX_OR_FAIL(); return d->graph_set_layer(layer, counts, edges, stride) ? 0 : -1; }
DEV_API int hnswdev_graph_commit(void *ctx) { CTX_OR_FAIL(); return d->graph_commit() ? 0 : -1; }
DEV_API int hnswdev_knn_search(void *ctx, const float *queries, int nq, int entry_point, int k_beam, int k_out, int *out_ids, float *out_dists, int *out_flags)
{
    CTX_OR_FAIL();
    return d->knn_search(queries, nq, entry_point, k_beam, k_out, out_ids, out_dists, out_flags) ? 0 : -1;
}
DEV_API int hnswdev_range_search(void *ctx, const float *queries, int nq, int entry_point, float range, int *out_counts, int *out_flags)
{
    CTX_OR_FAIL();
    return d->range_search(queries, nq, entry_point, range, out_counts, out_flags) ? 0 : -1;
}
DEV_API int hnswdev_range_results(void *ctx, int *out_ids, float *out_dists) { CTX_OR_FAIL(); return d->range_results(out_ids, out_dists) ? 0 : -1; }
DEV_API int hnswdev_sync(void *ctx) { CTX_OR_FAIL(); return d->sync() ? 0 : -1; }
DEV_API int hnswdev_set_profiling(void *ctx, int enabled) { CTX_OR_FAIL(); d->set_profiling(enabled != 0); return 0; }
DEV_API int hnswdev_get_stats(void *ctx, hnswdev_stats *out)
{
    CTX_OR_FAIL();
    if (!out) return -1;
    d->get_stats(out);
    return 0;
}
DEV_API int hnswdev_reset_stats(void *ctx) { CTX_OR_FAIL(); d->reset_stats(); return 0; }
static int copy_error(const std::string &s, char *buf, int buf_len)
{
    if (buf && buf_len > 0) {
        int w = std::min<int>((int)s.size(), buf_len - 1);
        memcpy(buf, s.data(), (size_t)w);
        buf[w] = 0;
    }
    return (int)s.size();
}
DEV_API int hnswdev_last_error(char *buf, int buf_len) { return copy_error(hnsw::get_dev_error(), buf, buf_len); }
DEV_API int hnswdev_ctx_last_error(void *ctx, char *buf, int buf_len)
{
    Device *d = (Device *)ctx;
    if (!d) return copy_error("null hnswdev context", buf, buf_len);
    return copy_error(d->error(), buf, buf_len);
}
// test hook: correctly rounded device double sqrt (cosine epilogue), checked against the host's
DEV_API int hnswdev_test_sqrt_rn(int device, const double *in, double *out, int n)
{
    return hnsw::device_sqrt_rn(device, in, out, n) ? 0 : -1;
}

} // extern "C"
