// mdrp_capi.hip — C ABI (include/mdrp.h) and host orchestration of the chunked, phase-split LO-RANSAC.
// Everything numerical runs in the gfx950 kernels of mdrp_kernels.h; the host only sizes buffers, groups pairs by
// correspondence count (one sample table per distinct N) and reads back one 16-byte progress record per chunk.
#include "../../include/mdrp.h"
#include "mdrp_kernels.h"
#include "mdrp_classic.h"
// MDRP_SPLIT_TU (the default build): the k_final family and the baselines' kernels are instantiated in mdrp_tu.hip, compiled in parallel with
// this file; a single-unit build (experiment builds with -D switches: mdrp_amd/build.py single=True) instantiates them here, implicitly.
#ifdef MDRP_SPLIT_TU
#define MDRP_INST extern
#include "mdrp_instances.h"
namespace mdrp {
MDRP_INSTANCES_FINAL_64
MDRP_INSTANCES_FINAL_256
MDRP_INSTANCES_CLASSIC
}
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace mdrp;

static_assert(sizeof(mdrp_model) == sizeof(Model), "model layout");
static_assert(sizeof(mdrp_camera) == sizeof(CamDev), "camera layout");
static_assert(sizeof(mdrp_result) == sizeof(ResultDev), "result layout");

namespace {

thread_local std::string g_err;

#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            char buf_[512];                                                                                \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            g_err = buf_;                                                                                  \
            return MDRP_ERR_HIP;                                                                           \
        }                                                                                                  \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return MDRP_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        // grow with headroom so alternating sizes do not thrash
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { g_err = std::string("hipMalloc failed: ") + hipGetErrorString(e); p = nullptr; return MDRP_ERR_HIP; }
        cap = want;
        return MDRP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// read back once per super-chunk: device counters at int32 index 2.. of the `counters` buffer
struct Progress {
    int32_t n_active; int32_t lo_overflow; // lo_overflow: a chunk found more triggers than the LM engine's problem table holds per pass
    unsigned long long max_needed;
    unsigned long long evals;       // (model x correspondence) evaluations the CPU loop would do: sum over pairs of models * n
    unsigned long long evals_mfma;  // evaluations executed by k_count on the matrix cores (padded to 16 x 16 tiles)
    unsigned long long evals_sweep; // evaluations handed to the fp64 sweep (survivors * n)
    unsigned long long evals_bound; // evaluations executed by k_bound in fp32 (k_count's survivors * n)
    int32_t wish_sum, wish_pairs;   // final refinements: sum over the pairs of first_chunk_wish(inlier ratio of the result), number of pairs (RunParams::inl_stat)
};
static_assert(sizeof(Progress) == 14 * sizeof(int32_t), "Progress ends where the LO queue heads begin");
constexpr int CNT_INL_STAT = 14; // int32 index of Progress::wish_sum in the `counters` buffer
constexpr int CNT_LO_HEAD = 16; // int32 index of the LO queue heads (one per chunk) in the `counters` buffer
constexpr int CNT_XCD_HEAD = 32; // int32 index of the per-XCD LO queue heads: [chunk][8] at LO_XCD_STRIDE ints (lo_take, mdrp_kernels.h)
constexpr size_t COUNTERS_BYTES = sizeof(int32_t) * (CNT_XCD_HEAD + 8 * LO_XCD_STRIDE);
constexpr size_t LM_STATS_BYTES = 6 * sizeof(unsigned long long); // mdrp_handle::lm_stats

// Every entry point runs on the handle's device and puts the caller's current device back on return (the caller is
// usually torch, which tracks its own current device).
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

} // namespace

struct mdrp_handle {
    std::mutex mu; // one call at a time per handle (scratch, events and counters are per handle); different handles run concurrently
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipStream_t aux_stream = nullptr;  // the second chunk's sampler + solver run here, beside the first chunk's sweep
    hipStream_t aux_stream2 = nullptr; // the sample tables and the super-chunk's LO launch run here
    hipStream_t copy_stream = nullptr; // host-buffer calls: the H2D slices of the correspondences
    hipEvent_t ev_copied = nullptr, ev_prepped = nullptr;
    DevBuf fuse;                       // fused tail: control words (64 B) | done_cnt[batch] | fin_done[batch] | ready[batch]
    static constexpr int NC_MAX = 8; // chunks of a super-chunk
    hipEvent_t ev_lo = nullptr, ev_tables = nullptr, ev_sampled[2] = {}, ev_solved[NC_MAX] = {}, ev_scanned[NC_MAX] = {};
    int num_cu = 256;
    // persistent device buffers
    DevBuf pts, dep, st, samples;
    DevBuf params;                     // per-call parameters in ONE upload: table states | cameras | table sizes | table of pair | n per pair
    unsigned char *params_host = nullptr; // pinned staging of the same
    size_t params_host_cap = 0;
    DevBuf models, slot_score, slot_inl, tags, model_count, triggers, work_pair, counters, results, mask, plan;
    DevBuf tags_s, tags2_s; // survivor lists ordered by candidate density (k_sort_tags)
    DevBuf tags2, model_count2, samples2; // odd chunks of a super-chunk (chunk c + 1 is solved beside the sweep of chunk c)
    DevBuf tags_v, surv_count; // survivors of k_count (unsorted, with density keys) and their number per pair
    DevBuf cand_stat; // [2 batch] u64: candidates | evaluations of the run's first chunk per pair (k_count's split point)
    DevBuf und_count; // k_count's two phases: hypotheses phase A left undecided, per pair (stride 2); the list itself and the partial counts alias the sorted tag lists
    DevBuf rfrag;              // MFMA A fragments of the correspondences (k_prep): [pair][ceil(n_max/16)][64] x 16 B
    DevBuf cplan;              // work plan of k_count / k_bound
    DevBuf surv2_count;        // survivors of k_bound per pair
    DevBuf lo_mask;            // 5-point LO: inlier subset of the refined model, one row per LO workgroup
    DevBuf red5;               // 5-point solver: the Reduce5 blocks between its two kernels, [pair][ceil(chunk / 64)][76][64] doubles
    DevBuf lm_stats;                  // six u64: correspondences evaluated by the LM cost / accumulate sweeps of the LO kernel | of the final kernel |
                                      // fused tail: gate time-outs | final-refinement wait time-outs
    unsigned long long *lm_stats_host = nullptr; // pinned copy, valid after finish_timing
    int64_t fuse_gate_timeouts = 0, fuse_wait_timeouts = 0; // of the last call
    int64_t first_chunk = 0;                                // of the last call (mdrp_stats::first_chunk)
    double seen_wish[3] = {-1.0, -1.0, -1.0};               // per monodepth estimator: mean first_chunk_wish over the results of its last call that measured it
    int32_t *wish_host = nullptr; hipEvent_t ev_wish = nullptr; // unfused runs: the two sums are copied behind the final refinements and read by the next call
    int wish_kind = -1;                                     // ... of this estimator, once ev_wish has completed (-1: nothing pending)
    bool fuse_disabled = false;       // a bounded wait of the fused tail expired on this handle: streams do not overlap here, run unfused ...
    int fuse_retry_in = 0;            // ... for this many API calls, then try the fused tail again (a busy moment on a shared GPU is not a profiler)
    int fuse_backoff = 64;            // ... doubled after every consecutive expired wait (capped), reset by a call whose fused tail ran through
    DevBuf in_x1, in_x2, in_d1, in_d2; // staging when the caller passes host memory
    DevBuf unit_a, unit_b, unit_c, unit_d, unit_e, unit_f;
    Progress *progress_host = nullptr; // pinned
    // sweep timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<int> ev_what; // 0 = k_score (fp64 sweep), 1 = k_count (MFMA), 2 = LO, 3 = final refinement, 4 = k_bound, 5 = minimal solver
    size_t ev_used = 0;
    double sweep_ms = 0.0, count_ms = 0.0;
    double kind_ms[6] = {0, 0, 0, 0, 0, 0};
    int64_t kind_launches[6] = {0, 0, 0, 0, 0, 0};
    int64_t lm_cost_evals = 0, lm_accum_evals = 0, fin_cost_evals = 0, fin_accum_evals = 0;
    int64_t count_launches = 0;
    int64_t sweep_launches = 0, sweep_evals = 0, mfma_evals = 0, fp64_evals = 0, bound_evals = 0;
    int last_batch = 0;
};

namespace {

// one LM instantiation per kernel: dispatch (kind, estimate_shift) on the host
#define MDRP_LM_DISPATCH_T(KERNEL, T, kind, shift, grid, smem, stream, ...)                                          \
    do {                                                                                                             \
        if ((kind) == 0 && (shift)) hipLaunchKernelGGL((KERNEL<0, true, T>), grid, dim3(T), smem, stream, __VA_ARGS__);  \
        else if ((kind) == 0) hipLaunchKernelGGL((KERNEL<0, false, T>), grid, dim3(T), smem, stream, __VA_ARGS__);       \
        else if ((kind) == 1) hipLaunchKernelGGL((KERNEL<1, false, T>), grid, dim3(T), smem, stream, __VA_ARGS__);       \
        else hipLaunchKernelGGL((KERNEL<2, false, T>), grid, dim3(T), smem, stream, __VA_ARGS__);                        \
    } while (0)
// threads per LM problem: one wavefront (64) when problems outnumber SIMDs, a 256-thread workgroup otherwise
#define MDRP_LM_DISPATCH(KERNEL, threads, kind, shift, grid, smem, stream, ...)                                      \
    do {                                                                                                             \
        if ((threads) == 64) MDRP_LM_DISPATCH_T(KERNEL, 64, kind, shift, grid, smem, stream, __VA_ARGS__);           \
        else MDRP_LM_DISPATCH_T(KERNEL, 256, kind, shift, grid, smem, stream, __VA_ARGS__);                          \
    } while (0)

// k_final: the user's loss type of the inlier-only refinement is a template parameter as well (rp.final_loss): one instantiation per loss type of
// BundleOptions (an unknown type is the TRIVIAL loss, as in loss_value)
#define MDRP_FINAL_DISPATCH_L(T, FL, kind, shift, grid, smem, stream, ...)                                                  \
    do {                                                                                                                   \
        if ((kind) == 0 && (shift)) hipLaunchKernelGGL((k_final<0, true, T, FL>), grid, dim3(T), smem, stream, __VA_ARGS__);  \
        else if ((kind) == 0) hipLaunchKernelGGL((k_final<0, false, T, FL>), grid, dim3(T), smem, stream, __VA_ARGS__);       \
        else if ((kind) == 1) hipLaunchKernelGGL((k_final<1, false, T, FL>), grid, dim3(T), smem, stream, __VA_ARGS__);       \
        else hipLaunchKernelGGL((k_final<2, false, T, FL>), grid, dim3(T), smem, stream, __VA_ARGS__);                        \
    } while (0)
#define MDRP_FINAL_DISPATCH_T(T, floss, kind, shift, grid, smem, stream, ...)                                               \
    do {                                                                                                                   \
        switch (floss) {                                                                                                   \
        case 1: MDRP_FINAL_DISPATCH_L(T, 1, kind, shift, grid, smem, stream, __VA_ARGS__); break;                          \
        case 2: MDRP_FINAL_DISPATCH_L(T, 2, kind, shift, grid, smem, stream, __VA_ARGS__); break;                          \
        case 3: MDRP_FINAL_DISPATCH_L(T, 3, kind, shift, grid, smem, stream, __VA_ARGS__); break;                          \
        case 4: MDRP_FINAL_DISPATCH_L(T, 4, kind, shift, grid, smem, stream, __VA_ARGS__); break;                          \
        case 5: MDRP_FINAL_DISPATCH_L(T, 5, kind, shift, grid, smem, stream, __VA_ARGS__); break;                          \
        default: MDRP_FINAL_DISPATCH_L(T, 0, kind, shift, grid, smem, stream, __VA_ARGS__); break; /* TRIVIAL, like loss_value's default */ \
        }                                                                                                                  \
    } while (0)
#define MDRP_FINAL_DISPATCH(threads, floss, kind, shift, grid, smem, stream, ...)                                           \
    do {                                                                                                                   \
        if ((threads) == 64) MDRP_FINAL_DISPATCH_T(64, floss, kind, shift, grid, smem, stream, __VA_ARGS__);               \
        else MDRP_FINAL_DISPATCH_T(256, floss, kind, shift, grid, smem, stream, __VA_ARGS__);                              \
    } while (0)

// scoring sweeps: pose models with cheirality (calibrated monodepth, 5-point), F = diag(1,1,f2) E diag(1,1,f1) (focal estimators),
// or a raw fundamental matrix in the model's first nine doubles (7-point)
#define MDRP_SWEEP_DISPATCH(KERNEL, kind, grid, block, smem, stream, ...)                                                       \
    do {                                                                                                                        \
        if ((kind) == MDRP_CALIB || (kind) == MDRP_RELPOSE_5PT) hipLaunchKernelGGL((KERNEL<true, false>), grid, block, smem, stream, __VA_ARGS__); \
        else if ((kind) == MDRP_FUNDAMENTAL_7PT) hipLaunchKernelGGL((KERNEL<false, true>), grid, block, smem, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<false, false>), grid, block, smem, stream, __VA_ARGS__);                                 \
    } while (0)
// classic LM kernels: (kind, threads per problem)
#define MDRP_CLASSIC_LM_DISPATCH(KERNEL, threads, kind, grid, smem, stream, ...)                                                 \
    do {                                                                                                                        \
        if ((kind) == MDRP_RELPOSE_5PT && (threads) == 64) hipLaunchKernelGGL((KERNEL<CLASSIC_RELPOSE, 64>), grid, dim3(64), smem, stream, __VA_ARGS__); \
        else if ((kind) == MDRP_RELPOSE_5PT) hipLaunchKernelGGL((KERNEL<CLASSIC_RELPOSE, 256>), grid, dim3(256), smem, stream, __VA_ARGS__); \
        else if ((kind) == MDRP_SHARED_6PT && (threads) == 64) hipLaunchKernelGGL((KERNEL<CLASSIC_SHARED, 64>), grid, dim3(64), smem, stream, __VA_ARGS__); \
        else if ((kind) == MDRP_SHARED_6PT) hipLaunchKernelGGL((KERNEL<CLASSIC_SHARED, 256>), grid, dim3(256), smem, stream, __VA_ARGS__); \
        else if ((threads) == 64) hipLaunchKernelGGL((KERNEL<CLASSIC_FUND, 64>), grid, dim3(64), smem, stream, __VA_ARGS__);      \
        else hipLaunchKernelGGL((KERNEL<CLASSIC_FUND, 256>), grid, dim3(256), smem, stream, __VA_ARGS__);                        \
    } while (0)

int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

// LDS work lists: 2 buffers of u16 indices, stride = n_max rounded to 64 (0 disables the lists)
int lm_list_stride(int n_max) { return n_max <= LM_LIST_MAX_N ? ((n_max + 63) / 64) * 64 : 0; }
size_t lm_list_bytes(int n_max) { return (size_t)2 * lm_list_stride(n_max) * sizeof(uint16_t); }
// k_final: a third list (the indices a record mask lets through, lm_mask_index) while the three stay within 32 KiB of dynamic LDS
int lm_mask_index_on(int n_max) { return lm_list_stride(n_max) > 0 && (size_t)3 * lm_list_stride(n_max) * sizeof(uint16_t) <= 32768 ? 1 : 0; }
size_t lm_final_list_bytes(int n_max) { return (size_t)(2 + lm_mask_index_on(n_max)) * lm_list_stride(n_max) * sizeof(uint16_t); }

int get_events(mdrp_handle *h, hipEvent_t *a, hipEvent_t *b, int what = 0) {
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t x, y;
        HIPCHK(hipEventCreate(&x));
        HIPCHK(hipEventCreate(&y));
        h->ev_pool.emplace_back(x, y);
        h->ev_what.push_back(0);
    }
    h->ev_what[h->ev_used] = what;
    *a = h->ev_pool[h->ev_used].first;
    *b = h->ev_pool[h->ev_used].second;
    h->ev_used++;
    return MDRP_OK;
}

int solver_for(int kind, int est_shift) {
    if (kind == MDRP_CALIB) return est_shift ? SOLVER_SHIFT : SOLVER_P3P;
    return kind == MDRP_SHARED_FOCAL ? SOLVER_SHARED : SOLVER_VARYING;
}

// one pass = a contiguous range of pairs that fits the scratch budget
// `host` (or null): the caller's HOST buffers of this pass; x1 ... d2 are then the handle's device staging buffers, still to be filled
struct HostSrc { const double *x1, *x2, *d1, *d2; };
constexpr int HOST_SLICE_PAIRS = 256; // pairs per H2D slice of a host-buffer call (24.6 MB at N = 2000: ~0.5 ms of PCIe per slice)

int run_pass(mdrp_handle *h, int kind, const double *x1, const double *x2, const double *d1, const double *d2, int batch,
             int n_max, const int32_t *n_host, const mdrp_camera *cam1, const mdrp_camera *cam2, const mdrp_ransac_opt *ro,
             const mdrp_bundle_opt *bo, int chunk_cap, uint8_t *mask_dev, ResultDev *results_dev, const HostSrc *host, int batch_call) {
    hipStream_t s = h->stream;
    const int est_shift = (kind == MDRP_CALIB && ro->monodepth_estimate_shift) ? 1 : 0;
    const bool classic = kind >= MDRP_RELPOSE_5PT;              // non-monodepth baselines (mdrp_classic.h)
    const int mps = kind == MDRP_RELPOSE_5PT ? 12 : (kind == MDRP_SHARED_6PT ? 16 : 4); // model slots per sample
    const int ssz = kind == MDRP_RELPOSE_5PT ? 5 : (kind == MDRP_SHARED_6PT ? 6 : (kind == MDRP_FUNDAMENTAL_7PT ? 7 : 3)); // sample size

    // ---- group pairs by correspondence count: one sample table per distinct N
    std::vector<int32_t> table_of(batch), tab_n;
    {
        std::map<int32_t, int32_t> ids;
        for (int i = 0; i < batch; ++i) {
            auto it = ids.find(n_host[i]);
            if (it == ids.end()) { it = ids.emplace(n_host[i], (int32_t)tab_n.size()).first; tab_n.push_back(n_host[i]); }
            table_of[i] = it->second;
        }
    }
    const int n_tables = (int)tab_n.size();
    std::vector<uint64_t> tab_state(n_tables, ro->seed);

    const size_t slots = (size_t)batch * chunk_cap * mps;
    int rc;
    if ((rc = h->pts.ensure(sizeof(double) * PT_STRIDE * batch * n_max))) return rc;
    if (!classic && (rc = h->dep.ensure(sizeof(double) * 2 * batch * n_max))) return rc;
    if ((rc = h->st.ensure(sizeof(PairState) * batch))) return rc;
    if ((rc = h->samples.ensure(sizeof(uint32_t) * ssz * (size_t)n_tables * chunk_cap))) return rc;
    // the small per-call inputs travel as one block (six pageable copies cost ~60 us of an idle GPU in front of k_prep)
    const size_t off_state = 0, off_cam1 = off_state + sizeof(uint64_t) * (size_t)n_tables, off_cam2 = off_cam1 + sizeof(CamDev) * (size_t)batch,
                 off_tn = off_cam2 + sizeof(CamDev) * (size_t)batch, off_tof = off_tn + sizeof(int32_t) * (size_t)n_tables,
                 off_nper = off_tof + sizeof(int32_t) * (size_t)batch, params_bytes = off_nper + sizeof(int32_t) * (size_t)batch;
    if ((rc = h->params.ensure(params_bytes))) return rc;
    if (h->params_host_cap < params_bytes) {
        if (h->params_host) (void)hipHostFree(h->params_host);
        h->params_host = nullptr; h->params_host_cap = 0;
        HIPCHK(hipHostMalloc((void **)&h->params_host, params_bytes + params_bytes / 2, hipHostMallocDefault));
        h->params_host_cap = params_bytes + params_bytes / 2;
    }
    unsigned char *pd = h->params.as<unsigned char>();
    uint64_t *d_table_state = reinterpret_cast<uint64_t *>(pd + off_state);
    CamDev *d_cams1 = reinterpret_cast<CamDev *>(pd + off_cam1), *d_cams2 = reinterpret_cast<CamDev *>(pd + off_cam2);
    int32_t *d_table_n = reinterpret_cast<int32_t *>(pd + off_tn), *d_table_of = reinterpret_cast<int32_t *>(pd + off_tof),
            *d_nper = reinterpret_cast<int32_t *>(pd + off_nper);
    if ((rc = h->models.ensure(sizeof(Model) * slots))) return rc;
    if ((rc = h->slot_score.ensure(sizeof(double) * slots))) return rc;
    if ((rc = h->slot_inl.ensure(sizeof(int32_t) * slots))) return rc;
    if ((rc = h->tags.ensure(sizeof(uint32_t) * slots))) return rc;
    if ((rc = h->model_count.ensure(sizeof(int32_t) * 2 * batch))) return rc;
    if ((rc = h->model_count2.ensure(sizeof(int32_t) * 2 * batch))) return rc;
    if ((rc = h->tags2.ensure(sizeof(uint32_t) * slots))) return rc;
    if ((rc = h->tags_s.ensure(sizeof(uint32_t) * slots))) return rc;
    if ((rc = h->tags2_s.ensure(sizeof(uint32_t) * slots))) return rc;
    if ((rc = h->samples2.ensure(sizeof(uint32_t) * ssz * (size_t)n_tables * chunk_cap))) return rc;
    if ((rc = h->tags_v.ensure(sizeof(uint32_t) * slots))) return rc;
    if ((rc = h->surv_count.ensure(sizeof(int32_t) * batch))) return rc;
    if ((rc = h->cand_stat.ensure(sizeof(unsigned long long) * 2 * batch))) return rc;
    if ((rc = h->und_count.ensure(sizeof(int32_t) * 2 * batch))) return rc;
    if ((rc = h->cplan.ensure(sizeof(int32_t) * ((size_t)batch + 1)))) return rc;
    if ((rc = h->surv2_count.ensure(sizeof(int32_t) * batch))) return rc;
    const size_t groups_max = ((size_t)n_max + 15) / 16;
    if ((rc = h->rfrag.ensure(std::max<size_t>(1024, (size_t)batch * groups_max * 1024)))) return rc;
    const int trig_cap = chunk_cap;
    if ((rc = h->triggers.ensure(sizeof(Trigger) * (size_t)batch * trig_cap))) return rc;
    if ((rc = h->work_pair.ensure(sizeof(int32_t) * (3 * (size_t)batch + 2)))) return rc; // LO plan of a super-chunk: prefix | begin | end | total
    if ((rc = h->counters.ensure(COUNTERS_BYTES))) return rc;
    if ((rc = h->plan.ensure(sizeof(int32_t) * (2 * (size_t)batch + 2 + 4 + 16)))) return rc; // two prefix arrays + {dense, total, head} // [2] n_active, [4..5] max_needed (u64), [6..7] evals (u64), [8], [9] LO queue heads of the two chunks

    {
        unsigned char *ph = h->params_host; // free: the previous call on this handle ended with a stream synchronisation
        std::memcpy(ph + off_state, tab_state.data(), sizeof(uint64_t) * n_tables);
        const bool cams = kind == MDRP_CALIB || kind == MDRP_RELPOSE_5PT || kind == MDRP_SHARED_6PT;
        // MDRP_SHARED_6PT reads the principal point from cam1 only (include/mdrp.h): cam2 may be NULL there
        if (cams) { std::memcpy(ph + off_cam1, cam1, sizeof(CamDev) * batch); std::memcpy(ph + off_cam2, cam2 ? cam2 : cam1, sizeof(CamDev) * batch); }
        else std::memset(ph + off_cam1, 0, 2 * sizeof(CamDev) * (size_t)batch);
        std::memcpy(ph + off_tn, tab_n.data(), sizeof(int32_t) * n_tables);
        std::memcpy(ph + off_tof, table_of.data(), sizeof(int32_t) * batch);
        std::memcpy(ph + off_nper, n_host, sizeof(int32_t) * batch);
        HIPCHK(hipMemcpyAsync(pd, ph, params_bytes, hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipEventRecord(h->ev_tables, s)); // sample tables can be drawn from here on

    RunParams rp;
    std::memset(&rp, 0, sizeof rp);
    rp.kind = kind; rp.solver = solver_for(kind, est_shift); rp.est_shift = est_shift;
    rp.score_initial = ro->score_initial_model ? 1 : 0;
    rp.batch = batch; rp.n_max = n_max;
    rp.max_iterations = ro->max_iterations; rp.min_iterations = ro->min_iterations;
    rp.dyn_mult = ro->dyn_num_trials_mult; rp.log_prob_missing = std::log(1.0 - ro->success_prob);
    rp.weight_sampson = ro->monodepth_weight_sampson > 0.0f ? (double)ro->monodepth_weight_sampson : 0.0; // the wrappers hand max(ws, 0) on (@0x2246e6)
    rp.final_max_it = (int)std::min<uint64_t>(bo->max_iterations, 1u << 30); rp.final_loss = bo->loss_type;
    rp.grad_tol = bo->gradient_tol; rp.step_tol = bo->step_tol; rp.lambda0 = bo->initial_lambda;
    rp.lambda_min = bo->min_lambda; rp.lambda_max = bo->max_lambda;
    rp.chunk_len = chunk_cap; rp.chunk_off = 0; rp.slot_stride = chunk_cap * mps; rp.super_len = chunk_cap; rp.chunk_start = 0;
    rp.mps = mps; rp.sample_sz = ssz;

    // normalisation, record layout, MFMA fragments and pair states of the pairs [p0, p0 + pc) (one workgroup per pair, bases offset by p0)
    auto launch_prep = [&](int p0, int pc, hipStream_t ps_) {
        const size_t o2 = (size_t)2 * p0 * n_max, o1 = (size_t)p0 * n_max;
        double *pts_p = h->pts.as<double>() + o1 * PT_STRIDE;
        uint4 *rfrag_p = h->rfrag.as<uint4>() + (size_t)p0 * groups_max * 64;
        if (classic)
            hipLaunchKernelGGL(kc_prep, dim3(pc), dim3(256), 0, ps_, rp, x1 + o2, x2 + o2, d_nper + p0, d_table_of + p0, d_cams1 + p0, d_cams2 + p0,
                               ro->max_epipolar_error, bo->loss_scale, pts_p, h->st.as<PairState>() + p0, rfrag_p);
        else
            hipLaunchKernelGGL(k_prep, dim3(pc), dim3(256), 0, ps_, rp, x1 + o2, x2 + o2, d1 + o1, d2 + o1, d_nper + p0, d_table_of + p0, d_cams1 + p0, d_cams2 + p0,
                               ro->max_epipolar_error, ro->max_reproj_error, bo->loss_scale, pts_p, h->dep.as<double>() + 2 * o1, h->st.as<PairState>() + p0, rfrag_p);
    };
    if (!host) { // device-resident inputs: everything is there (host buffers: the copies and k_prep are issued slice by slice below)
        launch_prep(0, batch, s);
        HIPCHK(hipGetLastError());
    }

    // Run-time knobs (DESIGN.md 10 lists all of them): the stream pipeline, the fp32 bound stage, lanes per LM problem.
    const bool lo_overlap = env_int("MDRP_LO_OVERLAP", 1) != 0; // three-stream pipeline; 0 = every kernel on the handle's stream
    const bool use_bound = env_int("MDRP_BOUND", 1) != 0;       // fp32 lower-bound stage between k_count and the fp64 sweep
    // lanes per LO problem: one wavefront when there are many short problems; four when the batch is small or the pairs are large (N = 5000:
    // a one-wavefront problem is 4 ms long and the launch ends with its stragglers — 45.6 against 43.9 ms per 1024 varying-focal pairs)
    // (chosen from the CALL's batch, not this pass's: the lane count fixes the LM's summation tree, and how a call is cut into passes depends on the
    // memory that happens to be free — a pair must get the same record whatever pass it falls into)
    const int lo_threads = env_int("MDRP_LO_THREADS", (batch_call >= 128 && n_max < 4096) ? 64 : 256);
    const int final_threads = env_int("MDRP_FINAL_THREADS", batch_call >= 4096 ? 64 : 256);
    // the inlier-only final refinement walks a compacted index of the inliers instead of masking every record (where the three lists fit 32 KiB of LDS)
    const int mask_index = lm_mask_index_on(n_max);
    // work lists of the classic LM (mdrp_classic.h ClmList)
    const int clm_list_stride = lm_list_stride(n_max);
    const size_t clm_list_bytes = (size_t)2 * clm_list_stride * sizeof(uint16_t);
    // Fused tail (mdrp_kernels.h FuseTail): when the end of the run is known on the host (the super-chunk reaches max_iterations), the
    // last LO launch replays each pair as its last trigger is refined (no k_walk launch), and k_final starts - on the main stream,
    // behind k_gate - as soon as the LO queue is empty, taking pairs in the order they became ready: the final refinements fill
    // the wavefront slots the LO's stragglers leave free, instead of LO | k_walk | k_final one after the other.
    // Measured per 1024 pairs (N = 2000, 10^4 iterations): calibrated 11.75 -> 11.2 ms, shared focal 12.5 -> 11.9, outlier-free 18.6 -> 17.4;
    // with the shift solver's 9-parameter LM the final refinements are 1.5x the LO's tail and gain nothing (12.75 -> 12.9 ms): off there.
    // After a bounded wait expired on this handle (kernels of two streams do not run side by side here: serialising profiler,
    // AMD_SERIALIZE_KERNEL, a busy shared GPU) the handle stays unfused, unless MDRP_FUSE_TAIL is set explicitly.
    // (fuse_disabled / fuse_retry_in are advanced once per API call in estimate_device, not per pass)
    const bool fuse_env = env_int("MDRP_FUSE_TAIL", ((kind == MDRP_CALIB && est_shift) || h->fuse_disabled) ? 0 : 1) != 0;
    bool final_done = false;
    // the 5-point LO keeps the inlier subset of the model it refines: one row per LO workgroup and chunk (LOs of two chunks overlap)
    if (kind == MDRP_RELPOSE_5PT && (rc = h->red5.ensure(sizeof(double) * (size_t)batch * ((size_t)(chunk_cap + 63) / 64) * RED5_STRIDE * 64))) return rc;
    const size_t lo_mask_rows = (size_t)h->num_cu * 8; // kc_lo launches num_cu * (8 | 2) workgroups
    if ((kind == MDRP_RELPOSE_5PT || kind == MDRP_SHARED_6PT) && (rc = h->lo_mask.ensure(lo_mask_rows * (size_t)std::max(n_max, 1)))) return rc;
    int32_t *cnt = h->counters.as<int32_t>();
    rp.inl_stat = classic ? nullptr : cnt + CNT_INL_STAT;
    const size_t tile_bytes = SCORE_TILE_BYTES;

    uint64_t it0 = 0;
    // iterations that certainly run: the reference cannot stop before min_iterations + 1 (or max_iterations)
    const uint64_t certain = ro->max_iterations == 0 ? 1 : std::min<uint64_t>(ro->max_iterations, ro->min_iterations + 1);
    // A SUPER-CHUNK is a range of iterations that shares one LO + walk pass and one host read-back; it is swept in one or
    // more CHUNKS (solve / count / bound / score / scan launch trains).  Inside the certain range nothing can stop, so the
    // first super-chunk is split into a short chunk (128 iterations) that establishes the records and the rest, whose
    // hypotheses are retired against those records by k_count (MFMA) and k_bound (fp32) unless they might break one.
    // Beyond the certain range a super-chunk is one chunk sized by the largest remaining dynamic_max_iter.
    // leading chunk lengths of the first super-chunk; the last chunk takes the rest.  The first chunk has no records to
    // retire anything against, so it is scored exactly in full: keep it short.  Every later chunk goes through k_count /
    // k_bound against the records of the chunks before it.  Measured on the benchmark shape: "128" 84.5 k pairs/s, "64" 82 k,
    // "256" 82.5 k, "512" 82 k, "256,768" 79 k (every chunk costs ~10 launches and a solver hand-over).  The 7-point estimator gets a second leading
    // chunk: at 50 % outliers one sample in 128 is outlier-free, so the records after 128 iterations are often those of a poor model and the rest of
    // the run would be counted and bounded against a bar that retires little (round 6: k_bound 5.7 ms of a 16.4 ms step; with "128,1024" the bar
    // the last 8848 iterations meet is that of 1152: 62.3 -> 87.3 k pairs/s; "128,512" 86.6 k, "256,1024" 87.6 k, "128,3300,3300" 78.3 k)
    std::vector<uint64_t> lead;
    {
        const char *e = getenv("MDRP_CHUNKS");
        // Round 6, with the two-phase count: at 50 % outliers the step is flat in the first chunk's length from 128 to 512 iterations (calibrated P3P
        // 8.45-8.49 / 8.44-8.55 / 8.40-8.41 / 8.51-8.66 ms at 128 / 256 / 384 / 512; shared focal 8.05 -> 7.99 at 256 or 384) — what a longer first
        // chunk costs in exact scoring it returns as a tighter bar — but NOT at other outlier ratios: one 3-point sample in 64 is outlier-free at
        // 75 % outliers, one in 300 at 85 %, and a pair whose first chunk holds none sends the whole rest of its run through the fp32 bound and the
        // exact sweep: 75 % outliers 124 k pairs/s with 128, 142 k with 256, 147 k with 384; 85 %: 78 k / 93 k / 104 k (136 k with 1024).  The
        // outlier-free shape pays for it the other way round (every hypothesis of the first chunk is a good one and is scored in full: 72.3 k with
        // 128, 71.1 k with 256, 69.5 k with 384).  Default: 256 for the 3-point estimators where the run is long enough to pay for it (a sixteenth of
        // the iterations that certainly run, between 128 and 256), up to 512 for the 5-point one by the same rule (solver-bound: flat), 128,1024 for the 7-point one
        // (above); MDRP_CHUNKS=384 or 128,1024 for data with fewer than one inlier in four.
        std::string spec = e ? e : (kind == MDRP_FUNDAMENTAL_7PT ? "128,1024" : "128");
        size_t pos = 0;
        while (pos < spec.size() && (int)lead.size() < mdrp_handle::NC_MAX - 1) {
            const size_t q = spec.find(',', pos);
            const long v = atol(spec.substr(pos, q == std::string::npos ? std::string::npos : q - pos).c_str());
            if (v > 0) lead.push_back((uint64_t)v);
            if (q == std::string::npos) break;
            pos = q + 1;
        }
        if (h->wish_kind >= 0 && hipEventQuery(h->ev_wish) == hipSuccess) { // an unfused run's sums have arrived
            if (h->wish_host[1] > 0) h->seen_wish[h->wish_kind] = (double)h->wish_host[0] / (double)h->wish_host[1];
            h->wish_kind = -1;
        }
        if (!e && kind == MDRP_RELPOSE_5PT) lead.assign(1, std::min<uint64_t>(512, std::max<uint64_t>(128, certain / 16 / 64 * 64)));
        if (!e && !classic) {
            lead.assign(1, std::min<uint64_t>(256, std::max<uint64_t>(128, certain / 16 / 64 * 64)));
            // ... and where the handle's previous call with this estimator (kind 0..2 here) has results to go by: the mean over its pairs of what each
            // would have liked (first_chunk_wish, mdrp_kernels.h: 6 / r^3 iterations for the pair's inlier ratio r, between 256 and 1024; 128 for nearly
            // outlier-free pairs) — the lengths the sweeps above found best at 0, 50, 75 and 85 % outliers, and for a batch that mixes them (20 / 50 / 70 /
            // 85 % by pair: 93.7 k pairs/s at 128, 101 k at 256, 105 k at 384-512, 104 k at 1024).  A long run only (the first chunk stays under an
            // eighth of it).
            if (h->seen_wish[kind] >= 0.0 && certain >= 8192)
                lead[0] = (uint64_t)std::min(1024.0, std::max(128.0, std::ceil(h->seen_wish[kind] / 64.0) * 64.0));
        }
    }
    uint64_t max_needed = 0;
    rp.slot_stride = chunk_cap * mps;
    while (true) {
        uint64_t lens[mdrp_handle::NC_MAX] = {0};
        int n_chunks = 1;
        if (it0 < certain) {
            const uint64_t span = std::min<uint64_t>(certain - it0, (uint64_t)chunk_cap);
            uint64_t used = 0;
            n_chunks = 0;
            if (it0 == 0)
                for (uint64_t l : lead) { // a leading chunk only while at least as much again remains behind it
                    if (used + 2 * l > span) break;
                    lens[n_chunks++] = l; used += l;
                }
            lens[n_chunks++] = span - used;
        } else {
            lens[0] = std::min<uint64_t>(std::min<uint64_t>(std::max<uint64_t>(max_needed, 256), ro->max_iterations - it0), (uint64_t)chunk_cap);
        }
        uint64_t super_len = 0;
        for (int c = 0; c < n_chunks; ++c) super_len += lens[c];
        if (it0 == 0) h->first_chunk = (int64_t)lens[0];
        rp.chunk_start = it0; rp.super_len = (int)super_len;
        HIPCHK(hipMemsetAsync(h->counters.p, 0, COUNTERS_BYTES, s));
        if (it0 == 0) HIPCHK(hipMemsetAsync(h->cand_stat.p, 0, sizeof(unsigned long long) * 2 * batch, s));
        // Three-stream pipeline over the chunks of a super-chunk (the benchmark shape: 128 | 9872 iterations):
        //   main:  prep solve0 count0 sort0 score0 scan0 | (wait solve1) count1 bound1 sort1 score1 scan1 | gate, final refinements (fused tail) | walk
        //   aux :       (after solve0) solve1 ...
        //   aux2:  samples0 samples1 (beside prep)                                                        | (after the last scan) LO: ALL triggers of the super-chunk
        // Chunk c + 1 is solved while chunk c is swept.  Chunks alternate between two sets of tag lists / model counters / sample tables; slots
        // and triggers of different chunks are disjoint.  A super-chunk has ONE LO launch, behind its last scan (round 5: two launches, each with
        // its own tail of long problems, cost 4.3 ms where one costs 2.7; LO problems are independent of each other, so results are bit-identical).
        const bool piped = n_chunks > 1 && lo_overlap;
        const bool fuse_tail = fuse_env && piped && it0 + super_len >= ro->max_iterations;
        int32_t *fz_ctl = nullptr, *fz_done = nullptr, *fz_fin = nullptr, *fz_ready = nullptr;
        if (fuse_tail) {
            if ((rc = h->fuse.ensure(64 + 3 * sizeof(int32_t) * (size_t)batch))) return rc;
            fz_ctl = h->fuse.as<int32_t>(); fz_done = fz_ctl + 16; fz_fin = fz_done + batch; fz_ready = fz_fin + batch;
        }
        hipStream_t aux = piped ? h->aux_stream : s, aux2 = piped ? h->aux_stream2 : s;
        int offs[mdrp_handle::NC_MAX] = {0};
        for (int c = 1; c < n_chunks; ++c) offs[c] = offs[c - 1] + (int)lens[c - 1];
        // The sample tables of the first two chunks do not depend on anything but (seed, N): they are drawn on the (still idle)
        // LO stream while k_prep runs, so the one-workgroup-per-table sampler (0.18 ms for 10^4 samples) is off the solver's path.
        const int samp_threads = env_int("MDRP_SAMPLE_THREADS", ssz == 3 ? SAMP_THREADS : (ssz == 5 ? 512 : 256)); // ~ samples between two rejections
        auto launch_samples = [&](hipStream_t st_, int len_, uint32_t *smp_) {
            if (ssz == 6) hipLaunchKernelGGL(kc_samples<6>, dim3(n_tables), dim3(samp_threads), 0, st_, n_tables, d_table_n, d_table_state, len_, smp_);
            else if (ssz == 5) hipLaunchKernelGGL(kc_samples<5>, dim3(n_tables), dim3(samp_threads), 0, st_, n_tables, d_table_n, d_table_state, len_, smp_);
            else if (ssz == 7) hipLaunchKernelGGL(kc_samples<7>, dim3(n_tables), dim3(samp_threads), 0, st_, n_tables, d_table_n, d_table_state, len_, smp_);
            else hipLaunchKernelGGL(k_samples, dim3(n_tables), dim3(samp_threads), 0, st_, n_tables, d_table_n, d_table_state, len_, smp_);
        };
        bool presampled[2] = {false, false};
        if (piped && it0 == 0) {
            HIPCHK(hipStreamWaitEvent(aux2, h->ev_tables, 0));
            for (int c = 0; c < 2 && c < n_chunks; ++c) {
                launch_samples(aux2, (int)lens[c], ((c & 1) ? h->samples2 : h->samples).as<uint32_t>());
                HIPCHK(hipEventRecord(h->ev_sampled[c], aux2));
                presampled[c] = true;
            }
        }
        // Every kernel of the front indexes its per-pair arrays as base[pair]: a RANGE of pairs [p0, p0 + pc) is the same launch on offset bases
        // with rp.batch = pc (the plans are per-launch scratch).  The whole batch is the range [0, batch); the host-buffer path below runs the
        // first chunk and the second chunk's solver slice by slice while later slices are still on their way over PCIe.
        auto issue_solve = [&](int c, hipStream_t st_, int p0, int pc) -> int {
            RunParams r = rp;
            r.chunk_len = (int)lens[c]; r.chunk_off = offs[c]; r.batch = pc;
            const bool odd = c & 1;
            const size_t so = (size_t)p0 * rp.slot_stride;
            uint32_t *tg = (odd ? h->tags2 : h->tags).as<uint32_t>() + so;
            uint32_t *smp = (odd ? h->samples2 : h->samples).as<uint32_t>();
            int32_t *mc = (odd ? h->model_count2 : h->model_count).as<int32_t>() + 2 * (size_t)p0;
            HIPCHK(hipMemsetAsync(mc, 0, sizeof(int32_t) * 2 * pc, st_));
            if (c < 2 && presampled[c]) HIPCHK(hipStreamWaitEvent(st_, h->ev_sampled[c], 0));
            else launch_samples(st_, r.chunk_len, smp);
            hipEvent_t v0, v1;
            int rc_;
            if ((rc_ = get_events(h, &v0, &v1, 5))) return rc_;
            HIPCHK(hipEventRecord(v0, st_));
            struct Stop { hipEvent_t e; hipStream_t s; ~Stop() { (void)hipEventRecord(e, s); } } stop_{v1, st_}; // after the solver launch below
            PairState *st_p = h->st.as<PairState>() + p0;
            double *pts_p = h->pts.as<double>() + (size_t)p0 * n_max * PT_STRIDE;
            Model *models_p = h->models.as<Model>() + so;
            int32_t *inl_p = h->slot_inl.as<int32_t>() + so;
            if (classic) {
                const dim3 sgrid((r.chunk_len + 63) / 64, pc);
                if (kind == MDRP_SHARED_6PT)
                    hipLaunchKernelGGL(kc_solve<CLASSIC_SHARED>, sgrid, dim3(64), 0, st_, r, st_p, smp, pts_p, models_p, inl_p, tg, mc);
                else if (kind == MDRP_RELPOSE_5PT) {
                    // three kernels (mdrp_classic.h): null space and roots + poses at six wavefronts per CU, the elimination between them at three
                    double *red5 = h->red5.as<double>() + (size_t)p0 * sgrid.x * RED5_STRIDE * 64;
                    hipLaunchKernelGGL(kc_solve5_null, sgrid, dim3(64), SOLVE5N_LDS_BYTES, st_, r, st_p, smp, pts_p, red5);
                    hipLaunchKernelGGL(kc_solve5_reduce, sgrid, dim3(64), SOLVE5_LDS_BYTES, st_, r, st_p, red5);
                    hipLaunchKernelGGL(kc_solve5_roots, sgrid, dim3(64), SOLVE5B_LDS_BYTES, st_, r, st_p, smp, pts_p, red5, models_p, inl_p, tg, mc);
                }
                else
                    hipLaunchKernelGGL(kc_solve<CLASSIC_FUND>, sgrid, dim3(64), SOLVE7_LDS_BYTES, st_, r, st_p, smp, pts_p, models_p, inl_p, tg, mc);
                return MDRP_OK;
            }
            double *dep_p = h->dep.as<double>() + (size_t)p0 * n_max * 2;
#define MDRP_SOLVE_LAUNCH(S)                                                                                                   \
    hipLaunchKernelGGL(k_solve<S>, dim3((r.chunk_len + solve_threads(S) - 1) / solve_threads(S), pc), dim3(solve_threads(S)), 0, st_, r, st_p, smp, pts_p, dep_p, models_p, inl_p, tg, mc, 0, r.chunk_len)
            switch (r.solver) {
            case SOLVER_P3P: MDRP_SOLVE_LAUNCH(SOLVER_P3P); break;
            case SOLVER_SHIFT: MDRP_SOLVE_LAUNCH(SOLVER_SHIFT); break;
            case SOLVER_SHARED: MDRP_SOLVE_LAUNCH(SOLVER_SHARED); break;
            default: MDRP_SOLVE_LAUNCH(SOLVER_VARYING); break;
            }
#undef MDRP_SOLVE_LAUNCH
            return MDRP_OK;
        };
        // count -> bound -> sort -> plan -> exact score -> scan of chunk c for the pairs [p0, p0 + pc), on the main stream
        auto sweep_chunk = [&](int c, int p0, int pc) -> int {
            RunParams r = rp;
            r.chunk_len = (int)lens[c]; r.chunk_off = offs[c]; r.batch = pc;
            const int len = r.chunk_len;
            const bool odd = c & 1;
            const size_t so = (size_t)p0 * rp.slot_stride;
            PairState *st_p = h->st.as<PairState>() + p0;
            const double *pts_p = h->pts.as<double>() + (size_t)p0 * n_max * PT_STRIDE;
            const Model *models_p = h->models.as<Model>() + so;
            uint32_t *tags_sc = (odd ? h->tags2_s : h->tags_s).as<uint32_t>() + so;
            int32_t *mcount_c = (odd ? h->model_count2 : h->model_count).as<int32_t>() + 2 * (size_t)p0;
            uint32_t *tags_c = (odd ? h->tags2 : h->tags).as<uint32_t>() + so;
            uint32_t *tags_v = h->tags_v.as<uint32_t>() + so;
            int32_t *surv1 = h->surv_count.as<int32_t>() + p0, *surv2 = h->surv2_count.as<int32_t>() + p0;
            hipEvent_t e0, e1;
            int rc_;
            if ((rc_ = get_events(h, &e0, &e1, 0))) return rc_;
            // candidate counts on the matrix cores against the records of the chunks before this one; survivors only go on
            unsigned long long *cstats = reinterpret_cast<unsigned long long *>(cnt + 6);
            {
                hipEvent_t c0, c1;
                if ((rc_ = get_events(h, &c0, &c1, 1))) return rc_;
                // phase A: every hypothesis over the pair's leading tiles (all tiles where the records do not allow a split); phase B: the undecided
                // ones over the rest (k_count, mdrp_kernels.h).  A run's first chunk has no record: phase A is the whole count, no phase B.
                const bool two_phase = !(it0 == 0 && c == 0);
                unsigned long long *cand_stat = h->cand_stat.as<unsigned long long>() + 2 * (size_t)p0;
                // The undecided list lives where this chunk's SORTED list will be written once the counts are done (k_sort_tags, below), the partial counts
                // in the other parity's sorted list, whose last reader was the previous chunk's exact sweep: no scratch of their own.
                uint32_t *tags_u = tags_sc;
                int32_t *und_part = reinterpret_cast<int32_t *>((odd ? h->tags_s : h->tags2_s).as<uint32_t>() + so), *und_cnt = h->und_count.as<int32_t>() + 2 * (size_t)p0;
                const uint4 *rfrag_p = h->rfrag.as<uint4>() + (size_t)p0 * groups_max * 64;
                hipLaunchKernelGGL(k_count_plan, dim3(1), dim3(PLAN_THREADS), 0, s, pc, st_p, mcount_c, 2, CNT_WG_MODELS, h->cplan.as<int32_t>(),
                                   surv1, (const int32_t *)nullptr, two_phase ? und_cnt : (int32_t *)nullptr); // (also clears the counters k_count appends to)
                const dim3 cgrid((unsigned)pc * (unsigned)((len * mps + CNT_WG_MODELS - 1) / CNT_WG_MODELS));
                HIPCHK(hipEventRecord(c0, s));
                MDRP_SWEEP_DISPATCH(k_count, kind, cgrid, dim3(CNT_THREADS), 0, s, r, st_p, rfrag_p, models_p,
                                    tags_c, mcount_c, h->cplan.as<int32_t>(), tags_v, surv1, cstats, (int32_t *)nullptr, (const int32_t *)nullptr,
                                    1, tags_u, und_part, und_cnt, (const int32_t *)nullptr, cand_stat);
                if (two_phase) {
                    hipLaunchKernelGGL(k_count_plan, dim3(1), dim3(PLAN_THREADS), 0, s, pc, st_p, und_cnt, 2, CNT_WG_MODELS, h->cplan.as<int32_t>(),
                                       (int32_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr);
                    MDRP_SWEEP_DISPATCH(k_count, kind, cgrid, dim3(CNT_THREADS), 0, s, r, st_p, rfrag_p, models_p,
                                        tags_u, und_cnt, h->cplan.as<int32_t>(), tags_v, surv1, cstats, (int32_t *)nullptr, (const int32_t *)nullptr,
                                        2, (uint32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, und_part, cand_stat);
                    h->count_launches++; // (mdrp_stats::count_launches counts kernel launches: a rocprof summary shows the same number)
                }
                HIPCHK(hipEventRecord(c1, s));
                h->count_launches++;
            }
            const uint32_t *surv_tags = tags_v;
            const int32_t *surv_cnt = surv1;
            // (calls of a few pairs skip the fp32 stage: its lane-per-hypothesis record loop is 0.1 ms of latency, and k_score_w takes k_count's survivors at once)
            if (use_bound && batch_call > 16 && !(it0 == 0 && c == 0)) { // a run's first chunk has no records yet: nothing to retire
                // fp32 lower bound of the score for k_count's survivors; its survivors go back into the chunk's tag list
                hipLaunchKernelGGL(k_count_plan, dim3(1), dim3(PLAN_THREADS), 0, s, pc, st_p, surv1, 1, BND_THREADS, h->cplan.as<int32_t>(), surv2, (const int32_t *)nullptr);
                const dim3 bgrid((unsigned)pc * (unsigned)((len * mps + BND_THREADS - 1) / BND_THREADS));
                unsigned long long *bstats = reinterpret_cast<unsigned long long *>(cnt + 12);
                hipEvent_t b0, b1;
                if ((rc_ = get_events(h, &b0, &b1, 4))) return rc_;
                HIPCHK(hipEventRecord(b0, s));
                MDRP_SWEEP_DISPATCH(k_bound, kind, bgrid, dim3(BND_THREADS), 0, s, r, st_p, pts_p, models_p, tags_v, surv1, h->cplan.as<int32_t>(), tags_c, surv2, bstats);
                HIPCHK(hipEventRecord(b1, s));
                surv_tags = tags_c; surv_cnt = surv2;
            }
            int32_t *plan = h->plan.as<int32_t>(), *totals = plan + 2 * (size_t)batch + 2; // (the slice's plan uses the head of the buffer, `totals` stays where it is)
            if (batch_call <= SCORE_WAVE_MAX_PAIRS) {
                // few pairs: one WAVEFRONT per hypothesis (k_score_w) — a lane per hypothesis is a 0.3 ms serial record loop however few there are
                hipLaunchKernelGGL(k_count_plan, dim3(1), dim3(PLAN_THREADS), 0, s, pc, st_p, surv_cnt, 1, SCW_THREADS / 64, plan, (int32_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr);
                hipLaunchKernelGGL(k_sort_tags, dim3(pc), dim3(256), 0, s, r, st_p, mcount_c, surv_cnt, surv_tags, tags_sc);
                HIPCHK(hipEventRecord(e0, s));
                const long long ub = (long long)pc * ((len * mps + SCW_THREADS / 64 - 1) / (SCW_THREADS / 64));
                const dim3 grid((unsigned)std::min<long long>(ub, (long long)h->num_cu * 32));
                MDRP_SWEEP_DISPATCH(k_score_w, kind, grid, dim3(SCW_THREADS), 0, s, r, st_p, pts_p, models_p, tags_sc, mcount_c,
                                    h->slot_score.as<double>() + so, h->slot_inl.as<int32_t>() + so, plan);
            } else {
                hipLaunchKernelGGL(k_sort_tags, dim3(pc), dim3(256), 0, s, r, st_p, mcount_c, surv_cnt, surv_tags, tags_sc);
                hipLaunchKernelGGL(k_plan, dim3(1), dim3(PLAN_THREADS), 0, s, pc, mcount_c, plan, totals);
                HIPCHK(hipEventRecord(e0, s));
                const dim3 grid((unsigned)pc * (unsigned)((len * mps + SCORE_THREADS - 1) / SCORE_THREADS));
                MDRP_SWEEP_DISPATCH(k_score, kind, grid, dim3(SCORE_THREADS), tile_bytes, s, r, st_p, pts_p, models_p, tags_sc, mcount_c,
                                    h->slot_score.as<double>() + so, h->slot_inl.as<int32_t>() + so, plan, totals);
            }
            HIPCHK(hipEventRecord(e1, s));
            h->sweep_launches++;
            Trigger *trig_p = h->triggers.as<Trigger>() + (size_t)p0 * trig_cap;
            unsigned long long *scan_stats = reinterpret_cast<unsigned long long *>(cnt + 10);
            // (four iterations per lane where the chunk is long: a 9872-iteration chunk is 39 steps of one wavefront instead of 154)
#define MDRP_SCAN(M, I) hipLaunchKernelGGL((k_scan<M, I>), dim3(pc), dim3(64), 0, s, r, st_p, h->slot_score.as<double>() + so, h->slot_inl.as<int32_t>() + so, trig_p, trig_cap, mcount_c, scan_stats)
            if (mps == 16) MDRP_SCAN(16, 1);
            else if (mps == 12) MDRP_SCAN(12, 1);
            else if (len >= 1024) MDRP_SCAN(4, 4);
            else MDRP_SCAN(4, 1);
#undef MDRP_SCAN
            return MDRP_OK;
        };
        // ---- host buffers (MDRP_MEM_HOST): the correspondences arrive slice by slice on the copy stream; k_prep and the second chunk's solver (the
        // long one: 1.6 ms per 1024 pairs) of slice i run while slice i + 1 is on its way over PCIe
        bool solved1 = false; // the second chunk's solver has been issued (slice by slice)
        int swept0 = 0;       // pairs whose first chunk the sliced front has swept already
        if (host && it0 == 0) {
            const bool sliced = piped && n_chunks >= 2 && batch >= 2 * HOST_SLICE_PAIRS;
            const int sl = sliced ? HOST_SLICE_PAIRS : batch;
            for (int p0 = 0; p0 < batch; p0 += sl) {
                const int pc = std::min(sl, batch - p0);
                const size_t o2 = (size_t)2 * p0 * n_max, o1 = (size_t)p0 * n_max, n2 = sizeof(double) * 2 * (size_t)pc * n_max, n1 = sizeof(double) * (size_t)pc * n_max;
                hipStream_t cs = sliced ? h->copy_stream : s;
                HIPCHK(hipMemcpyAsync(const_cast<double *>(x1) + o2, host->x1 + o2, n2, hipMemcpyHostToDevice, cs));
                HIPCHK(hipMemcpyAsync(const_cast<double *>(x2) + o2, host->x2 + o2, n2, hipMemcpyHostToDevice, cs));
                if (host->d1 && host->d2) {
                    HIPCHK(hipMemcpyAsync(const_cast<double *>(d1) + o1, host->d1 + o1, n1, hipMemcpyHostToDevice, cs));
                    HIPCHK(hipMemcpyAsync(const_cast<double *>(d2) + o1, host->d2 + o1, n1, hipMemcpyHostToDevice, cs));
                }
                if (!sliced) { launch_prep(p0, pc, s); break; }
                // k_prep and the long solver of the slice on the solver stream, beside the next slices' copies (the main stream sweeps the first chunk)
                HIPCHK(hipEventRecord(h->ev_copied, cs)); HIPCHK(hipStreamWaitEvent(aux, h->ev_copied, 0));
                if (p0 == 0) HIPCHK(hipStreamWaitEvent(aux, h->ev_tables, 0)); // (the per-call parameters were uploaded on the main stream)
                launch_prep(p0, pc, aux);
                HIPCHK(hipEventRecord(h->ev_prepped, aux));
                if ((rc = issue_solve(1, aux, p0, pc))) return rc;
                // The first chunk is swept in TWO parts only: everything but the last slice as soon as the last-but-one slice is prepared (beside the
                // last slice's copy), the last slice behind its own k_prep.  (Its exact sweep is a fixed ~0.5 ms of serial record loops per workgroup
                // whatever the number of pairs: one sweep per slice cost four times that on the main stream, 10.3 ms per step against 10.0.)
                const bool last = p0 + sl >= batch, last_but_one = !last && p0 + 2 * sl >= batch;
                if (last_but_one || last) HIPCHK(hipStreamWaitEvent(s, h->ev_prepped, 0)); // (the prep of this slice and, by stream order, of every slice before it)
                if (last_but_one) { swept0 = p0 + pc; if ((rc = issue_solve(0, s, 0, swept0)) || (rc = sweep_chunk(0, 0, swept0))) return rc; }
                if (last) { if ((rc = issue_solve(0, s, swept0, batch - swept0)) || (rc = sweep_chunk(0, swept0, batch - swept0))) return rc; swept0 = batch; }
            }
            if (sliced) { HIPCHK(hipEventRecord(h->ev_solved[1], aux)); solved1 = true; }
        }
        if (swept0 < batch && (rc = issue_solve(0, s, 0, batch))) return rc;
        if (piped) HIPCHK(hipEventRecord(h->ev_solved[0], s));
        for (int c = 0; c < n_chunks; ++c) {
            rp.chunk_len = (int)lens[c]; rp.chunk_off = offs[c];
            if (c == 0 && swept0 >= batch) { // (the sliced front has swept the first chunk already)
                if (piped) HIPCHK(hipEventRecord(h->ev_scanned[0], s));
                continue;
            }
            if (piped && c + 1 < n_chunks && !(c == 0 && solved1)) {
                // the sampler tables advance in chunk order; chunk c + 1 reuses the lists chunk c - 1 was swept from
                HIPCHK(hipStreamWaitEvent(aux, c == 0 ? h->ev_solved[0] : h->ev_scanned[c - 1], 0));
                if ((rc = issue_solve(c + 1, aux, 0, batch))) return rc;
                HIPCHK(hipEventRecord(h->ev_solved[c + 1], aux));
            }
            if (piped && c > 0) HIPCHK(hipStreamWaitEvent(s, h->ev_solved[c], 0));
            if (!piped && c > 0 && (rc = issue_solve(c, s, 0, batch))) return rc;
            if ((rc = sweep_chunk(c, 0, batch))) return rc;
            if (c + 1 < n_chunks) { // (the solver of chunk c + 2 waits for this chunk's scan: it reuses this chunk's lists)
                if (piped) HIPCHK(hipEventRecord(h->ev_scanned[c], s));
                continue;
            }
            // ---- behind the super-chunk's last scan: the LO of ALL its triggers, one persistent launch
            int32_t *lo_plan = h->work_pair.as<int32_t>();
            hipLaunchKernelGGL(k_lo_plan, dim3(1), dim3(PLAN_THREADS), 0, s, batch, h->st.as<PairState>(), (const int32_t *)nullptr, lo_plan);
            if (fuse_tail) {
                HIPCHK(hipMemsetAsync(fz_ctl, 0, 64 + 2 * sizeof(int32_t) * (size_t)batch, s));
                HIPCHK(hipMemsetAsync(fz_ready, 0xFF, sizeof(int32_t) * (size_t)batch, s));
            }
            if (piped) { HIPCHK(hipEventRecord(h->ev_scanned[c], s)); HIPCHK(hipStreamWaitEvent(aux2, h->ev_scanned[c], 0)); }
            const int lo_blocks = h->num_cu * (lo_threads == 64 ? 8 : 2);
            const FuseTail fz = fuse_tail ? FuseTail{fz_done, fz_ready, fz_ctl, h->st.as<PairState>()} : FuseTail{nullptr, nullptr, nullptr, nullptr};
            unsigned long long *lm_stats = h->lm_stats.as<unsigned long long>();
            int32_t *xheads = cnt + CNT_XCD_HEAD; // one LO queue per XCD (lo_take)
            hipEvent_t l0, l1; // HIP events on the stream the LO runs on (mdrp_stats::lo_ms)
            if ((rc = get_events(h, &l0, &l1, 2))) return rc;
            HIPCHK(hipEventRecord(l0, aux2));
            if (classic)
                MDRP_CLASSIC_LM_DISPATCH(kc_lo, lo_threads, kind, dim3(lo_blocks), clm_list_bytes, aux2, rp, h->st.as<PairState>(), h->pts.as<double>(),
                                         h->models.as<Model>(), h->triggers.as<Trigger>(), trig_cap, lo_plan, cnt + CNT_LO_HEAD, xheads,
                                         h->lo_mask.as<uint8_t>(), clm_list_stride, fz);
            else
                MDRP_LM_DISPATCH(k_lo, lo_threads, kind, est_shift, dim3(lo_blocks), lm_list_bytes(n_max), aux2, rp,
                                 h->st.as<PairState>(), h->pts.as<double>(), h->dep.as<double>(), h->models.as<Model>(), h->triggers.as<Trigger>(),
                                 trig_cap, lo_plan, cnt + CNT_LO_HEAD, xheads, lm_list_stride(n_max), lm_stats, fz);
            HIPCHK(hipEventRecord(l1, aux2));
        }
        if (fuse_tail) { // final refinements on the main stream, released when the last LO launch's queue is empty
            const int32_t *plan_l = h->work_pair.as<int32_t>();
            const int lo_blocks_l = h->num_cu * (lo_threads == 64 ? 8 : 2); // = lo_blocks of the launch above
            // bounded waits (k_gate): far beyond anything a healthy run needs (the LO queue of 1024 pairs is empty after ~1 ms)
            // ... and scaled with the problem size: one LO problem is ~0.4 ms at N = 2000 and grows linearly with N (4 ms at 5000 on one wavefront)
            // ... capped at 200 ms / 100 ms: under persistently serialised dispatch (rocprofv3 --pmc, AMD_SERIALIZE_KERNEL, a debugger) every re-try of the
            // fused tail pays these in full (3.3 s at N = 70001 before the cap)
            const long long n_scale = std::max(1, (n_max + 1999) / 2000);
            const unsigned long long gate_ticks = 100ull * (unsigned long long)env_int("MDRP_FUSE_GATE_US", (int)std::min<long long>((50000 + 40ll * batch) * n_scale, 200000));
            const unsigned long long wait_ticks = 100ull * (unsigned long long)env_int("MDRP_FUSE_WAIT_US", (int)std::min<long long>((20000 + 4ll * batch) * n_scale, 100000));
            hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, s, (const int32_t *)(cnt + CNT_LO_HEAD), plan_l + 3 * (size_t)batch + 1,
                               (const int32_t *)fz_ctl, lo_blocks_l, gate_ticks, h->lm_stats.as<unsigned long long>() + 4);
            hipEvent_t g0, g1;
            if ((rc = get_events(h, &g0, &g1, 3))) return rc;
            HIPCHK(hipEventRecord(g0, s));
            if (classic)
                MDRP_CLASSIC_LM_DISPATCH(kc_final, final_threads, kind, dim3(batch), clm_list_bytes, s, rp, h->st.as<PairState>(), h->pts.as<double>(), mask_dev, results_dev,
                                         (const int32_t *)fz_ready, fz_fin, wait_ticks, h->lm_stats.as<unsigned long long>() + 5, clm_list_stride);
            else
                MDRP_FINAL_DISPATCH(final_threads, rp.final_loss, kind, est_shift, dim3(batch), (mask_index ? lm_final_list_bytes(n_max) : lm_list_bytes(n_max)), s, rp, h->st.as<PairState>(),
                                 h->pts.as<double>(), h->dep.as<double>(), mask_dev, results_dev, lm_list_stride(n_max), mask_index,
                                 h->lm_stats.as<unsigned long long>() + 2, (const int32_t *)fz_ready, fz_fin, wait_ticks,
                                 h->lm_stats.as<unsigned long long>() + 5);
            HIPCHK(hipEventRecord(g1, s));
            final_done = true;
        }
        if (piped) { HIPCHK(hipEventRecord(h->ev_lo, aux2)); HIPCHK(hipStreamWaitEvent(s, h->ev_lo, 0)); }
        if (fuse_tail) { // behind the LO launch: the pairs a bounded wait gave up on (none in a healthy run: 1024 workgroups that read a flag)
            if (classic)
                MDRP_CLASSIC_LM_DISPATCH(kc_final, final_threads, kind, dim3(batch), clm_list_bytes, s, rp, h->st.as<PairState>(), h->pts.as<double>(), mask_dev, results_dev,
                                         (const int32_t *)nullptr, fz_fin, 0ull, (unsigned long long *)nullptr, clm_list_stride);
            else
                MDRP_FINAL_DISPATCH(final_threads, rp.final_loss, kind, est_shift, dim3(batch), (mask_index ? lm_final_list_bytes(n_max) : lm_list_bytes(n_max)), s, rp, h->st.as<PairState>(),
                                 h->pts.as<double>(), h->dep.as<double>(), mask_dev, results_dev, lm_list_stride(n_max), mask_index,
                                 h->lm_stats.as<unsigned long long>() + 2, (const int32_t *)nullptr, fz_fin, 0ull, (unsigned long long *)nullptr);
        }
        if (!fuse_tail)
            hipLaunchKernelGGL(k_walk, dim3((batch + 63) / 64), dim3(64), 0, s, rp, h->st.as<PairState>(), h->models.as<Model>(),
                               h->triggers.as<Trigger>(), trig_cap, cnt + 2, reinterpret_cast<unsigned long long *>(cnt + 4),
                               (const int32_t *)nullptr, 0, 0, 0);
        HIPCHK(hipGetLastError());
        // progress record: pairs still iterating, iterations they still need, evaluations swept (sum over pairs of models * n)
        HIPCHK(hipMemcpyAsync(h->progress_host, cnt + 2, sizeof(Progress), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        h->sweep_evals += (int64_t)h->progress_host->evals;
        h->mfma_evals += (int64_t)h->progress_host->evals_mfma;
        h->fp64_evals += (int64_t)h->progress_host->evals_sweep;
        h->bound_evals += (int64_t)h->progress_host->evals_bound;
        if (final_done && h->progress_host->wish_pairs > 0) // (fused tail: the final refinements are behind this read-back; otherwise they are still to come)
            h->seen_wish[kind] = (double)h->progress_host->wish_sum / (double)h->progress_host->wish_pairs;
        if (getenv("MDRP_DEBUG"))
            fprintf(stderr, "[mdrp] super-chunk start %llu len %llu (%d chunks): evals %llu (mfma %llu, fp32 bound %llu, fp64 sweep %llu = %.2f %%) active %d max_needed %llu\n",
                    (unsigned long long)it0, (unsigned long long)super_len, n_chunks, h->progress_host->evals, h->progress_host->evals_mfma,
                    h->progress_host->evals_bound, h->progress_host->evals_sweep, 100.0 * (double)h->progress_host->evals_sweep / (double)std::max<unsigned long long>(1, h->progress_host->evals),
                    h->progress_host->n_active, h->progress_host->max_needed);
        it0 += super_len;
        if (h->progress_host->n_active == 0 || it0 >= ro->max_iterations) break;
        max_needed = h->progress_host->max_needed;
    }

    if (final_done) return MDRP_OK; // fused tail: the final refinements ran beside the last LO launch
    hipEvent_t f0, f1;
    if ((rc = get_events(h, &f0, &f1, 3))) return rc;
    HIPCHK(hipEventRecord(f0, s));
    if (classic)
        MDRP_CLASSIC_LM_DISPATCH(kc_final, final_threads, kind, dim3(batch), clm_list_bytes, s, rp, h->st.as<PairState>(), h->pts.as<double>(), mask_dev, results_dev,
                                 (const int32_t *)nullptr, (int32_t *)nullptr, 0ull, (unsigned long long *)nullptr, clm_list_stride);
    else {
        MDRP_FINAL_DISPATCH(final_threads, rp.final_loss, kind, est_shift, dim3(batch), (mask_index ? lm_final_list_bytes(n_max) : lm_list_bytes(n_max)), s, rp, h->st.as<PairState>(),
                         h->pts.as<double>(), h->dep.as<double>(), mask_dev, results_dev, lm_list_stride(n_max), mask_index,
                         h->lm_stats.as<unsigned long long>() + 2, (const int32_t *)nullptr, (int32_t *)nullptr, 0ull, (unsigned long long *)nullptr);
    }
    HIPCHK(hipEventRecord(f1, s));
    if (rp.inl_stat && h->wish_kind < 0) { // (the fused tail's sums came with the progress record; these arrive when the stream gets here: the next call looks)
        HIPCHK(hipMemcpyAsync(h->wish_host, rp.inl_stat, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(h->ev_wish, s));
        h->wish_kind = kind;
    }
    HIPCHK(hipGetLastError());
    return MDRP_OK;
}

int estimate_device(mdrp_handle *h, int kind, const double *x1, const double *x2, const double *d1, const double *d2, int batch,
                    int n_max, const int32_t *n_per_pair, const mdrp_camera *cam1, const mdrp_camera *cam2,
                    const mdrp_ransac_opt *ro, const mdrp_bundle_opt *bo, uint8_t *mask_dev, const HostSrc *host = nullptr) {
    const bool known_kind = kind >= 0 && kind <= 5;
    if (!h || batch < 0 || n_max < 0 || !known_kind || !ro || !bo) { g_err = "invalid argument"; return MDRP_ERR_INVALID; }
    if (kind == MDRP_SHARED_6PT && batch > 0 && !cam1) { g_err = "the 6-point estimator needs the principal point in cam1"; return MDRP_ERR_INVALID; }
    if ((kind == MDRP_CALIB || kind == MDRP_RELPOSE_5PT) && batch > 0 && (!cam1 || !cam2)) { g_err = "calibrated estimator needs cameras"; return MDRP_ERR_INVALID; }
    if (kind <= 2 && batch > 0 && n_max > 0 && (!d1 || !d2)) { g_err = "monodepth estimator needs depths"; return MDRP_ERR_INVALID; }
    // RansacOptions switches of the reference that are not built are refused, never ignored (the reference would return different results)
    if (ro->progressive_sampling) { g_err = "progressive_sampling (PROSAC, RandomSampler::initialize_prosac) is not built"; return MDRP_ERR_UNSUPPORTED; }
    if (ro->real_focal_check && (kind == MDRP_SHARED_6PT || kind == MDRP_FUNDAMENTAL_7PT)) { g_err = "real_focal_check is not built"; return MDRP_ERR_UNSUPPORTED; }
    const int mps = kind == MDRP_RELPOSE_5PT ? 12 : (kind == MDRP_SHARED_6PT ? 16 : 4);
    if (h->fuse_disabled && --h->fuse_retry_in <= 0) h->fuse_disabled = false; // once per API call (not per pass): the handle tries the fused tail again
    h->ev_used = 0; h->sweep_ms = 0; h->sweep_launches = 0; h->sweep_evals = 0; h->mfma_evals = 0; h->fp64_evals = 0; h->bound_evals = 0; h->count_launches = 0; h->count_ms = 0; h->last_batch = batch; h->lm_cost_evals = 0; h->lm_accum_evals = 0;
    int rc;
    if ((rc = h->results.ensure(sizeof(ResultDev) * std::max(batch, 1)))) return rc;
    if ((rc = h->lm_stats.ensure(LM_STATS_BYTES))) return rc;
    HIPCHK(hipMemsetAsync(h->lm_stats.p, 0, LM_STATS_BYTES, h->stream));
    if (batch == 0) return MDRP_OK;
    std::vector<int32_t> n_host(batch);
    for (int i = 0; i < batch; ++i) {
        n_host[i] = n_per_pair ? n_per_pair[i] : n_max;
        if (n_host[i] < 0 || n_host[i] > n_max) { g_err = "n_per_pair out of range"; return MDRP_ERR_INVALID; }
    }
    uint8_t *mask = mask_dev;
    if (!mask) {
        if ((rc = h->mask.ensure((size_t)batch * std::max(n_max, 1)))) return rc;
        mask = h->mask.as<uint8_t>();
    }
    // chunk capacity and pairs per pass from the scratch budget
    uint64_t chunk_cap64 = std::min<uint64_t>(std::max<uint64_t>(ro->max_iterations, 1), std::max<uint64_t>(ro->min_iterations + 1, 4096));
    chunk_cap64 = std::min<uint64_t>(chunk_cap64, 16384);
    const int chunk_cap = (int)chunk_cap64;
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t per_pair = (size_t)chunk_cap * mps * (sizeof(Model) + sizeof(double) + 2 * sizeof(int32_t) + 4 * sizeof(uint32_t) /*tag lists*/) +
                            (size_t)chunk_cap * (sizeof(Trigger) + 8) + (size_t)n_max * (PT_STRIDE + 2) * sizeof(double) + 1024;
    const size_t per_pair_all = per_pair + (size_t)chunk_cap * mps * sizeof(uint32_t) /*tags_v*/ + ((size_t)n_max + 15) / 16 * 1024 /*rfrag*/ +
                                (kind == MDRP_RELPOSE_5PT ? ((size_t)chunk_cap + 63) / 64 * RED5_STRIDE * 64 * sizeof(double) : 0) /*red5*/;
    size_t budget = std::min<size_t>((size_t)(0.5 * (double)free_b), (size_t)96 << 30);
    int per_pass = (int)std::max<size_t>(1, std::min<size_t>((size_t)batch, budget / per_pair_all));
    per_pass = std::min(per_pass, 65535); // k_solve / k_probe put the pair index on grid.y
    per_pass = std::max(1, std::min(per_pass, env_int("MDRP_PAIRS_PER_PASS", per_pass))); // (tests: several passes on a small batch)
    for (int p0 = 0; p0 < batch; p0 += per_pass) {
        const int nb = std::min(per_pass, batch - p0);
        HostSrc hs{};
        if (host) hs = HostSrc{host->x1 + (size_t)2 * p0 * n_max, host->x2 + (size_t)2 * p0 * n_max, host->d1 ? host->d1 + (size_t)p0 * n_max : nullptr,
                               host->d2 ? host->d2 + (size_t)p0 * n_max : nullptr};
        rc = run_pass(h, kind, x1 + (size_t)2 * p0 * n_max, x2 + (size_t)2 * p0 * n_max, d1 ? d1 + (size_t)p0 * n_max : nullptr,
                      d2 ? d2 + (size_t)p0 * n_max : nullptr, nb,
                      n_max, n_host.data() + p0, cam1 ? cam1 + p0 : nullptr, cam2 ? cam2 + p0 : nullptr, ro, bo, chunk_cap,
                      mask + (size_t)p0 * n_max, h->results.as<ResultDev>() + p0, host ? &hs : nullptr, batch);
        if (rc) return rc;
    }
    return MDRP_OK;
}

int finish_timing(mdrp_handle *h) {
    if (h->lm_stats.p) HIPCHK(hipMemcpyAsync(h->lm_stats_host, h->lm_stats.p, LM_STATS_BYTES, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->lm_cost_evals = (int64_t)h->lm_stats_host[0]; h->lm_accum_evals = (int64_t)h->lm_stats_host[1];
    h->fin_cost_evals = (int64_t)h->lm_stats_host[2]; h->fin_accum_evals = (int64_t)h->lm_stats_host[3];
    h->fuse_gate_timeouts = (int64_t)h->lm_stats_host[4]; h->fuse_wait_timeouts = (int64_t)h->lm_stats_host[5];
    if (!(h->fuse_gate_timeouts || h->fuse_wait_timeouts) && !h->fuse_disabled) h->fuse_backoff = 64; // a clean call (fused or not needed): the back-off starts over
    if ((h->fuse_gate_timeouts || h->fuse_wait_timeouts) && !h->fuse_disabled) {
        // results are unaffected (the pass behind the LO launch refined what the waits gave up on), the call was slower than unfused.
        // Exponential back-off: 64, 128, ... 16384 unfused API calls after consecutive expired waits (a busy moment on a shared GPU costs 64 calls,
        // a profiler that serialises every dispatch soon costs nothing)
        h->fuse_disabled = true;
        h->fuse_retry_in = h->fuse_backoff;
        h->fuse_backoff = std::min(h->fuse_backoff * 2, 16384);
        if (!getenv("MDRP_QUIET"))
            fprintf(stderr, "[mdrp] fused tail: %lld gate / %lld final-refinement waits timed out (kernels of two streams did not overlap: "
                            "profiler or serialised dispatch?); this handle runs unfused for its next %d calls\n",
                    (long long)h->fuse_gate_timeouts, (long long)h->fuse_wait_timeouts, h->fuse_retry_in);
    }
    for (int k = 0; k < 6; ++k) { h->kind_ms[k] = 0; h->kind_launches[k] = 0; }
    for (size_t i = 0; i < h->ev_used; ++i) {
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, h->ev_pool[i].first, h->ev_pool[i].second));
        const int k = std::min(std::max(h->ev_what[i], 0), 5);
        h->kind_ms[k] += t; h->kind_launches[k]++;
    }
    h->sweep_ms = h->kind_ms[0]; h->count_ms = h->kind_ms[1];
    return MDRP_OK;
}

} // namespace

// serialise calls on one handle and run them on the handle's device; the caller's current device is restored on return
#define MDRP_ENTER(h)                                                                      \
    std::lock_guard<std::mutex> lock_((h)->mu);                                            \
    DeviceGuard guard_((h)->device);                                                       \
    if (!guard_.ok) { g_err = "hipSetDevice failed"; return MDRP_ERR_HIP; }

#ifndef MDRP_SRC_HASH
#define MDRP_SRC_HASH "unknown"
#endif

static int create_handle(int device, hipStream_t stream, bool own_stream, mdrp_handle **out);

extern "C" {

const char *mdrp_last_error(void) { return g_err.c_str(); }
// the build embeds a hash of the source files (mdrp_amd/build.py) so that a stale prebuilt library can be told from the tree
int mdrp_abi_version(void) { return MDRP_ABI_VERSION; }
const char *mdrp_version(void) { return "mdrp-hip 0.5 (gfx950) MDRP_SRC_HASH=" MDRP_SRC_HASH; }

// HIP_VERSION of the toolchain this library was compiled with (the runtime is bound at load time: mdrp_amd/_capi.py compares the two)
int mdrp_hip_build_version(void) { return HIP_VERSION; }

// the header's mdrp_create / mdrp_create_on_stream macros hand over the ABI the HOST was compiled against: another version, or another size of
// mdrp_ransac_opt, is refused here — before any call could read option fields past the end of a smaller struct
static int check_host_abi(int abi_version, int ransac_opt_bytes) {
    if (abi_version == MDRP_ABI_VERSION && ransac_opt_bytes == (int)sizeof(mdrp_ransac_opt)) return MDRP_OK;
    char buf[256];
    snprintf(buf, sizeof buf, "host was compiled against ABI %#x (mdrp_ransac_opt %d bytes), this library speaks %#x (%d bytes): recompile the host against include/mdrp.h",
             abi_version, ransac_opt_bytes, MDRP_ABI_VERSION, (int)sizeof(mdrp_ransac_opt));
    g_err = buf;
    return MDRP_ERR_INVALID;
}
int mdrp_create_(int device, void *stream, mdrp_handle **out, int abi_version, int ransac_opt_bytes) {
    if (int rc = check_host_abi(abi_version, ransac_opt_bytes)) return rc;
    return create_handle(device, (hipStream_t)stream, stream == nullptr, out);
}
int mdrp_create_on_stream_(int device, void *stream, mdrp_handle **out, int abi_version, int ransac_opt_bytes) {
    if (int rc = check_host_abi(abi_version, ransac_opt_bytes)) return rc;
    return create_handle(device, (hipStream_t)stream, false, out);
}

} // extern "C"

static int create_handle(int device, hipStream_t stream, bool own_stream, mdrp_handle **out) {
    if (!out) return MDRP_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_err = "no usable HIP device";
        return MDRP_ERR_NO_DEVICE;
    }
    DeviceGuard guard_(device);
    if (!guard_.ok) { g_err = "hipSetDevice failed"; return MDRP_ERR_HIP; }
    mdrp_handle *h = new mdrp_handle();
    h->device = device;
    if (!own_stream) { h->stream = stream; h->owns_stream = false; } // NULL here = the device's legacy default stream
    else { HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->owns_stream = true; }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipHostMalloc((void **)&h->progress_host, sizeof(Progress), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&h->lm_stats_host, LM_STATS_BYTES, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&h->wish_host, 2 * sizeof(int32_t), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&h->ev_wish, hipEventDisableTiming));
    std::memset(h->lm_stats_host, 0, LM_STATS_BYTES);
    {   // high priority: the few long LO wavefronts should be placed first, the sweeps fill the remaining slots
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        HIPCHK(hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, prio_hi));
        HIPCHK(hipStreamCreateWithPriority(&h->aux_stream2, hipStreamNonBlocking, prio_hi));
    }
    HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->ev_copied, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_prepped, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_lo, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_tables, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_sampled[0], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_sampled[1], hipEventDisableTiming));
    for (int i = 0; i < mdrp_handle::NC_MAX; ++i) {
        HIPCHK(hipEventCreateWithFlags(&h->ev_solved[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_scanned[i], hipEventDisableTiming));
    }
    const int tile_bytes = (int)SCORE_TILE_BYTES;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_score<true>), hipFuncAttributeMaxDynamicSharedMemorySize, tile_bytes));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_score<false>), hipFuncAttributeMaxDynamicSharedMemorySize, tile_bytes));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_score<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, tile_bytes));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kc_solve5_reduce), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SOLVE5_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kc_solver_unit<CLASSIC_RELPOSE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SOLVE5_LDS_BYTES));
    *out = h;
    return MDRP_OK;
}

extern "C" {

void mdrp_destroy(mdrp_handle *h) {
    if (!h) return;
    DeviceGuard guard_(h->device);
    (void)hipStreamSynchronize(h->stream);
    DevBuf *bufs[] = {&h->pts, &h->dep, &h->st, &h->samples, &h->params, &h->fuse, &h->models, &h->slot_score, &h->slot_inl, &h->tags, &h->model_count, &h->triggers, &h->work_pair,
                      &h->counters, &h->results, &h->mask, &h->in_x1, &h->in_x2, &h->in_d1, &h->in_d2, &h->unit_a,
                      &h->unit_b, &h->unit_c, &h->unit_d, &h->unit_e, &h->unit_f, &h->plan, &h->tags2, &h->model_count2, &h->samples2, &h->tags_s, &h->tags2_s,
                      &h->tags_v, &h->surv_count, &h->und_count, &h->cand_stat, &h->red5, &h->rfrag, &h->cplan, &h->surv2_count, &h->lo_mask,
                      &h->lm_stats};
    for (DevBuf *b : bufs) b->release();
    for (auto &e : h->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (h->progress_host) (void)hipHostFree(h->progress_host);
    if (h->wish_host) (void)hipHostFree(h->wish_host);
    if (h->ev_wish) (void)hipEventDestroy(h->ev_wish);
    if (h->lm_stats_host) (void)hipHostFree(h->lm_stats_host);
    if (h->params_host) (void)hipHostFree(h->params_host);
    if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
    if (h->aux_stream2) { (void)hipStreamSynchronize(h->aux_stream2); (void)hipStreamDestroy(h->aux_stream2); }
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    if (h->ev_copied) (void)hipEventDestroy(h->ev_copied);
    if (h->ev_prepped) (void)hipEventDestroy(h->ev_prepped);
    if (h->ev_lo) (void)hipEventDestroy(h->ev_lo);
    if (h->ev_tables) (void)hipEventDestroy(h->ev_tables);
    for (int i = 0; i < 2; ++i) if (h->ev_sampled[i]) (void)hipEventDestroy(h->ev_sampled[i]);
    for (int i = 0; i < mdrp_handle::NC_MAX; ++i) {
        if (h->ev_solved[i]) (void)hipEventDestroy(h->ev_solved[i]);
        if (h->ev_scanned[i]) (void)hipEventDestroy(h->ev_scanned[i]);
    }
    if (h->owns_stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int mdrp_synchronize(mdrp_handle *h) {
    if (!h) return MDRP_ERR_INVALID;
    MDRP_ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    return MDRP_OK;
}

int mdrp_estimate_batch_async(mdrp_handle *h, int kind, const double *x1, const double *x2, const double *d1, const double *d2,
                              int batch, int n_max, const int32_t *n_per_pair, const mdrp_camera *cam1, const mdrp_camera *cam2,
                              const mdrp_ransac_opt *ropt, const mdrp_bundle_opt *bopt, uint8_t *inlier_mask_dev) {
    if (!h) { g_err = "invalid argument"; return MDRP_ERR_INVALID; }
    MDRP_ENTER(h);
    return estimate_device(h, kind, x1, x2, d1, d2, batch, n_max, n_per_pair, cam1, cam2, ropt, bopt, inlier_mask_dev);
}

static int fetch_results_locked(mdrp_handle *h, mdrp_result *out, int batch) {
    if (!out || batch < 0 || batch > h->last_batch) { g_err = "invalid argument"; return MDRP_ERR_INVALID; }
    HIPCHK(hipMemcpyAsync(out, h->results.p, sizeof(ResultDev) * batch, hipMemcpyDeviceToHost, h->stream));
    return finish_timing(h);
}

int mdrp_copy_results_device(mdrp_handle *h, void *dst_dev, int batch) {
    if (!h || !dst_dev || batch < 0 || batch > h->last_batch) { g_err = "invalid argument"; return MDRP_ERR_INVALID; }
    MDRP_ENTER(h);
    HIPCHK(hipMemcpyAsync(dst_dev, h->results.p, sizeof(ResultDev) * batch, hipMemcpyDeviceToDevice, h->stream));
    return finish_timing(h); // waits for the handle's stream: the records may be consumed on any other stream afterwards
}

int mdrp_fetch_results(mdrp_handle *h, mdrp_result *out, int batch) {
    if (!h) { g_err = "invalid argument"; return MDRP_ERR_INVALID; }
    MDRP_ENTER(h);
    return fetch_results_locked(h, out, batch);
}

int mdrp_estimate_batch(mdrp_handle *h, int kind, int mem_space, const double *x1, const double *x2, const double *d1,
                        const double *d2, int batch, int n_max, const int32_t *n_per_pair, const mdrp_camera *cam1,
                        const mdrp_camera *cam2, const mdrp_ransac_opt *ropt, const mdrp_bundle_opt *bopt, mdrp_result *out,
                        uint8_t *inlier_mask) {
    if (!h || !out || batch < 0 || n_max < 0) { g_err = "invalid argument"; return MDRP_ERR_INVALID; }
    MDRP_ENTER(h);
    const size_t np = (size_t)batch * n_max;
    int rc;
    uint8_t *mask_dev = inlier_mask;
    HostSrc hsrc{};
    bool use_host = false;
    if (mem_space == MDRP_MEM_HOST) {
        if ((rc = h->in_x1.ensure(sizeof(double) * 2 * np + 16)) || (rc = h->in_x2.ensure(sizeof(double) * 2 * np + 16)) ||
            (rc = h->in_d1.ensure(sizeof(double) * np + 16)) || (rc = h->in_d2.ensure(sizeof(double) * np + 16)))
            return rc;
        // the copies are issued by run_pass, slice by slice on the handle's copy stream, beside the first kernels of the slices before them
        hsrc = HostSrc{x1, x2, (d1 && d2) ? d1 : nullptr, (d1 && d2) ? d2 : nullptr};
        if (d1 && d2) { d1 = h->in_d1.as<double>(); d2 = h->in_d2.as<double>(); }
        x1 = h->in_x1.as<double>(); x2 = h->in_x2.as<double>();
        use_host = true;
        mask_dev = nullptr; // handle-owned device mask, copied back below
    }
    rc = estimate_device(h, kind, x1, x2, d1, d2, batch, n_max, n_per_pair, cam1, cam2, ropt, bopt, mask_dev, use_host ? &hsrc : nullptr);
    if (rc) return rc;
    if (mem_space == MDRP_MEM_HOST && inlier_mask && np > 0)
        HIPCHK(hipMemcpyAsync(inlier_mask, h->mask.p, np, hipMemcpyDeviceToHost, h->stream));
    return fetch_results_locked(h, out, batch);
}

int mdrp_last_sweep_stats(mdrp_handle *h, double *sweep_ms, int64_t *launches, int64_t *evaluations) {
    if (!h) return MDRP_ERR_INVALID;
    std::lock_guard<std::mutex> lock_(h->mu);
    if (sweep_ms) *sweep_ms = h->sweep_ms;
    if (launches) *launches = h->sweep_launches;
    if (evaluations) *evaluations = h->sweep_evals;
    return MDRP_OK;
}

int mdrp_last_stats_sized(mdrp_handle *h, mdrp_stats *out, size_t out_size) {
    if (!h || !out || out_size < sizeof(double)) return MDRP_ERR_INVALID;
    std::lock_guard<std::mutex> lock_(h->mu);
    mdrp_stats full, *caller = out;
    out = &full;
    out->count_ms = h->count_ms; out->count_launches = h->count_launches;
    out->sweep_ms = h->sweep_ms; out->sweep_launches = h->sweep_launches;
    out->evals_algorithmic = h->sweep_evals; out->evals_mfma = h->mfma_evals; out->evals_fp64 = h->fp64_evals; out->evals_bound = h->bound_evals;
    out->lo_ms = h->kind_ms[2]; out->lo_launches = h->kind_launches[2]; out->final_ms = h->kind_ms[3]; out->final_launches = h->kind_launches[3];
    out->bound_ms = h->kind_ms[4]; out->bound_launches = h->kind_launches[4]; out->solve_ms = h->kind_ms[5]; out->solve_launches = h->kind_launches[5];
    out->lm_cost_evals = h->lm_cost_evals; out->lm_accum_evals = h->lm_accum_evals;
    out->final_cost_evals = h->fin_cost_evals; out->final_accum_evals = h->fin_accum_evals;
    out->fuse_gate_timeouts = h->fuse_gate_timeouts; out->fuse_wait_timeouts = h->fuse_wait_timeouts;
    out->first_chunk = h->first_chunk;
    std::memcpy(caller, &full, std::min(out_size, sizeof full)); // a caller compiled against an older, shorter struct gets its prefix
    return MDRP_OK;
}

// the round-3 entry point writes the round-3 struct (everything before fuse_gate_timeouts) and never more
int mdrp_last_stats(mdrp_handle *h, mdrp_stats *out) { return mdrp_last_stats_sized(h, out, offsetof(mdrp_stats, fuse_gate_timeouts)); }

int mdrp_solver_batch(mdrp_handle *h, int solver, const double *x1h, const double *x2h, const double *d1, const double *d2,
                      int count, mdrp_model *out, int32_t *n_out) {
    if (!h || count < 0 || solver < 0 || solver > 3 || !out || !n_out) { g_err = "invalid argument"; return MDRP_ERR_INVALID; }
    if (count == 0) return MDRP_OK;
    MDRP_ENTER(h);
    int rc;
    if ((rc = h->unit_a.ensure(sizeof(double) * 9 * count)) || (rc = h->unit_b.ensure(sizeof(double) * 9 * count)) ||
        (rc = h->unit_c.ensure(sizeof(double) * 3 * count)) || (rc = h->unit_d.ensure(sizeof(double) * 3 * count)) ||
        (rc = h->unit_e.ensure(sizeof(Model) * 4 * count)) || (rc = h->unit_f.ensure(sizeof(int32_t) * count)))
        return rc;
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(h->unit_a.p, x1h, sizeof(double) * 9 * count, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->unit_b.p, x2h, sizeof(double) * 9 * count, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->unit_c.p, d1, sizeof(double) * 3 * count, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->unit_d.p, d2, sizeof(double) * 3 * count, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(h->unit_e.p, 0, sizeof(Model) * 4 * count, s));
    hipLaunchKernelGGL(k_solver_unit, dim3((count + 63) / 64), dim3(64), 0, s, solver, count, h->unit_a.as<double>(),
                       h->unit_b.as<double>(), h->unit_c.as<double>(), h->unit_d.as<double>(), h->unit_e.as<Model>(), h->unit_f.as<int32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, h->unit_e.p, sizeof(Model) * 4 * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(n_out, h->unit_f.p, sizeof(int32_t) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return MDRP_OK;
}

int mdrp_classic_solver_batch(mdrp_handle *h, int kind, const double *x1h, const double *x2h, int count, mdrp_model *out, int32_t *n_out) {
    if (!h || count < 0 || kind < MDRP_RELPOSE_5PT || kind > MDRP_FUNDAMENTAL_7PT || !out || !n_out) { g_err = "invalid argument"; return MDRP_ERR_INVALID; }
    if (count == 0) return MDRP_OK;
    MDRP_ENTER(h);
    const int K = kind == MDRP_RELPOSE_5PT ? 5 : (kind == MDRP_SHARED_6PT ? 6 : 7), M = kind == MDRP_RELPOSE_5PT ? MAX_MODELS_5PT : (kind == MDRP_SHARED_6PT ? MAX_MODELS_6PT : 3);
    int rc;
    if ((rc = h->unit_a.ensure(sizeof(double) * 3 * K * count)) || (rc = h->unit_b.ensure(sizeof(double) * 3 * K * count)) ||
        (rc = h->unit_e.ensure(sizeof(Model) * M * count)) || (rc = h->unit_f.ensure(sizeof(int32_t) * count)))
        return rc;
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(h->unit_a.p, x1h, sizeof(double) * 3 * K * count, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->unit_b.p, x2h, sizeof(double) * 3 * K * count, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(h->unit_e.p, 0, sizeof(Model) * M * count, s));
    if (kind == MDRP_SHARED_6PT)
        hipLaunchKernelGGL(kc_solver_unit<CLASSIC_SHARED>, dim3((count + 63) / 64), dim3(64), 0, s, count, h->unit_a.as<double>(),
                           h->unit_b.as<double>(), h->unit_e.as<Model>(), h->unit_f.as<int32_t>());
    else if (kind == MDRP_RELPOSE_5PT)
        hipLaunchKernelGGL(kc_solver_unit<CLASSIC_RELPOSE>, dim3((count + 63) / 64), dim3(64), SOLVE5_LDS_BYTES, s, count, h->unit_a.as<double>(),
                           h->unit_b.as<double>(), h->unit_e.as<Model>(), h->unit_f.as<int32_t>());
    else
        hipLaunchKernelGGL(kc_solver_unit<CLASSIC_FUND>, dim3((count + 63) / 64), dim3(64), SOLVE7_LDS_BYTES, s, count, h->unit_a.as<double>(),
                           h->unit_b.as<double>(), h->unit_e.as<Model>(), h->unit_f.as<int32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, h->unit_e.p, sizeof(Model) * M * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(n_out, h->unit_f.p, sizeof(int32_t) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return MDRP_OK;
}

int mdrp_score_models(mdrp_handle *h, int kind, int mem_space, const mdrp_model *models, int num_models, const double *x1,
                      const double *x2, int n, double sq_threshold, double *scores, int32_t *counts) {
    if (!h || num_models < 0 || n < 0 || kind < 0 || kind > 5 || !scores || !counts) { g_err = "invalid argument"; return MDRP_ERR_INVALID; }
    if (num_models == 0) return MDRP_OK;
    MDRP_ENTER(h);
    hipStream_t s = h->stream;
    const int chunk = (num_models + 3) / 4;
    const size_t slots = (size_t)chunk * 4;
    int rc;
    if ((rc = h->pts.ensure(sizeof(double) * PT_STRIDE * std::max(n, 1))) || (rc = h->st.ensure(sizeof(PairState))) ||
        (rc = h->models.ensure(sizeof(Model) * slots)) || (rc = h->slot_score.ensure(sizeof(double) * slots)) ||
        (rc = h->slot_inl.ensure(sizeof(int32_t) * slots)) || (rc = h->tags.ensure(sizeof(uint32_t) * slots)) ||
        (rc = h->model_count.ensure(2 * sizeof(int32_t))))
        return rc;
    const double *x1d = x1, *x2d = x2;
    const Model *md = reinterpret_cast<const Model *>(models);
    if (mem_space == MDRP_MEM_HOST) {
        if ((rc = h->in_x1.ensure(sizeof(double) * 2 * std::max(n, 1))) || (rc = h->in_x2.ensure(sizeof(double) * 2 * std::max(n, 1)))) return rc;
        HIPCHK(hipMemcpyAsync(h->in_x1.p, x1, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->in_x2.p, x2, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->models.p, models, sizeof(Model) * num_models, hipMemcpyHostToDevice, s));
        x1d = h->in_x1.as<double>(); x2d = h->in_x2.as<double>();
        md = h->models.as<Model>();
    } else {
        HIPCHK(hipMemcpyAsync(h->models.p, models, sizeof(Model) * num_models, hipMemcpyDeviceToDevice, s));
        md = h->models.as<Model>();
    }
    std::vector<uint32_t> tags(num_models);
    for (int i = 0; i < num_models; ++i) tags[i] = (uint32_t)i;
    HIPCHK(hipMemcpyAsync(h->tags.p, tags.data(), sizeof(uint32_t) * num_models, hipMemcpyHostToDevice, s));
    const int32_t counts2[2] = {num_models, 0};
    HIPCHK(hipMemcpyAsync(h->model_count.p, counts2, 2 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    PairState ps;
    std::memset(&ps, 0, sizeof ps);
    ps.n = n; ps.active = 1; ps.sq_thr = sq_threshold; ps.eps = std::sqrt(sq_threshold);
    ps.best_min_score = DBL_MAX; // no records: nothing is pruned
    HIPCHK(hipMemcpyAsync(h->st.p, &ps, sizeof ps, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pack_unit, dim3((n + 255) / 256 + 1), dim3(256), 0, s, n, x1d, x2d, (const double *)nullptr,
                       (const double *)nullptr, h->pts.as<double>(), (double *)nullptr);
    hipLaunchKernelGGL(k_box_unit, dim3(1), dim3(256), 0, s, n, h->pts.as<double>(), h->st.as<PairState>());
    RunParams rp;
    std::memset(&rp, 0, sizeof rp);
    rp.kind = kind; rp.batch = 1; rp.n_max = std::max(n, 1); rp.chunk_len = chunk; rp.chunk_off = 0; rp.slot_stride = chunk * 4; rp.super_len = chunk;
    rp.mps = 4; rp.sample_sz = 3;
    const size_t tile_bytes = SCORE_TILE_BYTES;
    if ((rc = h->plan.ensure(sizeof(int32_t) * 8))) return rc;
    int32_t *plan = h->plan.as<int32_t>(), *totals = plan + 4;
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(PLAN_THREADS), 0, s, 1, h->model_count.as<int32_t>(), plan, totals);
    const dim3 grid((unsigned)std::min(h->num_cu * 4, (num_models + SCORE_THREADS - 1) / SCORE_THREADS));
    h->ev_used = 0; h->sweep_launches = 1; h->sweep_evals = (int64_t)num_models * n;
    hipEvent_t e0, e1;
    if ((rc = get_events(h, &e0, &e1))) return rc;
    HIPCHK(hipEventRecord(e0, s));
    MDRP_SWEEP_DISPATCH(k_score, kind, grid, dim3(SCORE_THREADS), tile_bytes, s, rp, h->st.as<PairState>(), h->pts.as<double>(), md,
                        h->tags.as<uint32_t>(), h->model_count.as<int32_t>(), h->slot_score.as<double>(), h->slot_inl.as<int32_t>(), plan, totals);
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipGetLastError());
    const hipMemcpyKind back = mem_space == MDRP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIPCHK(hipMemcpyAsync(scores, h->slot_score.p, sizeof(double) * num_models, back, s));
    HIPCHK(hipMemcpyAsync(counts, h->slot_inl.p, sizeof(int32_t) * num_models, back, s));
    return finish_timing(h);
}

int mdrp_count_candidates(mdrp_handle *h, int kind, const mdrp_model *models, int num_models, const double *x1, const double *x2,
                          int n, double sq_threshold, int32_t *candidates) {
    if (!h || num_models < 0 || n < 0 || kind < 0 || kind > 5 || !candidates || !models) { g_err = "invalid argument"; return MDRP_ERR_INVALID; }
    if (num_models == 0) return MDRP_OK;
    MDRP_ENTER(h);
    hipStream_t s = h->stream;
    const int nn = std::max(n, 1);
    const size_t slots = (size_t)num_models;
    int rc;
    if ((rc = h->pts.ensure(sizeof(double) * PT_STRIDE * nn)) || (rc = h->st.ensure(sizeof(PairState))) ||
        (rc = h->models.ensure(sizeof(Model) * slots)) || (rc = h->slot_inl.ensure(sizeof(int32_t) * slots)) ||
        (rc = h->tags.ensure(sizeof(uint32_t) * slots)) || (rc = h->tags_v.ensure(sizeof(uint32_t) * slots)) ||
        (rc = h->model_count.ensure(2 * sizeof(int32_t))) || (rc = h->surv_count.ensure(sizeof(int32_t))) ||
        (rc = h->cplan.ensure(2 * sizeof(int32_t))) || (rc = h->rfrag.ensure((size_t)((nn + 15) / 16) * 1024)) ||
        (rc = h->unit_f.ensure(sizeof(int32_t) * slots)) || (rc = h->in_x1.ensure(sizeof(double) * 2 * nn)) ||
        (rc = h->in_x2.ensure(sizeof(double) * 2 * nn)))
        return rc;
    HIPCHK(hipMemcpyAsync(h->in_x1.p, x1, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->in_x2.p, x2, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->models.p, models, sizeof(Model) * num_models, hipMemcpyHostToDevice, s));
    std::vector<uint32_t> tags(num_models);
    for (int i = 0; i < num_models; ++i) tags[i] = (uint32_t)i;
    HIPCHK(hipMemcpyAsync(h->tags.p, tags.data(), sizeof(uint32_t) * num_models, hipMemcpyHostToDevice, s));
    const int32_t counts2[2] = {num_models, 0};
    HIPCHK(hipMemcpyAsync(h->model_count.p, counts2, 2 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(h->surv_count.p, 0, sizeof(int32_t), s));
    PairState ps;
    std::memset(&ps, 0, sizeof ps);
    ps.n = n; ps.active = 1; ps.sq_thr = sq_threshold; ps.eps = std::sqrt(sq_threshold);
    ps.best_min_score = DBL_MAX; // no records: nothing is retired
    HIPCHK(hipMemcpyAsync(h->st.p, &ps, sizeof ps, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pack_unit, dim3((n + 255) / 256 + 1), dim3(256), 0, s, n, h->in_x1.as<double>(), h->in_x2.as<double>(),
                       (const double *)nullptr, (const double *)nullptr, h->pts.as<double>(), (double *)nullptr);
    hipLaunchKernelGGL(k_box_unit, dim3(1), dim3(256), 0, s, n, h->pts.as<double>(), h->st.as<PairState>());
    hipLaunchKernelGGL(k_frag_unit, dim3((n + 16 + 255) / 256), dim3(256), 0, s, n, h->pts.as<double>(), h->rfrag.as<uint4>());
    RunParams rp;
    std::memset(&rp, 0, sizeof rp);
    rp.kind = kind; rp.batch = 1; rp.n_max = nn; rp.slot_stride = num_models; rp.mps = 4; rp.sample_sz = 3;
    hipLaunchKernelGGL(k_count_plan, dim3(1), dim3(PLAN_THREADS), 0, s, 1, h->st.as<PairState>(), h->model_count.as<int32_t>(), 2, CNT_WG_MODELS, h->cplan.as<int32_t>(),
                       (int32_t *)nullptr);
    const dim3 grid((unsigned)((num_models + CNT_WG_MODELS - 1) / CNT_WG_MODELS));
    MDRP_SWEEP_DISPATCH(k_count, kind, grid, dim3(CNT_THREADS), 0, s, rp, h->st.as<PairState>(), h->rfrag.as<uint4>(), h->models.as<Model>(),
                        h->tags.as<uint32_t>(), h->model_count.as<int32_t>(), h->cplan.as<int32_t>(),
                        h->tags_v.as<uint32_t>(), h->surv_count.as<int32_t>(), (unsigned long long *)nullptr, h->unit_f.as<int32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(candidates, h->unit_f.p, sizeof(int32_t) * num_models, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return MDRP_OK;
}

int mdrp_bound_models(mdrp_handle *h, int kind, const mdrp_model *models, int num_models, const double *x1, const double *x2,
                      int n, double sq_threshold, double *score_lb, int32_t *count_ub) {
    if (!h || num_models < 0 || n < 0 || kind < 0 || kind > 5 || !score_lb || !count_ub || !models) { g_err = "invalid argument"; return MDRP_ERR_INVALID; }
    if (num_models == 0) return MDRP_OK;
    MDRP_ENTER(h);
    hipStream_t s = h->stream;
    const int nn = std::max(n, 1);
    const size_t slots = (size_t)num_models;
    int rc;
    if ((rc = h->pts.ensure(sizeof(double) * PT_STRIDE * nn)) || (rc = h->st.ensure(sizeof(PairState))) ||
        (rc = h->models.ensure(sizeof(Model) * slots)) || (rc = h->slot_inl.ensure(sizeof(int32_t) * slots)) ||
        (rc = h->tags.ensure(sizeof(uint32_t) * slots)) || (rc = h->tags_v.ensure(sizeof(uint32_t) * slots)) ||
        (rc = h->surv_count.ensure(sizeof(int32_t))) || (rc = h->surv2_count.ensure(sizeof(int32_t))) ||
        (rc = h->cplan.ensure(2 * sizeof(int32_t))) || (rc = h->unit_a.ensure(sizeof(double) * slots)) ||
        (rc = h->unit_f.ensure(sizeof(int32_t) * slots)) || (rc = h->in_x1.ensure(sizeof(double) * 2 * nn)) ||
        (rc = h->in_x2.ensure(sizeof(double) * 2 * nn)))
        return rc;
    HIPCHK(hipMemcpyAsync(h->in_x1.p, x1, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->in_x2.p, x2, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->models.p, models, sizeof(Model) * num_models, hipMemcpyHostToDevice, s));
    std::vector<uint32_t> tags(num_models);
    for (int i = 0; i < num_models; ++i) tags[i] = (uint32_t)i;
    HIPCHK(hipMemcpyAsync(h->tags_v.p, tags.data(), sizeof(uint32_t) * num_models, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->surv_count.p, &num_models, sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(h->unit_a.p, 0, sizeof(double) * slots, s));
    HIPCHK(hipMemsetAsync(h->unit_f.p, 0, sizeof(int32_t) * slots, s));
    PairState ps;
    std::memset(&ps, 0, sizeof ps);
    ps.n = n; ps.active = 1; ps.sq_thr = sq_threshold; ps.eps = std::sqrt(sq_threshold);
    ps.best_min_score = DBL_MAX; // no records: nothing is retired, every model sees every correspondence
    HIPCHK(hipMemcpyAsync(h->st.p, &ps, sizeof ps, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pack_unit, dim3((n + 255) / 256 + 1), dim3(256), 0, s, n, h->in_x1.as<double>(), h->in_x2.as<double>(),
                       (const double *)nullptr, (const double *)nullptr, h->pts.as<double>(), (double *)nullptr);
    hipLaunchKernelGGL(k_box_unit, dim3(1), dim3(256), 0, s, n, h->pts.as<double>(), h->st.as<PairState>());
    RunParams rp;
    std::memset(&rp, 0, sizeof rp);
    rp.kind = kind; rp.batch = 1; rp.n_max = nn; rp.slot_stride = num_models; rp.mps = 4; rp.sample_sz = 3;
    hipLaunchKernelGGL(k_count_plan, dim3(1), dim3(PLAN_THREADS), 0, s, 1, h->st.as<PairState>(), h->surv_count.as<int32_t>(), 1, BND_THREADS,
                       h->cplan.as<int32_t>(), h->surv2_count.as<int32_t>());
    const dim3 grid((unsigned)((num_models + BND_THREADS - 1) / BND_THREADS));
    MDRP_SWEEP_DISPATCH(k_bound, kind, grid, dim3(BND_THREADS), 0, s, rp, h->st.as<PairState>(), h->pts.as<double>(), h->models.as<Model>(),
                        h->tags_v.as<uint32_t>(), h->surv_count.as<int32_t>(), h->cplan.as<int32_t>(),
                        h->tags.as<uint32_t>(), h->surv2_count.as<int32_t>(), (unsigned long long *)nullptr, h->unit_a.as<double>(), h->unit_f.as<int32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(score_lb, h->unit_a.p, sizeof(double) * num_models, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(count_ub, h->unit_f.p, sizeof(int32_t) * num_models, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return MDRP_OK;
}

int mdrp_refine_models(mdrp_handle *h, int kind, mdrp_model *models, int count, const double *x1, const double *x2,
                       const double *d1, const double *d2, int n, double scale_reproj, double weight_sampson,
                       const mdrp_bundle_opt *opt, int estimate_shift, double *final_cost) {
    if (!h || count < 0 || n < 0 || kind < 0 || kind > 5 || !opt || !models) { g_err = "invalid argument"; return MDRP_ERR_INVALID; }
    if (count == 0) return MDRP_OK;
    MDRP_ENTER(h);
    hipStream_t s = h->stream;
    int rc;
    const int nn = std::max(n, 1);
    if (kind >= MDRP_RELPOSE_5PT) {
        if ((rc = h->pts.ensure(sizeof(double) * PT_STRIDE * nn)) || (rc = h->in_x1.ensure(sizeof(double) * 2 * nn)) ||
            (rc = h->in_x2.ensure(sizeof(double) * 2 * nn)) || (rc = h->unit_e.ensure(sizeof(Model) * count)) ||
            (rc = h->unit_a.ensure(sizeof(double) * count)))
            return rc;
        HIPCHK(hipMemcpyAsync(h->in_x1.p, x1, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->in_x2.p, x2, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->unit_e.p, models, sizeof(Model) * count, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pack_unit, dim3((n + 255) / 256 + 1), dim3(256), 0, s, n, h->in_x1.as<double>(), h->in_x2.as<double>(),
                           (const double *)nullptr, (const double *)nullptr, h->pts.as<double>(), (double *)nullptr);
        LmOpt o;
        o.max_it = (int)std::min<uint64_t>(opt->max_iterations, 1u << 30); o.loss = opt->loss_type; o.loss_scale = opt->loss_scale;
        o.grad_tol = opt->gradient_tol; o.step_tol = opt->step_tol; o.lambda0 = opt->initial_lambda;
        o.lambda_min = opt->min_lambda; o.lambda_max = opt->max_lambda;
        MDRP_CLASSIC_LM_DISPATCH(kc_refine_unit, (count >= 2048 ? 64 : 256), kind, dim3(count), 0, s, count, h->unit_e.as<Model>(),
                                 h->pts.as<double>(), n, o, h->unit_a.as<double>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(models, h->unit_e.p, sizeof(Model) * count, hipMemcpyDeviceToHost, s));
        if (final_cost) HIPCHK(hipMemcpyAsync(final_cost, h->unit_a.p, sizeof(double) * count, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return MDRP_OK;
    }
    if ((rc = h->pts.ensure(sizeof(double) * PT_STRIDE * nn)) || (rc = h->dep.ensure(sizeof(double) * 2 * nn)) ||
        (rc = h->in_x1.ensure(sizeof(double) * 2 * nn)) || (rc = h->in_x2.ensure(sizeof(double) * 2 * nn)) ||
        (rc = h->in_d1.ensure(sizeof(double) * nn)) || (rc = h->in_d2.ensure(sizeof(double) * nn)) ||
        (rc = h->unit_e.ensure(sizeof(Model) * count)) || (rc = h->unit_a.ensure(sizeof(double) * count)))
        return rc;
    HIPCHK(hipMemcpyAsync(h->in_x1.p, x1, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->in_x2.p, x2, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->in_d1.p, d1, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->in_d2.p, d2, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->unit_e.p, models, sizeof(Model) * count, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pack_unit, dim3((n + 255) / 256 + 1), dim3(256), 0, s, n, h->in_x1.as<double>(), h->in_x2.as<double>(),
                       h->in_d1.as<double>(), h->in_d2.as<double>(), h->pts.as<double>(), h->dep.as<double>());
    LmOpt o;
    o.max_it = (int)std::min<uint64_t>(opt->max_iterations, 1u << 30); o.loss = opt->loss_type; o.loss_scale = opt->loss_scale;
    o.grad_tol = opt->gradient_tol; o.step_tol = opt->step_tol; o.lambda0 = opt->initial_lambda;
    o.lambda_min = opt->min_lambda; o.lambda_max = opt->max_lambda;
    MDRP_LM_DISPATCH(k_refine_unit, (count >= 2048 ? 64 : 256), kind, (kind == MDRP_CALIB && estimate_shift), dim3(count), lm_list_bytes(n), s,
                     count, h->unit_e.as<Model>(), h->pts.as<double>(), h->dep.as<double>(), n, scale_reproj, weight_sampson, o,
                     h->unit_a.as<double>(), lm_list_stride(n));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(models, h->unit_e.p, sizeof(Model) * count, hipMemcpyDeviceToHost, s));
    if (final_cost) HIPCHK(hipMemcpyAsync(final_cost, h->unit_a.p, sizeof(double) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return MDRP_OK;
}

} // extern "C"
