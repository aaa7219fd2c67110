// mdrp_kernels.h — gfx950 kernels of the RePoseD RANSAC hot path.
//
// The reference runs LO-RANSAC sequentially per image pair (ransac<> @0x22f030).  Its structure
// is exactly parallelisable: the sample sequence depends only on (seed, N); LO triggers depend only
// on the running records of the MINIMAL models; every LO starts from the triggering minimal model.  So per chunk
// of iterations:
//   k_samples  one wavefront per distinct N  splitmix64 sample table for the chunk (wave-speculative)        (a-3)
//   k_solve    one lane per minimal sample   solver -> <=4 models; compacted tag list                       (a-4..a-6')
//   k_count    one wavefront per 128 models  MFMA: candidate COUNT of every model over all correspondences (the Sampson
//                                            numerator is a [models x 9].[9 x correspondences] product); models that provably
//                                            cannot break a running record are retired here, the rest go on as survivors
//   k_sort_tags one workgroup per pair       survivors: dense / sparse class by candidate density, sparse counting-sorted by it
//   k_plan     one wavefront                 work items of the sweep (workgroups per pair and class)
//   k_score    one lane per hypothesis       Sampson/MSAC (+cheirality) sweep over all N correspondences,
//                                            correspondences staged through LDS, broadcast reads            (a-7)  HOT
//   k_scan     one wavefront per pair        ordered prefix scan of (count,score) records -> LO triggers     (a-2)
//   k_lo_plan  one wavefront                 this chunk's trigger range per pair, frozen (later scans only append)
//   k_lo       one wavefront (or workgroup)  LM refinement (<=25 it, TRUNCATED) + rescoring, per trigger     (a-8)
//   k_walk     one lane per pair             replays the reference's bookkeeping over the triggers,
//                                            dynamic stopping                                                (a-2)
//   k_final    one workgroup per pair        final LO, inlier mask, inlier-only LM, result record            (a-1, a-9)
// HBM layout (all fp64 unless noted): pts[B][n_max][6] = (x1.x, x1.y, x2.x, x2.y, 1/|(x1,1)|, 1/|(x2,1)|)
// normalised; dep[B][n_max][2] = (d1, d2); models[B][chunk][4] (96 B each); slot_score/slot_inl[B][chunk][4];
// tags[B][4*chunk] u32 compact list of live slots (k_solve); tags_sorted = the same, sparse class ordered by candidate
// density from the front, dense class from the back (k_sort_tags).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include "mdrp_logtab.h"
#include "mdrp_math.h"

// The library is built from several translation units compiled in parallel (mdrp_amd/build.py): mdrp_capi.hip (host API and most kernels) and
// mdrp_tu.hip compiled once per group of large kernel instantiations (mdrp_instances.h).  Kernel TEMPLATES are instantiated explicitly in
// exactly one unit (extern template elsewhere); the non-template kernels of these headers belong to the main unit and are `static`
// (unreferenced, dropped) in the others.
#ifdef MDRP_SECONDARY_TU
#define MDRP_GLOBAL static __global__
#else
#define MDRP_GLOBAL __global__
#endif

namespace mdrp {

constexpr int PT_STRIDE = 6;       // doubles per correspondence record
#define MDRP_TILE_PTS 512
#define MDRP_SCORE_MINWAVES 4
constexpr int TILE_PTS = MDRP_TILE_PTS; // correspondences per LDS tile (48 B each)
constexpr size_t SCORE_TILE_BYTES = (size_t)TILE_PTS * (PT_STRIDE * sizeof(double) + 4 * sizeof(float)); // + fp32 coordinates
#define MDRP_P1F_UNROLL 8
#define MDRP_P1_UNROLL 4
constexpr int PRUNE_EVERY = 4;     // bail-out test every PRUNE_EVERY groups of 32 records (power of two)
#define MDRP_DENSE_KEY 40 // of 64 probe records
#define MDRP_SOLVE_MINWAVES 2
#define MDRP_SCORE_THREADS 256
constexpr int SCORE_THREADS = MDRP_SCORE_THREADS; // waves of one workgroup share one LDS tile
constexpr int LM_THREADS = 256;
constexpr int MAX_NP = 9;
constexpr int MAX_ACC = MAX_NP * (MAX_NP + 1) / 2 + MAX_NP; // 54

struct PairState {
    int32_t n;          // correspondences (0 if the pair is degenerate: n < 3)
    int32_t table;      // sample table id (one per distinct n)
    int32_t active;     // still iterating
    int32_t n_triggers; // triggers found in the current super-chunk
    double eps;         // normalised max_epipolar_error
    double sq_thr;      // eps^2
    double scale_reproj;
    double lo_loss_scale;   // loss_scale of the LO refinement (eps; 1.0 for varying focal — reference quirk)
    double final_loss_scale; // focal estimators: user bundle loss_scale, normalised; calibrated: half the normalised epipolar threshold (k_prep)
    double norm;        // un-normalisation factor of the focals (1 for calibrated)
    double box[4];      // max |x1.x|, |x1.y|, |x2.x|, |x2.y| of the normalised correspondences (k_score's denominator bound)
    double cen[4];      // centroids subtracted by the 7-point baseline's normalisation (c1.x, c1.y, c2.x, c2.y); 0 otherwise
    uint64_t best_min_cnt;
    double best_min_score;
    uint64_t dyn_max_iter;
    uint64_t iterations, refinements, num_inliers;
    double inlier_ratio, model_score;
    Model best;
};

struct Trigger {
    uint32_t iter;   // iteration index inside the chunk
    int32_t k_ref;   // slot refined by LO (last record breaker of the iteration)
    int32_t k_min;   // slot that set a new best minimal score in this iteration, or -1
    int32_t cnt_min;
    double score_min;
    double ref_score; // filled by k_lo
    int32_t ref_cnt;
    int32_t cnt_ref; // inlier count of the minimal model k_ref (LO cost estimate: heavy problems are scheduled first)
    Model refined;
};

struct RunParams {
    int kind;           // MDRP_CALIB / SHARED / VARYING
    int solver;         // SOLVER_*
    int est_shift;
    int score_initial;  // RansacOptions::score_initial_model: the run starts from the state the reference is in after scoring (and
                        // LO-refining, to no effect) its reset initial model: records (0 inliers, N eps^2), one refinement
    int batch, n_max;
    int chunk_len;      // iterations in this chunk (one solve/score/scan launch train)
    int chunk_off;      // offset of this chunk inside its super-chunk
    int slot_stride;    // slots per pair in models/slot_*/tags: mps * (iterations a super-chunk can hold)
    int mps;            // model slots per sample: 4 (monodepth solvers, 7-point) or 12 (5-point: up to 10 poses)
    int sample_sz;      // 3 (monodepth), 5, 7: exponent of the inlier ratio in the dynamic stopping rule
    int super_len;      // iterations of the whole super-chunk (chunks that share one LO + walk pass)
    uint64_t chunk_start; // absolute iteration number of the super-chunk's first iteration
    uint64_t max_iterations, min_iterations;
    double dyn_mult, log_prob_missing;
    double weight_sampson;
    // user bundle options (final refinement)
    int final_max_it, final_loss;
    double grad_tol, step_tol, lambda0, lambda_min, lambda_max;
    int32_t *inl_stat;  // [2]: sum over the pairs of first_chunk_wish(inlier ratio of the result), and their number — the host sizes the NEXT call's first chunk from the mean
};

// ------------------------------------------------------------------------------------------------ reductions: cross-lane sums (gfx950)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// every lane of a row of 16 gets the row's sum (row_ror 8, 4, 2, 1)
__device__ __forceinline__ double row_sum16(double s) {
    s += dpp_f64<0x128>(s);
    s += dpp_f64<0x124>(s);
    s += dpp_f64<0x122>(s);
    s += dpp_f64<0x121>(s);
    return s;
}
typedef unsigned int lme_u2 __attribute__((ext_vector_type(2)));
// v_permlane32_swap: lanes 32-63 of a <-> lanes 0-31 of b;  v_permlane16_swap: odd rows (of 16 lanes) of a <-> even rows of b
__device__ __forceinline__ void swap32(double &a, double &b) {
    const lme_u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const lme_u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi.x, (int)lo.x); b = __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ void swap16(double &a, double &b) {
    const lme_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const lme_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi.x, (int)lo.x); b = __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ void swap32i(int &a, int &b) {
    const lme_u2 r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    a = (int)r.x; b = (int)r.y;
}
__device__ __forceinline__ void swap16i(int &a, int &b) {
    const lme_u2 r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
    a = (int)r.x; b = (int)r.y;
}
// sum over the wavefront, every lane gets it: two swap levels + the row rotations, no LDS round trips
__device__ __forceinline__ double wave_sum_swap(double v) {
    double b = v;
    swap32(v, b); v += b;
    b = v;
    swap16(v, b); v += b;
    return row_sum16(v);
}
__device__ __forceinline__ int wave_sum_swap_i(int v) {
    int b = v;
    swap32i(v, b); v += b;
    b = v;
    swap16i(v, b); v += b;
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);
    return v;
}
// Reduce-scatter of NA per-lane accumulators over the wavefront.  Level 1 pairs accumulators (2i, 2i+1): after the half swap one
// add leaves accumulator 2i in lanes 0-31 and 2i+1 in lanes 32-63; level 2 pairs those sums the same way over the rows of 16;
// the row rotations finish.  A tag travels through the same swaps, so the write-out index is whatever the hardware moved where.
template <int NA>
__device__ __forceinline__ void wave_reduce_scatter(const double *acc, double *out /*LDS or global, [NA]*/) {
    constexpr int M1 = (NA + 1) / 2, M2 = (M1 + 1) / 2;
    double w[M1];
    int t1[M1];
#pragma unroll
    for (int i = 0; i < M1; ++i) {
        double a = acc[2 * i], b = (2 * i + 1 < NA) ? acc[2 * i + 1] : 0.0;
        int ta = 2 * i, tb = (2 * i + 1 < NA) ? 2 * i + 1 : -1;
        swap32(a, b);
        swap32i(ta, tb);
        w[i] = a + b; t1[i] = ta;
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < M2; ++i) {
        double c = w[2 * i], d = (2 * i + 1 < M1) ? w[2 * i + 1] : 0.0;
        int tc = t1[2 * i], td = (2 * i + 1 < M1) ? t1[2 * i + 1] : -1;
        swap16(c, d);
        swap16i(tc, td);
        const double x = row_sum16(c + d);
        if ((lane & 15) == 0 && tc >= 0) out[tc] = x;
    }
}

// Wave sums: the xor butterfly (offsets 32, 16, 8, 4, 2, 1) written with the gfx950 half / quarter swaps and row rotations instead of
// six ds_bpermute round trips per value.  Bit-identical to the butterfly: every level adds the same two partial sums (in either
// order), so round 3's results are unchanged.
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_swap(v); }
__device__ __forceinline__ int wave_sum_i(int v) { return wave_sum_swap_i(v); }

// sums NV per-thread values over the workgroup; every thread gets the totals.  scratch: NWAVES*NV doubles of LDS.
// NV > 2: the wavefront part is a reduce-scatter (one add reduces two values per level: ~190 instead of ~630 instructions for 35
// sums); its totals land in LDS and are read back by every lane.  The summation tree is the butterfly's.
template <int NV, int NTHREADS>
__device__ __forceinline__ void block_sum(double *vals, double *scratch) {
    constexpr int NW = NTHREADS / 64;
    const int wave = threadIdx.x >> 6;
    if (NV <= 2) {
        if (NW == 1) {
#pragma unroll
            for (int i = 0; i < NV; ++i) vals[i] = wave_sum(vals[i]);
            return;
        }
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const double s = wave_sum(vals[i]);
            if (lane == 0) scratch[wave * NV + i] = s;
        }
    } else {
        if (NW == 1) __builtin_amdgcn_wave_barrier(); // (one wavefront: LDS writes of earlier uses are in program order)
        wave_reduce_scatter<NV>(vals, scratch + wave * NV);
        if (NW == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int i = 0; i < NV; ++i) vals[i] = scratch[i];
            return;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += scratch[w * NV + i];
        vals[i] = s;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ MFMA fragments
// A operand of v_mfma_f32_16x16x32_bf16 for 16 consecutive correspondences ("group"): lane l holds row l & 15 (the
// correspondence), K slots 8 (l >> 4) .. + 7, as 8 bf16 in one uint4.  K layout (mdrp_math.h, count_setup):
//   lanes  0..15: mh_0..7    lanes 16..31: ml_0..7    lanes 32..47: mh_0..7    lanes 48..63: 1, 1, 1, 0, 0, 0, 0, 0
__device__ __forceinline__ uint4 pack_bf16x8(const uint16_t v[8]) {
    return make_uint4((uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16),
                      (uint32_t)v[4] | ((uint32_t)v[5] << 16), (uint32_t)v[6] | ((uint32_t)v[7] << 16));
}
__device__ __forceinline__ void store_record_fragment(uint4 *__restrict__ frag, int i, double a, double b, double c, double d) {
    double m[8];
    count_monomials(a, b, c, d, m);
    uint4 *g = frag + (size_t)(i >> 4) * 64 + (i & 15);
    double mag = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) mag = fmax(mag, fabs(m[j])); // fmax drops NaN: test the sum too
    if (!(mag < 1e15) || !(m[0] + m[1] + m[3] + m[4] == m[0] + m[1] + m[3] + m[4])) { // non-finite or absurd coordinates: an all-zero row gives
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);                                   // C = 0, "never a definite outlier" (k_count's sign test must not see NaN)
        g[0] = z; g[16] = z; g[32] = z; g[48] = z;
        return;
    }
    uint16_t mh[8], ml[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bf16_split(m[j], mh[j], ml[j]);
    const uint4 hi = pack_bf16x8(mh);
    g[0] = hi; g[16] = pack_bf16x8(ml); g[32] = hi;
    g[48] = make_uint4(0x3F803F80u, 0x00003F80u, 0u, 0u);
}
__device__ __forceinline__ void clear_record_fragment(uint4 *__restrict__ frag, int i) {
    uint4 *g = frag + (size_t)(i >> 4) * 64 + (i & 15);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    g[0] = z; g[16] = z; g[32] = z; g[48] = z;
}

// ------------------------------------------------------------------------------------------------ prep
// One workgroup per pair: normalise the correspondences into the pts/dep records and set up the pair state.
// calibrated: Camera::unproject + thresholds * (1/f1 + 1/f2)/2 (estimate_monodepth_relative_pose @0x2242bf-0x224348)
// focal:      x / scale, scale = sum(|x1_i| + |x2_i|) / (sqrt2 N) (normalize_points @0x4f6ae0), thresholds / scale
struct CamDev { int32_t model_id, pad_; double p[4]; };

MDRP_GLOBAL __launch_bounds__(256) void k_prep(RunParams rp, const double *__restrict__ x1, const double *__restrict__ x2,
                                              const double *__restrict__ d1, const double *__restrict__ d2,
                                              const int32_t *__restrict__ n_per_pair, const int32_t *__restrict__ table_of_pair,
                                              const CamDev *__restrict__ cam1, const CamDev *__restrict__ cam2,
                                              double max_epi, double max_reproj, double bundle_loss_scale,
                                              double *__restrict__ pts, double *__restrict__ dep, PairState *__restrict__ st,
                                              uint4 *__restrict__ rfrag /*[pair][ceil(n_max/16)][64] MFMA A fragments, or null*/) {
    __shared__ double red[4];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int n = n_per_pair[pair];
    const size_t base = (size_t)pair * rp.n_max;
    double k = 1.0, norm = 1.0;
    double fx1 = 1, fy1 = 1, cx1 = 0, cy1 = 0, fx2 = 1, fy2 = 1, cx2 = 0, cy2 = 0;
    if (rp.kind == 0) {
        const CamDev a = cam1[pair], b = cam2[pair];
        if (a.model_id == 1) { fx1 = a.p[0]; fy1 = a.p[1]; cx1 = a.p[2]; cy1 = a.p[3]; } else { fx1 = fy1 = a.p[0]; cx1 = a.p[1]; cy1 = a.p[2]; }
        if (b.model_id == 1) { fx2 = b.p[0]; fy2 = b.p[1]; cx2 = b.p[2]; cy2 = b.p[3]; } else { fx2 = fy2 = b.p[0]; cx2 = b.p[1]; cy2 = b.p[2]; }
        k = 0.5 * (1.0 / (0.5 * (fx1 + fy1)) + 1.0 / (0.5 * (fx2 + fy2)));
    } else {
        double acc = 0;
        for (int i = tid; i < n; i += 256) {
            const double a = x1[2 * (base + i)], b = x1[2 * (base + i) + 1], c = x2[2 * (base + i)], d = x2[2 * (base + i) + 1];
            acc += sqrt(a * a + b * b) + sqrt(c * c + d * d);
        }
        acc = wave_sum(acc);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        norm = (red[0] + red[1] + red[2] + red[3]) / (1.4142135623730951 * (double)(n > 0 ? n : 1));
        k = 1.0 / norm;
    }
    double bx[4] = {0, 0, 0, 0};
    for (int i = tid; i < n; i += 256) {
        double a = x1[2 * (base + i)], b = x1[2 * (base + i) + 1], c = x2[2 * (base + i)], d = x2[2 * (base + i) + 1];
        if (rp.kind == 0) { a = (a - cx1) / fx1; b = (b - cy1) / fy1; c = (c - cx2) / fx2; d = (d - cy2) / fy2; }
        else { a /= norm; b /= norm; c /= norm; d /= norm; }
        bx[0] = fmax(bx[0], fabs(a)); bx[1] = fmax(bx[1], fabs(b)); bx[2] = fmax(bx[2], fabs(c)); bx[3] = fmax(bx[3], fabs(d));
        double *p = pts + (base + i) * PT_STRIDE;
        p[0] = a; p[1] = b; p[2] = c; p[3] = d;
        p[4] = 1.0 / sqrt(a * a + b * b + 1.0);
        p[5] = 1.0 / sqrt(c * c + d * d + 1.0);
        dep[2 * (base + i)] = d1[base + i];
        dep[2 * (base + i) + 1] = d2[base + i];
        if (rfrag) store_record_fragment(rfrag + (size_t)pair * ((rp.n_max + 15) / 16) * 64, i, a, b, c, d);
    }
    if (rfrag) { // rows past n in the last group: all-zero rows give C = 0, never a definite outlier
        const int g_end = ((n + 15) / 16) * 16;
        for (int i = n + tid; i < g_end; i += 256) clear_record_fragment(rfrag + (size_t)pair * ((rp.n_max + 15) / 16) * 64, i);
    }
    __shared__ double redbox[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = bx[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
        if ((tid & 63) == 0) redbox[tid >> 6][q] = v;
    }
    __syncthreads();
    if (tid == 0) {
        PairState s;
#pragma unroll
        for (int q = 0; q < 4; ++q) s.box[q] = fmax(fmax(redbox[0][q], redbox[1][q]), fmax(redbox[2][q], redbox[3][q]));
        s.n = n >= 3 ? n : 0;
        s.table = table_of_pair[pair];
        s.active = n >= 3;
        s.n_triggers = 0;
        s.eps = max_epi * k;
        s.sq_thr = s.eps * s.eps;
        const double rep = max_reproj * k;
        s.scale_reproj = rep > 0.0 ? (s.eps * s.eps) / (rep * rep) : 0.0;
        s.lo_loss_scale = rp.kind == 2 ? 1.0 : s.eps;
        // the focal wrappers divide the caller's BundleOptions::loss_scale by the normalisation scale; the calibrated wrapper OVERWRITES it
        // with half the normalised epipolar threshold, (1/f2 + 1/f1) * (max_epipolar_error * 0.25) (reference binary @0x224704; the
        // two readings coincide at the reference's own settings, max_epipolar_error 2 and loss_scale 1)
        s.final_loss_scale = rp.kind == 0 ? (1.0 / (0.5 * (fx2 + fy2)) + 1.0 / (0.5 * (fx1 + fy1))) * (max_epi * 0.25) : bundle_loss_scale * k;
        s.norm = norm;
        s.cen[0] = s.cen[1] = s.cen[2] = s.cen[3] = 0.0;
        s.best_min_cnt = 0; s.best_min_score = DBL_MAX;
        s.dyn_max_iter = rp.max_iterations;
        s.iterations = 0; s.refinements = 0; s.num_inliers = 0;
        s.inlier_ratio = 0.0; s.model_score = DBL_MAX;
        model_identity(s.best);
        if (rp.score_initial && n >= 3) { // the identity pose has E = 0: every r^2 is 0 / 0, nothing is an inlier
            s.best_min_score = s.sq_thr * (double)n;
            s.model_score = s.best_min_score;
            s.refinements = 1;
        }
        st[pair] = s;
    }
}

// ------------------------------------------------------------------------------------------------ samples
// One workgroup per distinct N.  The sample sequence is a pure function of (seed, N) (RandomSampler @0x4f8970): splitmix64
// is counter based (state after k raw draws = state + k*GAMMA), and a sample consumes K raw draws plus one per rejected
// duplicate (probability ~K^2/2N).  Thread i speculates that its sample starts K*i draws after the workgroup's state; threads up
// to and including the first one that saw a rejection are correct, the workgroup commits those and continues from that thread's
// end state.  ~2N/K^2 samples per rejection and 1024 threads: a 10k-sample table of triples (N = 2000) takes ~25 steps instead
// of 10k serial ones (0.18 ms with one wavefront per table, on the critical path of the run's first solver launch).
constexpr int SAMP_THREADS = 1024; // at most; the launch picks blockDim.x by sample size (samples_threads): a step commits what lies
                                    // before the first rejection (~2N/K^2 samples), more threads than that only lengthen the step
template <int K, class Draw>
__device__ __forceinline__ void samples_block(uint64_t n, uint64_t &state_io, int chunk_len, uint32_t *__restrict__ out, Draw &&draw) {
    __shared__ int s_first[SAMP_THREADS / 64];
    __shared__ unsigned long long s_state;
    const uint64_t GAMMA = 0x9e3779b97f4a7c15ULL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nwaves = nthreads >> 6;
    uint64_t state = state_io;
    int done = 0;
    while (done < chunk_len) {
        uint64_t s = state + (uint64_t)(K * tid) * GAMMA;
        const uint64_t s0 = s;
        uint32_t smp[K];
        draw(n, s, smp);
        const bool rejected = (s - s0) != (uint64_t)K * GAMMA;
        const unsigned long long ball = __ballot(rejected);
        if (lane == 0) s_first[wave] = ball ? wave * 64 + (__ffsll((long long)ball) - 1) : SAMP_THREADS;
        __syncthreads();
        int first = nthreads - 1;
        for (int w = nwaves - 1; w >= 0; --w) { const int v = s_first[w]; if (v < SAMP_THREADS) first = v; }
        const int nvalid = min(first + 1, chunk_len - done);
        if (tid < nvalid) {
#pragma unroll
            for (int k = 0; k < K; ++k) out[(size_t)K * (done + tid) + k] = smp[k];
        }
        if (tid == nvalid - 1) s_state = s;
        __syncthreads();
        state = s_state;
        done += nvalid;
    }
    state_io = state;
}
MDRP_GLOBAL __launch_bounds__(SAMP_THREADS) void k_samples(int n_tables, const int32_t *__restrict__ table_n, uint64_t *__restrict__ table_state,
                                                          int chunk_len, uint32_t *__restrict__ samples /*[n_tables][chunk_len][3]*/) {
    const int t = blockIdx.x;
    if (t >= n_tables) return;
    const uint64_t n = (uint64_t)table_n[t];
    if (n < 3) return;
    uint64_t state = table_state[t];
    __syncthreads(); // every thread holds the state before thread 0 advances it
    samples_block<3>(n, state, chunk_len, samples + (size_t)t * chunk_len * 3,
                     [](uint64_t n_, uint64_t &s_, uint32_t *o) { draw_sample3(n_, s_, o[0], o[1], o[2]); });
    if (threadIdx.x == 0) table_state[t] = state;
}

// ------------------------------------------------------------------------------------------------ Sampson terms
struct SampsonTerms { double C2, den; };

// d = a*b + c as the three-address VOP3 v_fma_f64.  hipcc (ROCm 7.2) otherwise selects the two-address
// v_fmac_f64 here and has to copy the loop-invariant addend (an entry of E) with v_mov_b64 before every use:
// 5 extra fp64-rate moves per Sampson evaluation (+25 % VALU work in the hot loop).
__device__ __forceinline__ double fma3(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return fma(a, b, c);
#endif
}

__device__ __forceinline__ SampsonTerms sampson_terms(const double E[9], double a, double b, double c, double d) {
    const double e0 = fma(E[0], a, fma3(E[1], b, E[2]));
    const double e1 = fma(E[3], a, fma3(E[4], b, E[5]));
    const double e2 = fma(E[6], a, fma3(E[7], b, E[8]));
    const double g0 = fma(E[0], c, fma3(E[3], d, E[6]));
    const double g1 = fma(E[1], c, fma3(E[4], d, E[7]));
    const double C = fma(c, e0, fma(d, e1, e2));
    SampsonTerms r;
    r.den = fma(e0, e0, fma(e1, e1, fma(g0, g0, g1 * g1)));
    r.C2 = C * C;
    return r;
}

// Conservative fp32 phase-1 filter of the sweep for one hypothesis: mdrp_math.h filter_setup / filter_keeps
__device__ __forceinline__ void bound_setup(const double E[9], const PairState &ps, double thr, float Ef[9], float &tb, double &thr_dmax) {
    filter_setup(E, ps.box, thr, Ef, tb, thr_dmax);
}

// fp32 record layout of the phase-1 filter: two records per 32 B, component-major (a0 a1 b0 b1)(c0 c1 d0 d1)
__device__ __forceinline__ void store_rec32(float4 *__restrict__ recs32, int i, double a, double b, double c, double d) {
    float *q = reinterpret_cast<float *>(recs32) + (i >> 1) * 8 + (i & 1);
    q[0] = (float)a; q[2] = (float)b; q[4] = (float)c; q[6] = (float)d;
}

constexpr int PROBE_PTS = 64; // scale of the candidate-density key: key = candidates per 64 correspondences (k_count)

// ------------------------------------------------------------------------------------------------ solve
// One lane per minimal sample.  Models go to models[pair][iter][k]; live slots are appended to the pair's tag list
// with ONE atomic per wave (wave-aggregated prefix sum); model_count[2 * pair] counts them.
// Nothing in the kernel is shared between wavefronts, so the workgroup size only decides how slots are refilled: with one wavefront per workgroup a
// finished wavefront is replaced at once instead of when the slowest of four has finished (the solvers' run time varies with the number of
// roots): the kernel alone is 25 % shorter (P3P 1.66 -> 1.25 ms, shared focal 1.52 -> 1.30, shift 1.85 -> 1.30).  But the long solver launch runs
// BESIDE the first chunk's exact sweep, and the faster it takes the chip's slots the later that sweep's lane-per-hypothesis loops (0.5 ms of latency
// that nothing shortens) get going: A/B on one box, three runs each — shared focal 8.05-8.27 -> 7.82-7.95 ms per step, shift solver 8.72-8.86 ->
// 8.81-8.89, calibrated P3P 8.33-8.49 -> 8.49-8.74 (its first sweep then ends 0.6 ms after the solver instead of inside it; stream priorities
// change nothing).  So: one wavefront per workgroup for the focal solvers, four for the calibrated ones.
constexpr int solve_threads(int solver) { return solver == SOLVER_P3P || solver == SOLVER_SHIFT ? 256 : 64; }
template <int SOLVER> // one solver per kernel (host dispatch): a runtime switch made every launch carry the registers of the largest
__global__ __launch_bounds__(solve_threads(SOLVER), MDRP_SOLVE_MINWAVES) void k_solve(RunParams rp, const PairState *__restrict__ st, const uint32_t *__restrict__ samples,
                                               const double *__restrict__ pts, const double *__restrict__ dep,
                                               Model *__restrict__ models, int32_t *__restrict__ slot_inl,
                                               uint32_t *__restrict__ tags, int32_t *__restrict__ model_count,
                                               int it_begin, int it_end /*iterations [it_begin, it_end) of the chunk: one launch solves a sub-range*/) {
    const int pair = blockIdx.y;
    const int it = it_begin + blockIdx.x * solve_threads(SOLVER) + threadIdx.x;
    const PairState &ps = st[pair];
    if (!ps.active) return;
    const bool live = it < it_end;
    int n = 0;
    bool nan_model = false;
    Model out[4];
    auto load_sample = [&](Sample3 &s) {
        const uint32_t *sm = samples + ((size_t)ps.table * rp.chunk_len + it) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t idx = (size_t)pair * rp.n_max + sm[k];
            const double *p = pts + idx * PT_STRIDE;
            s.x1[k][0] = p[0]; s.x1[k][1] = p[1]; s.x2[k][0] = p[2]; s.x2[k][1] = p[3];
            s.d1[k] = dep[2 * idx]; s.d2[k] = dep[2 * idx + 1];
        }
    };
    if (live) {
        Sample3 s;
        load_sample(s);
        n = run_solver(SOLVER, s, out);
        // The reference's P3P emits NaN poses for ~3 % of the samples (p3p_reference_nan: exactly predictable).  A NaN hypothesis scores N * thr with
        // no inlier: it can only be a record while nothing has been scored yet — then it costs the reference one LO, and it is the run's answer if no
        // sample ever gives a real pose.  It needs no sweep: slot state -3 tells k_scan its (count, score) = (0, N * thr); it is not on the tag list.
    }
    if (SOLVER == SOLVER_P3P) {
        // ... and it only matters BEFORE the first real pose of the run: a lane asks only if no earlier iteration of its own wavefront has one
        // (k_scan decides exactly; this merely keeps ~80 % of the wavefronts — those whose first sample has a pose — out of the predicate)
        const unsigned long long valid = __ballot(live && n > 0);
        const int first_valid = valid ? __ffsll((long long)valid) - 1 : 64;
        if (live && (int)(threadIdx.x & 63) < first_valid) {
            Sample3 s; // read again (L2): kept live across the solver it would cost the kernel its third wavefront per SIMD
            load_sample(s);
            nan_model = p3p_reference_nan(s);
        }
    }
    // wave-aggregated append to the pair's tag list: one atomic per wave (k_sort_tags orders and classifies the list)
    const int lane = threadIdx.x & 63;
    int pre = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(pre, o, 64);
        if (lane >= o) pre += v;
    }
    const int tot = __shfl(pre, 63, 64);
    int base = 0;
    if (lane == 63 && tot > 0) base = atomicAdd(&model_count[2 * pair], tot);
    base = __shfl(base, 63, 64);
    if (!live) return;
    const size_t slot0 = (size_t)pair * rp.slot_stride + (size_t)(rp.chunk_off + it) * 4;
    int pos = base + pre - n;
    const size_t tag_base = (size_t)pair * rp.slot_stride;
    // slot states of the iteration in one 16-byte store: -1 = empty, -2 = "no record" — the default of every live slot:
    // k_count / k_bound retire most hypotheses without touching their slots again, k_score overwrites the survivors'
    *reinterpret_cast<int4 *>(slot_inl + slot0) = make_int4(nan_model ? -3 : (n > 0 ? -2 : -1), n > 1 ? -2 : -1, n > 2 ? -2 : -1, n > 3 ? -2 : -1);
    if (nan_model) {
        Model m;
        model_identity(m);
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        m.q[0] = m.q[1] = m.q[2] = m.q[3] = m.t[0] = m.t[1] = m.t[2] = m.scale = qnan;
        models[slot0] = m;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < n) {
            models[slot0 + k] = out[k];
            tags[tag_base + pos] = (uint32_t)((rp.chunk_off + it) * 4 + k);
            ++pos;
        }
    }
}

// ------------------------------------------------------------------------------------------------ score (HOT)
// One lane per hypothesis; the pair's correspondences are staged through LDS in 512-record tiles (24 KiB) and read as
// wave-wide broadcasts (every lane reads the same record), so a tile is fetched from HBM/L2 once per workgroup.
// Algorithmic bytes: 32 B per (model x correspondence) evaluation (x1, x2 as four fp64 — what the CPU loop reads,
// SURVEY.md §8d).  The kernel is VALU-issue bound (88 % VALU-busy, PMC), so everything below is about issuing fewer
// (and cheaper) instructions per evaluation without changing a single result:
//   * k_sort_tags classifies hypotheses by their candidate density on the pair's first 64 records and sorts the SPARSE
//     ones (garbage, > 90 %) by it; DENSE ones (> 40 of 64 probe records survive phase 1) take the coherent single pass.
//   * SPARSE, per 64 records:  phase 1 is branch-free, needs only the numerator C = x2' E x1 and runs in packed fp32 on an
//     fp32 copy of the tile (score_tile_f32: conservative threshold, two records per v_pk_fma_f32); survivors (a superset
//     of the true candidates, ~5 %) set a bit in a per-lane mask.  phase 2: every lane pops ITS OWN bits (per-lane LDS
//     addresses) and runs the exact fp64 test, quotient, check_cheirality and accumulation in record order; its trip count
//     is the largest candidate count of any lane of the wavefront — hence the sort: similar hypotheses share a workgroup.
//   * DENSE: candidate sets of the 64 lanes nearly coincide (the true inliers), so the plain per-record branch is coherent:
//     one pass, no recomputation.
//   * bail-out against the records of earlier chunks (struct Prune): exact, skips ~30 % of the sparse work at 50 % outliers
//     and nearly all of it on clean data.
//   * v_fma_f64 with a loop-invariant addend goes through inline asm (fma3): hipcc picks v_fmac_f64 + v_mov_b64 otherwise.
// exact inlier test + accumulation for one record (compute_sampson_msac_score @0x4f61d0 body, check_cheirality @0x1dce00)
template <bool POSE>
__device__ __forceinline__ void score_point(const double *__restrict__ rec, const double E[9], const double R[9], const double t[3],
                                            double thr, double &score, int &cnt) {
    const double2 *P = reinterpret_cast<const double2 *>(rec);
    const double2 p01 = P[0], p23 = P[1];
    const double a = p01.x, b = p01.y, c = p23.x, d = p23.y;
    const SampsonTerms s = sampson_terms(E, a, b, c, d);
    if (!(s.C2 < thr * (1.0 + 1e-12) * s.den)) return; // not even a candidate (phase 1 may hand over a superset)
    const double r2 = s.C2 / s.den;
    if (r2 < thr) {
        bool ok = true;
        if (POSE) { // unit bearings via the precomputed inverse norms, min depth 0.01
            const double2 p45 = P[2];
            const double u0 = fma(R[0], a, fma(R[1], b, R[2]));
            const double u1 = fma(R[3], a, fma(R[4], b, R[5]));
            const double u2 = fma(R[6], a, fma(R[7], b, R[8]));
            const double uh = fma(u0, c, fma(u1, d, u2));
            const double ut = fma(u0, t[0], fma(u1, t[1], u2 * t[2]));
            const double ht = fma(c, t[0], fma(d, t[1], t[2]));
            const double A = -uh * p45.x * p45.y;
            const double b1 = -ut * p45.x, b2 = ht * p45.y;
            const double l1 = fma(-A, b2, b1), l2 = fma(-A, b1, b2);
            const double md = 0.01 * fma(-A, A, 1.0);
            ok = (l1 > md) && (l2 > md);
        }
        if (ok) { score += r2; ++cnt; }
    }
}

// `recs` may be an LDS tile (broadcast ds_reads) or the pair's records in global memory (wave-uniform addresses ->
// scalar loads into SGPRs, which v_fma_f64 takes directly as an operand).  The pose (R,t) is only needed by phase 2,
// so it is rebuilt from the model's quaternion when a group has candidates instead of living in 24 VGPRs.
// Bail-out against the pair's records from EARLIER chunks (exact): a hypothesis only matters if it beats a running
// record of the minimal models (score_models<> @0x22ebc0: more inliers OR better score than any earlier minimal model).
// Records only improve over a run, so the records at the end of the previous chunk are a valid (weaker) bar for every
// model of this chunk.  After `processed` records a lane is provably irrelevant when
//     cnt + (n - processed) <= rec_cnt      (cannot end with more inliers)      and
//     score + thr (processed - cnt) >= rec_score   (all remaining terms are >= 0: cannot end with a better score).
// Its slot is then written as "not a record" (count -2); k_scan treats it like an empty slot, so the trajectory is
// identical to scoring everything.  A wavefront stops computing once all of its lanes are out.
struct Prune {
    long long rec_cnt;
    double rec_score; // already inflated by 1e-12 relative; DBL_MAX disables pruning
    int n;            // correspondences of the pair
    int processed;    // records consumed so far (all tiles)
    bool dead;        // this lane is out
    bool wave_dead;
};

// Dense hypotheses (all 64 lanes close to the true model): their candidate sets nearly coincide (the true inliers), so
// the plain per-record branch is coherent across the wavefront and the Sampson terms need not be recomputed in a second
// phase: one pass, records as wave-wide broadcasts, inlier work under a branch most lanes take together.
template <bool POSE>
__device__ __forceinline__ void score_tile_dense(const double *__restrict__ recs, int npts, const double E[9], const Model *__restrict__ mp,
                                                 double thr, double &score, int &cnt, Prune &pr) {
    const double thr_hi = thr * (1.0 + 1e-12);
    double R[9], t[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = 0;
    if (POSE) {
        double q[4];
        q[0] = mp->q[0]; q[1] = mp->q[1]; q[2] = mp->q[2]; q[3] = mp->q[3];
        t[0] = mp->t[0]; t[1] = mp->t[1]; t[2] = mp->t[2];
        quat_to_R(q, R);
    }
    for (int p0 = 0; p0 < npts; p0 += 32 * PRUNE_EVERY) {
        const int g = min(32 * PRUNE_EVERY, npts - p0);
        if (!pr.dead) {
#pragma unroll 2
            for (int j = 0; j < g; ++j) {
                const double2 *P = reinterpret_cast<const double2 *>(recs + (size_t)(p0 + j) * PT_STRIDE);
                const double2 p01 = P[0], p23 = P[1];
                const double a = p01.x, b = p01.y, c = p23.x, d = p23.y;
                const SampsonTerms s = sampson_terms(E, a, b, c, d);
                if (s.C2 < thr_hi * s.den) {
                    const double r2 = s.C2 / s.den;
                    if (r2 < thr) {
                        bool ok = true;
                        if (POSE) {
                            const double2 p45 = P[2];
                            const double u0 = fma(R[0], a, fma(R[1], b, R[2]));
                            const double u1 = fma(R[3], a, fma(R[4], b, R[5]));
                            const double u2 = fma(R[6], a, fma(R[7], b, R[8]));
                            const double uh = fma(u0, c, fma(u1, d, u2));
                            const double ut = fma(u0, t[0], fma(u1, t[1], u2 * t[2]));
                            const double ht = fma(c, t[0], fma(d, t[1], t[2]));
                            const double A = -uh * p45.x * p45.y;
                            const double b1 = -ut * p45.x, b2 = ht * p45.y;
                            const double l1 = fma(-A, b2, b1), l2 = fma(-A, b1, b2);
                            const double md = 0.01 * fma(-A, A, 1.0);
                            ok = (l1 > md) && (l2 > md);
                        }
                        if (ok) { score += r2; ++cnt; }
                    }
                }
            }
        }
        pr.processed += g;
        if (pr.rec_score < DBL_MAX) {
            pr.dead = pr.dead || (((long long)cnt + (long long)(pr.n - pr.processed) <= pr.rec_cnt) &&
                                  (score + thr * (double)(pr.processed - cnt) >= pr.rec_score));
            if (__all(pr.dead)) { pr.wave_dead = true; return; }
        }
    }
}

// numerator C = x2' E x1 only (8 FMA); true when C^2 < thr * Dmax, i.e. the record MAY be an inlier.  12 ops instead of 22.
// (Keeping the exact (E x1) half of the denominator and bounding only the (E' x2) half measured slower: +3 ops per
// evaluation cost more than the fewer false candidates saved.)
__device__ __forceinline__ bool sampson_maybe(const double E[9], double a, double b, double c, double d, double thr_hi, double dmax) {
    const double e0 = fma(E[0], a, fma3(E[1], b, E[2]));
    const double e1 = fma(E[3], a, fma3(E[4], b, E[5]));
    const double e2 = fma(E[6], a, fma3(E[7], b, E[8]));
    const double C = fma(c, e0, fma(d, e1, e2));
    return C * C < thr_hi * dmax;
}

// BOUND (sparse hypotheses): phase 1 compares the numerator against thr * Dmax, where Dmax >= den for every record of
// the pair (per-hypothesis bound over the pair's coordinate box).  C^2 >= thr * Dmax >= thr * den proves an outlier
// with 12 instead of 22 ops; the few records that survive (a superset of the true candidates, ~1 %) get the exact test
// in phase 2.  Dense hypotheses keep the exact denominator in phase 1 (the bound would send most records to phase 2).
template <bool POSE, bool BOUND>
__device__ __forceinline__ void score_tile(const double *__restrict__ recs, int npts, const double E[9], const Model *__restrict__ mp,
                                           double thr, double thr_dmax, double &score, int &cnt, Prune &pr) {
    const double thr_hi = thr * (1.0 + 1e-12);
    for (int p0 = 0; p0 < npts; p0 += 32) {
        const int g = min(32, npts - p0);
        const double *base = recs + (size_t)p0 * PT_STRIDE;
        uint32_t mask = 0;
        if (g == 32) {
#pragma unroll 1
            for (int j0 = 0; j0 < 32; j0 += MDRP_P1_UNROLL) {
#pragma unroll
                for (int jj = 0; jj < MDRP_P1_UNROLL; ++jj) {
                    const int j = j0 + jj;
                    const double2 *P = reinterpret_cast<const double2 *>(base + j * PT_STRIDE);
                    const double2 p01 = P[0], p23 = P[1];
                    if (BOUND) {
                        mask |= sampson_maybe(E, p01.x, p01.y, p23.x, p23.y, thr_hi, thr_dmax) ? (1u << j) : 0u;
                    } else {
                        const SampsonTerms s = sampson_terms(E, p01.x, p01.y, p23.x, p23.y);
                        mask |= (s.C2 < thr_hi * s.den) ? (1u << j) : 0u;
                    }
                }
            }
        } else {
            for (int j = 0; j < g; ++j) {
                const double2 *P = reinterpret_cast<const double2 *>(base + j * PT_STRIDE);
                const double2 p01 = P[0], p23 = P[1];
                if (BOUND) {
                    mask |= sampson_maybe(E, p01.x, p01.y, p23.x, p23.y, thr_hi, thr_dmax) ? (1u << j) : 0u;
                } else {
                    const SampsonTerms s = sampson_terms(E, p01.x, p01.y, p23.x, p23.y);
                    mask |= (s.C2 < thr_hi * s.den) ? (1u << j) : 0u;
                }
            }
        }
        if (pr.dead) mask = 0;
        if (mask) {
            double R[9], t[3];
            if (POSE) {
                double q[4];
                q[0] = mp->q[0]; q[1] = mp->q[1]; q[2] = mp->q[2]; q[3] = mp->q[3];
                t[0] = mp->t[0]; t[1] = mp->t[1]; t[2] = mp->t[2];
                quat_to_R(q, R);
            }
            while (mask) { // per-lane candidates, ascending record order (same accumulation order as the CPU loop)
                const int j = __ffs(mask) - 1;
                mask &= mask - 1;
                score_point<POSE>(base + j * PT_STRIDE, E, R, t, thr, score, cnt);
            }
        }
        pr.processed += g;
        if (pr.rec_score < DBL_MAX && ((p0 >> 5) & (PRUNE_EVERY - 1)) == PRUNE_EVERY - 1) {
            pr.dead = pr.dead || (((long long)cnt + (long long)(pr.n - pr.processed) <= pr.rec_cnt) &&
                                  (score + thr * (double)(pr.processed - cnt) >= pr.rec_score));
            if (__all(pr.dead)) { pr.wave_dead = true; return; }
        }
    }
}

// Sparse hypotheses, phase 1 in fp32.  The phase-1 test only has to be CONSERVATIVE (keep a superset of the records the
// exact fp64 test accepts), so it runs at the fp32 VALU rate (2x fp64) on an fp32 copy of the coordinates:
//   exact:  C^2 < thr_hi * den   =>   |C| < T := sqrt(thr_hi * Dmax)
//   fp32:   |C32 - C| <= 7u * M,  u = 2^-24,  M = sum of |E_ij| |x2_i|max |x1_j|max   (13 input + 8 FMA roundings, depth 7)
//   keep the record unless |C32| > tb,  tb >= T + 2e-6 * M  (5x margin on 7u = 4.2e-7); NaN keeps; tb = inf keeps all.
// Survivors (~1-2 % of the records) are re-tested and scored exactly in fp64 by phase 2, so results are bit-identical.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool POSE>
__device__ __forceinline__ void score_tile_f32(const double *__restrict__ recs, const float4 *__restrict__ recs32, int npts,
                                               const double E[9], const float Ef[9], float tb, const Model *__restrict__ mp,
                                               double thr, double &score, int &cnt, Prune &pr) {
    // recs32: two records per 32 B, component-major: (a0 a1 b0 b1)(c0 c1 d0 d1) -> v_pk_fma_f32 without shuffles.
    // Phase 1 fills the candidate mask of a 64-record window before phase 2 runs: phase 2 is a per-lane loop, the
    // wavefront pays the MAXIMUM candidate count over its lanes, and max / mean falls with the window length (windows of
    // 128 / 256 records measured equal / slower: the multi-word bit bookkeeping eats the gain).
    constexpr int WIN = 64;
    static_assert(TILE_PTS % WIN == 0, "window must divide the tile");
    f32x2 Ev[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Ev[i] = (f32x2)(Ef[i]);
    for (int p0 = 0; p0 < npts; p0 += WIN) {
        const int g = min(WIN, npts - p0);
        const double *base = recs + (size_t)p0 * PT_STRIDE;
        uint32_t half[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q0 = 32 * h;
            if (q0 < g) { // wave-uniform
                const float4 *b32 = recs32 + p0 + q0;
                uint32_t mask = 0;
#pragma unroll 1
                for (int j0 = 0; j0 < 32; j0 += MDRP_P1F_UNROLL) {
#pragma unroll
                    for (int jj = 0; jj < MDRP_P1F_UNROLL; jj += 2) {
                        const int j = j0 + jj;
                        const float4 ab = b32[j], cd = b32[j + 1];
                        const f32x2 a = {ab.x, ab.y}, b = {ab.z, ab.w}, c = {cd.x, cd.y}, d = {cd.z, cd.w};
                        const f32x2 e0 = __builtin_elementwise_fma(Ev[0], a, __builtin_elementwise_fma(Ev[1], b, Ev[2]));
                        const f32x2 e1 = __builtin_elementwise_fma(Ev[3], a, __builtin_elementwise_fma(Ev[4], b, Ev[5]));
                        const f32x2 e2 = __builtin_elementwise_fma(Ev[6], a, __builtin_elementwise_fma(Ev[7], b, Ev[8]));
                        const f32x2 C = __builtin_elementwise_fma(c, e0, __builtin_elementwise_fma(d, e1, e2));
                        // mask = 2 * mask + keep: compare into an SGPR pair, add-with-carry shifts it in (2 instructions per
                        // record instead of compare + select + constant move + or); record j lands in bit 31 - j
                        const float cx = C.x, cy = C.y;
                        unsigned long long kx, ky, co;
                        asm("v_cmp_ngt_f32_e64 %0, |%1|, %2" : "=s"(kx) : "v"(cx), "v"(tb));
                        asm("v_cmp_ngt_f32_e64 %0, |%1|, %2" : "=s"(ky) : "v"(cy), "v"(tb));
                        asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(mask), "=s"(co) : "s"(kx));
                        asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(mask), "=s"(co) : "s"(ky));
                    }
                }
                mask = __brev(mask); // back to record j in bit j
                const int valid = g - q0; // records past the end of the tile hold stale LDS
                if (valid < 32) mask &= (1u << valid) - 1u;
                half[h] = mask;
            }
        }
        uint64_t m = (uint64_t)half[0] | ((uint64_t)half[1] << 32);
        if (pr.dead) m = 0;
        if (m) {
            double R[9], t[3];
            if (POSE) {
                double q[4];
                q[0] = mp->q[0]; q[1] = mp->q[1]; q[2] = mp->q[2]; q[3] = mp->q[3];
                t[0] = mp->t[0]; t[1] = mp->t[1]; t[2] = mp->t[2];
                quat_to_R(q, R);
            }
            while (m) { // per-lane candidates, ascending record order (same accumulation order as the CPU loop)
                const int j = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                score_point<POSE>(base + j * PT_STRIDE, E, R, t, thr, score, cnt);
            }
        }
        pr.processed += g;
        if (pr.rec_score < DBL_MAX) {
            pr.dead = pr.dead || (((long long)cnt + (long long)(pr.n - pr.processed) <= pr.rec_cnt) &&
                                  (score + thr * (double)(pr.processed - cnt) >= pr.rec_score));
            if (__all(pr.dead)) { pr.wave_dead = true; return; }
        }
    }
}

// largest p with prefix[p] <= w   (prefix non-decreasing, prefix[0] = 0, w < prefix[batch])
__device__ __forceinline__ int plan_find(const int32_t *__restrict__ prefix, int batch, int w) {
    int lo = 0, hi = batch;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= w) lo = mid; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------ count (MFMA)
// Candidate count of EVERY hypothesis over ALL correspondences of its pair, on the matrix cores, and the retirement of the
// hypotheses that provably cannot matter.
//   * A hypothesis matters only if it beats a running record of the minimal models (score_models<> @0x22ebc0: more inliers
//     OR a better score than every earlier minimal model).  With `cand` >= its true inlier count (conservative filter,
//     mdrp_math.h count_setup):  count <= cand, and score = sum of r^2 over inliers + thr (N - count) >= thr (N - cand).
//     So  cand <= rec_cnt  and  thr (N - cand) >= rec_score  prove it irrelevant, where (rec_cnt, rec_score) are the records
//     at the end of the previous chunk (records only improve, so they are a valid bar for every later model).  Its slot
//     is marked "no record" (-2) exactly like a hypothesis the sweep's Prune retires; k_scan skips it.  Garbage hypotheses
//     (> 90 % at 50 % outliers: ~5 % of the correspondences survive the filter) never reach the fp64 sweep at all.
//   * Survivors are appended to the pair's survivor list with their candidate density (per 64 correspondences) in the top
//     byte: k_sort_tags classifies and sorts them for k_score, which computes their exact scores.
//   * In the first chunk of a run there is no record yet: everything survives and this kernel only provides the densities.
// Work split: a wavefront owns 128 hypotheses = 8 MFMA tiles of 16 (B operand, in registers for the whole sweep) and streams
// the pair's correspondences as prebuilt A fragments (k_prep; 1 KiB per 16 correspondences, coalesced 16 B per lane, L2
// resident: every wavefront of the pair reads the same 64 N bytes).  Output tile: lane l holds hypothesis l & 15 against
// correspondences 4 (l >> 4) .. + 3 of the group, so the four |C| > tb tests of a lane belong to ONE hypothesis and add into
// one counter per tile.  Per 16 x 16 evaluations: one MFMA (16 cycles) + 4 x (v_cmp + v_addc).
#define MDRP_CNT_TILES 8
constexpr int CNT_TILES = MDRP_CNT_TILES;          // MFMA tiles (16 hypotheses) per wavefront: 4 or 8
constexpr int CNT_ROUNDS = (16 * CNT_TILES) / 64;  // prologue / epilogue rounds of 64 hypotheses
static_assert(CNT_TILES == 4 || CNT_TILES == 8, "a wavefront owns 64 or 128 hypotheses");
constexpr int CNT_WAVE_MODELS = 16 * CNT_TILES;    // 128
constexpr int CNT_THREADS = 256;
constexpr int CNT_WG_MODELS = CNT_WAVE_MODELS * (CNT_THREADS / 64);
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// plan[0..B] = prefix sum of workgroups per pair (ceil(counts[stride * pair] / per_wg)); one wavefront
// The plan kernels are prefix sums over the pairs of the batch, each between two sweeps on the critical path of the step: one
// 1024-thread workgroup issues all its (dependent, ~2 us) loads at once where one wavefront walked the batch in 16 steps.
constexpr int PLAN_THREADS = 1024;
// exclusive prefix of v over the workgroup (PLAN_THREADS threads); total = sum.  s_w: PLAN_THREADS / 64 + 1 ints of LDS.
__device__ __forceinline__ int plan_block_scan(int v, int &total, int *s_w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    __syncthreads(); // s_w of the previous call is no longer read
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const int x = lane < PLAN_THREADS / 64 ? s_w[lane] : 0;
        int xi = x;
#pragma unroll
        for (int o = 1; o < PLAN_THREADS / 64; o <<= 1) { const int u = __shfl_up(xi, o, 64); if (lane >= o) xi += u; }
        if (lane < PLAN_THREADS / 64) s_w[lane] = xi - x;
        if (lane == PLAN_THREADS / 64 - 1) s_w[PLAN_THREADS / 64] = xi;
    }
    __syncthreads();
    total = s_w[PLAN_THREADS / 64];
    return inc - v + s_w[wave];
}
// zero_a / zero_b: per-pair counters the next sweep accumulates into (or null) - cleared here instead of by a memset node in front
// begin (or null): the plan covers list entries [begin[stride p], counts[stride p]) of every pair — a sub-range of a list that is still growing
MDRP_GLOBAL __launch_bounds__(PLAN_THREADS) void k_count_plan(int batch, const PairState *__restrict__ st, const int32_t *__restrict__ counts, int stride,
                                                             int per_wg, int32_t *__restrict__ plan, int32_t *__restrict__ zero_a,
                                                             const int32_t *__restrict__ begin = nullptr, int32_t *__restrict__ zero_b2 = nullptr /*[2 * batch], stride 2*/) {
    __shared__ int s_w[PLAN_THREADS / 64 + 1];
    int run = 0;
    for (int p0 = 0; p0 < batch; p0 += PLAN_THREADS) {
        const int p = p0 + threadIdx.x;
        const int b = (p < batch && st[p].active) ? (counts[stride * p] - (begin ? begin[stride * p] : 0) + per_wg - 1) / per_wg : 0;
        int tot;
        const int ex = plan_block_scan(b, tot, s_w);
        if (p < batch) { plan[p] = run + ex; if (zero_a) zero_a[p] = 0; if (zero_b2) zero_b2[2 * p] = 0; }
        run += tot;
    }
    if (threadIdx.x == 0) plan[batch] = run;
}

// Two-phase count (round 6).  A garbage hypothesis keeps ~5 % of the correspondences as candidates; with cand_A candidates among the first R_A
// records its total is at most cand_A + (N - R_A), and once THAT bound passes both record tests the hypothesis is retired without its remaining
// N - R_A evaluations.  Phase A sweeps the pair's first T_A tiles (256 records each) for every hypothesis and retires what is decided; the few
// that are not (true candidates, partial fits: ~8 % at 50 % outliers) go on an `undecided` list with their partial counts, and phase B — a second
// launch over that list only — sweeps the remaining tiles and applies the unchanged test to the exact total.  T_A is a per-pair function of the
// pair state alone (count_split_tiles: the point where a hypothesis with 1/16 candidates is decided), so both launches agree on it; any T_A gives
// the same decisions for every hypothesis that reaches the final test, and the early ones are implied by it (cand <= cand_A + N - R_A).
// Pairs whose records do not allow a split (no record yet; T_A within one tile of the end) are counted in full by phase A, as before.
// cand_stat[2 pair] / [2 pair + 1]: candidates and evaluations of the hypotheses a FULL count of this call has seen for the pair so far (the run's
// first chunk: 128 iterations of mostly garbage) — the pair's own garbage candidate rate (5 % for calibrated poses at 50 % outliers, 10-15 % for the
// focal estimators' fundamental matrices), from which the split point follows.
__device__ __forceinline__ int count_split_tiles(const PairState &ps, int n_tiles, const unsigned long long *__restrict__ cand_stat, int pair) {
    if (!(ps.best_min_score < DBL_MAX) || n_tiles < 4 || !cand_stat) return n_tiles;
    const unsigned long long cs = cand_stat[2 * pair], es = cand_stat[2 * pair + 1];
    if (es == 0) return n_tiles;
    const double rec_score = ps.best_min_score * (1.0 + 1e-12);
    long long bar = (long long)floor((double)ps.n - rec_score / ps.sq_thr); // thr (n - c) >= rec_score  for  c <= bar  (heuristic here; the decisions use the exact expression)
    if (bar > (long long)ps.best_min_cnt) bar = (long long)ps.best_min_cnt;
    if (bar <= 0) return n_tiles;
    const double g = fmin(0.5, 1.1 * (double)cs / (double)es + 0.005);      // garbage candidate rate, with a margin: a hypothesis at that rate is decided after ra records
    const long long ra = (long long)((double)((long long)ps.n - bar) / (1.0 - g)) + 32;
    const int ta = (int)((ra + 255) >> 8);
    return ta + 2 > n_tiles ? n_tiles : ta;
}

// PHASE 0: the whole sweep in one launch (unit path).  1: phase A (tags / model_count = the chunk's tag lists; undecided -> tags_und / und_part / und_count).
// 2: phase B (tags / model_count = the undecided lists, part_in = their partial counts).
template <bool POSE, bool RAWF = false> // RAWF: the model IS a fundamental matrix (first nine doubles, row-major) — 7-point baseline
__global__ __launch_bounds__(CNT_THREADS, 4) void k_count(RunParams rp, const PairState *__restrict__ st, const uint4 *__restrict__ rfrag,
                                                          const Model *__restrict__ models, const uint32_t *__restrict__ tags,
                                                          const int32_t *__restrict__ model_count, const int32_t *__restrict__ plan,
                                                          uint32_t *__restrict__ tags_surv,
                                                          int32_t *__restrict__ surv_count, unsigned long long *__restrict__ stats,
                                                          int32_t *__restrict__ cand_out /*unit path: [models] candidate counts, or null*/,
                                                          const int32_t *__restrict__ tag_begin = nullptr /*or: entries [tag_begin[2 p], model_count[2 p]) of the tag lists*/,
                                                          int phase = 0, uint32_t *__restrict__ tags_und = nullptr, int32_t *__restrict__ und_part = nullptr,
                                                          int32_t *__restrict__ und_count = nullptr, const int32_t *__restrict__ part_in = nullptr,
                                                          unsigned long long *__restrict__ cand_stat = nullptr /*[2 batch]: see count_split_tiles*/) {
    // LDS: first the B fragments of the workgroup's 512 hypotheses (prologue), then the A-fragment tiles of the sweep
    constexpr int A_TILE_GROUPS = 16;                                   // 16 groups = 256 correspondences = 16 KiB per tile
    __shared__ uint4 s_lds[2 * A_TILE_GROUPS * 64];                     // 32 KiB: two tiles (double buffer)
    uint4 (*s_frag)[CNT_WAVE_MODELS][4] = reinterpret_cast<uint4 (*)[CNT_WAVE_MODELS][4]>(s_lds); // [wave][hypothesis][Eh | El | E8 parts | tb]
    static_assert(sizeof(uint4) * (CNT_THREADS / 64) * CNT_WAVE_MODELS * 4 <= sizeof(uint4) * 2 * A_TILE_GROUPS * 64, "fragment staging fits the tile buffers");
    const int total = plan[rp.batch];
    const int w = blockIdx.x;
    if (w >= total) return;
    const int pair = plan_find(plan, rp.batch, w);
    const int blk = w - plan[pair];
    const PairState &ps = st[pair];
    const int tbeg = tag_begin ? tag_begin[2 * pair] : 0;
    const int n = ps.n, cnt = model_count[2 * pair] - tbeg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blk * CNT_WG_MODELS + wave * CNT_WAVE_MODELS; // first hypothesis of this wavefront in the (sub-range of the) pair's tag list
    const size_t slot_base = (size_t)pair * rp.slot_stride;
    const uint32_t *tags_p = tags + slot_base + tbeg;
    const double thr = ps.sq_thr;
    // ---- prologue: 64 lanes build the B fragments of 64 hypotheses per round (wave-private LDS); the loads of both rounds
    // are issued before the arithmetic of the first (tag -> model is a dependent pair of L2 round trips)
    uint32_t slot_r[CNT_ROUNDS] = {};
    int part_r[CNT_ROUNDS] = {}; // phase B: candidates counted by phase A
    double mq[CNT_ROUNDS][4], mt[CNT_ROUNDS][3], mf[CNT_ROUNDS][2], ms[CNT_ROUNDS][2];
#pragma unroll
    for (int r = 0; r < CNT_ROUNDS; ++r) {
        const int i = m0 + 64 * r + lane;
        if (i < cnt) { slot_r[r] = tags_p[i] & 0xFFFFFFu; if (phase == 2) part_r[r] = part_in[slot_base + tbeg + i]; }
    }
#pragma unroll
    for (int r = 0; r < CNT_ROUNDS; ++r) {
        const Model *mp = models + slot_base + slot_r[r]; // slot 0 of the pair for idle lanes: a valid address
        const double2 *P = reinterpret_cast<const double2 *>(mp);
        const double2 q01 = P[0], q23 = P[1], t01 = P[2], t2s = P[3], f12 = P[5];
        mq[r][0] = q01.x; mq[r][1] = q01.y; mq[r][2] = q23.x; mq[r][3] = q23.y;
        mt[r][0] = t01.x; mt[r][1] = t01.y; mt[r][2] = t2s.x;
        mf[r][0] = f12.x; mf[r][1] = f12.y;
        ms[r][0] = t2s.y; ms[r][1] = RAWF ? P[4].x : 0.0;
    }
#pragma unroll
    for (int r = 0; r < CNT_ROUNDS; ++r) {
        const int i = m0 + 64 * r + lane;
        uint16_t eh[8], el[8], e8[3];
        float tb = 1.0f; // here: (tb S)^2 of count_setup_scaled; zero coefficients against 1 keep everything
#pragma unroll
        for (int j = 0; j < 8; ++j) { eh[j] = 0; el[j] = 0; }
        e8[0] = e8[1] = e8[2] = 0;
        if (i < cnt) {
            double R[9], Em[9], E[9];
            if (RAWF) {
#pragma unroll
                for (int q = 0; q < 4; ++q) E[q] = mq[r][q];
                E[4] = mt[r][0]; E[5] = mt[r][1]; E[6] = mt[r][2]; E[7] = ms[r][0]; E[8] = ms[r][1];
            } else {
                quat_to_R(mq[r], R);
                essential_from_Rt(R, mt[r], Em);
                if (POSE) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) E[q] = Em[q];
                } else fundamental_from_E(Em, mf[r][0], mf[r][1], E);
            }
            count_setup_scaled(E, ps.box, thr, eh, el, e8, tb);
        }
        uint4 *dst = s_frag[wave][64 * r + lane];
        dst[0] = pack_bf16x8(eh);
        dst[1] = pack_bf16x8(el);
        dst[2] = make_uint4((uint32_t)e8[0] | ((uint32_t)e8[1] << 16), (uint32_t)e8[2], 0u, 0u);
        dst[3] = make_uint4(__float_as_uint(tb), 0u, 0u, 0u);
    }
    __syncthreads();
    const bool idle = m0 >= cnt; // nothing to count for this wavefront: it still helps to stage the tiles
    const int col = lane & 15, ksl = lane >> 4;
    const int part = ksl < 2 ? 0 : ksl - 1; // lanes 0..31 carry Eh, 32..47 El, 48..63 the constant term
    bf16x8_t bfrag[CNT_TILES];
    float tbv[CNT_TILES];  // (tb S)^2
    float cand[CNT_TILES]; // running candidate counts (exact in fp32: <= 2^24)
#pragma unroll
    for (int t = 0; t < CNT_TILES; ++t) {
        const uint4 *src = s_frag[wave][16 * t + col];
        bfrag[t] = __builtin_bit_cast(bf16x8_t, src[part]);
        const float tb2 = __uint_as_float(src[3].x);
        tbv[t] = tb2;
        cand[t] = 0.f;
    }
    __syncthreads(); // the fragments are in registers: the buffer now holds correspondence tiles
    // ---- sweep.  The pair's A fragments (64 N bytes, k_prep) are staged through LDS one 16-group tile at a time, double
    // buffered: read from global memory ONCE per workgroup instead of once per wavefront.
    // The group body is a hand-ordered software pipeline (inline asm; hipcc's own schedule reuses one accumulator quad
    // and pads every dependence with s_nop: 3x slower).  Per pair of MFMA tiles:
    //     16 VALU on the accumulators of group g   |   2 MFMAs of group g + 1 into the same accumulators
    // so every accumulator is read >= 40 instructions after its MFMA issued (no wait states needed; hipcc pads nothing
    // around inline asm).  The test, per evaluation:
    //     v_fma_f32 d = -C * C + tb^2, clamp      1 = candidate, 0 = definite outlier (scaled so that nothing falls between)
    //     v_add_f32                               tree of the eight values of a tile pair into the two counters
    // Plain fp32, not packed: tools/ubench/valu_rates.hip measures, per MFMA at 4 wavefronts per SIMD, 13.5 cycles for the
    // v_mfma_f32_16x16x32_bf16 alone, 20.5 with four v_fma_f32 beside it (the two pipes do not overlap within a SIMD: the
    // costs add), but 35.8 with four v_pk_fma_f32 / v_pk_add_f32 — packed fp32 next to MFMAs is an anti-lever on gfx950.
    // C is never NaN: rows of non-finite correspondences and the coefficients of unjudgeable models are zeroed when the
    // fragments are built.
    f32x4_t acc[CNT_TILES];
    auto mfma2 = [&](const uint4 &araw, int t0) {
        const bf16x8_t a = __builtin_bit_cast(bf16x8_t, araw);
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, 0\n\t"
                     "v_mfma_f32_16x16x32_bf16 %1, %2, %4, 0"
                     : "=&v"(acc[t0]), "=&v"(acc[t0 + 1]) : "v"(a), "v"(bfrag[t0]), "v"(bfrag[t0 + 1]));
    };
    auto test2 = [&](int t0) {
        float d0, d1, d2, d3, d4, d5, d6, d7;
        asm volatile("v_fma_f32 %2, -%10, %10, %18 clamp\n\t"
                     "v_fma_f32 %3, -%11, %11, %18 clamp\n\t"
                     "v_fma_f32 %4, -%12, %12, %18 clamp\n\t"
                     "v_fma_f32 %5, -%13, %13, %18 clamp\n\t"
                     "v_fma_f32 %6, -%14, %14, %19 clamp\n\t"
                     "v_fma_f32 %7, -%15, %15, %19 clamp\n\t"
                     "v_fma_f32 %8, -%16, %16, %19 clamp\n\t"
                     "v_fma_f32 %9, -%17, %17, %19 clamp\n\t"
                     "v_add_f32 %2, %2, %3\n\t"
                     "v_add_f32 %4, %4, %5\n\t"
                     "v_add_f32 %6, %6, %7\n\t"
                     "v_add_f32 %8, %8, %9\n\t"
                     "v_add_f32 %0, %0, %2\n\t"
                     "v_add_f32 %1, %1, %6\n\t"
                     "v_add_f32 %0, %0, %4\n\t"
                     "v_add_f32 %1, %1, %8"
                     : "+v"(cand[t0]), "+v"(cand[t0 + 1]), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                     : "v"(acc[t0][0]), "v"(acc[t0][1]), "v"(acc[t0][2]), "v"(acc[t0][3]), "v"(acc[t0 + 1][0]), "v"(acc[t0 + 1][1]),
                       "v"(acc[t0 + 1][2]), "v"(acc[t0 + 1][3]), "v"(tbv[t0]), "v"(tbv[t0 + 1]));
    };
    const int G = (n + 15) >> 4;
    const uint4 *A = rfrag + (size_t)pair * ((rp.n_max + 15) >> 4) * 64;
    const int n_tiles = (G + A_TILE_GROUPS - 1) / A_TILE_GROUPS;
    const int t_split = phase == 0 ? n_tiles : count_split_tiles(ps, n_tiles, cand_stat, pair);
    const int t_lo = phase == 2 ? t_split : 0, t_hi = phase == 1 ? t_split : n_tiles; // this launch sweeps the tiles [t_lo, t_hi)
    uint4 stage[4];
    auto fetch = [&](int tile) { // 256 threads x 4 x 16 B = one tile; past the last group: zeros (C = 0: never an outlier)
        const int g0 = tile * A_TILE_GROUPS;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = k * CNT_THREADS + tid; // uint4 index inside the tile
            stage[k] = (g0 + (idx >> 6)) < G ? A[(size_t)g0 * 64 + idx] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s_lds[buf * A_TILE_GROUPS * 64 + k * CNT_THREADS + tid] = stage[k];
    };
    if (t_lo >= t_hi) return; // (phase B of a pair that was not split: nothing on its undecided list either; uniform per workgroup)
    fetch(t_lo);
    commit(t_lo & 1);
    __syncthreads();
    // every tile is swept as 16 groups (zero rows past the end cost nothing but their share of the last tile)
    for (int tile = t_lo; tile < t_hi; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < t_hi) fetch(tile + 1);
        if (!idle) {
            const uint4 *T = s_lds + buf * A_TILE_GROUPS * 64 + lane;
            uint4 f0 = T[0], f1 = T[64];
#pragma unroll
            for (int t = 0; t < CNT_TILES; t += 2) mfma2(f0, t);
#pragma unroll
            for (int g = 0; g < A_TILE_GROUPS; g += 2) {
                // group g is in the accumulators; f1 = fragment of g + 1
                f0 = T[(size_t)min(g + 2, A_TILE_GROUPS - 1) * 64];
#pragma unroll
                for (int t = 0; t < CNT_TILES; t += 2) { test2(t); mfma2(f1, t); }
                f1 = T[(size_t)min(g + 3, A_TILE_GROUPS - 1) * 64];
                if (g + 2 < A_TILE_GROUPS) {
#pragma unroll
                    for (int t = 0; t < CNT_TILES; t += 2) { test2(t); mfma2(f0, t); }
                } else {
#pragma unroll
                    for (int t = 0; t < CNT_TILES; t += 2) test2(t);
                }
            }
        }
        if (tile + 1 < t_hi) commit(buf ^ 1); // the other buffer was last read before the previous barrier
        __syncthreads();
    }
    if (idle) return;
    // ---- epilogue: totals per hypothesis, retirement, survivor list
    uint32_t *s_out = reinterpret_cast<uint32_t *>(&s_frag[wave][0][0]); // wave-private, fragments are in registers now
    const int swept = (t_hi - t_lo) * A_TILE_GROUPS * 16, real = min(max(n - t_lo * A_TILE_GROUPS * 16, 0), swept);
    const int pad = swept - real; // zero rows past the end always count as candidates
    const bool split_a = phase == 1 && t_hi < n_tiles;             // phase A of a split pair: totals are not known yet
    const int rest = split_a ? n - t_hi * A_TILE_GROUPS * 16 : 0;  // records phase B still has to sweep
#pragma unroll
    for (int t = 0; t < CNT_TILES; ++t) {
        float o = cand[t];
        o += __shfl_xor(o, 16, 64);
        o += __shfl_xor(o, 32, 64);
        if (ksl == 0) s_out[16 * t + col] = (uint32_t)((int)o - pad);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const long long rec_cnt = (long long)ps.best_min_cnt;
    const double rec_score = ps.best_min_score < DBL_MAX ? ps.best_min_score * (1.0 + 1e-12) : DBL_MAX;
#pragma unroll
    for (int r = 0; r < CNT_ROUNDS; ++r) {
        const int i = m0 + 64 * r + lane;
        const bool live = i < cnt;
        const int cnd_here = live ? (int)s_out[64 * r + lane] + part_r[r] : 0;
        const int cnd = cnd_here + rest; // phase A of a split pair: the most the total can still become
        if (cand_stat && phase == 1 && !split_a && rec_cnt == 0 && !(ps.best_min_score < DBL_MAX)) { // the run's first chunk (no record yet): the pair's candidate statistics
            const int tot = wave_sum_i(cnd_here), lv = __popcll(__ballot(live));
            if (lane == 0 && lv) { atomicAdd(&cand_stat[2 * pair], (unsigned long long)tot); atomicAdd(&cand_stat[2 * pair + 1], (unsigned long long)lv * (unsigned long long)n); }
        }
        if (cand_out && live) cand_out[i] = cnd;
        const bool surv = live && ((long long)cnd > rec_cnt || thr * (double)(n - cnd) < rec_score); // else: its slot keeps k_solve's -2
        const unsigned long long ball = __ballot(surv);
        if (ball) {
            int base = 0;
            const int first = __ffsll((long long)ball) - 1;
            if (lane == first) base = atomicAdd(split_a ? &und_count[2 * pair] : &surv_count[pair], __popcll(ball));
            base = __shfl(base, first, 64);
            if (surv) {
                const size_t at = slot_base + base + __popcll(ball & ((1ull << lane) - 1ull));
                if (split_a) { tags_und[at] = slot_r[r]; und_part[at] = cnd_here; } // undecided: phase B finishes its count
                else {
                    const uint32_t key = (uint32_t)min(PROBE_PTS, (int)(((long long)cnd * PROBE_PTS + n - 1) / n));
                    tags_surv[at] = slot_r[r] | (key << 24);
                }
            }
        }
    }
    if (stats && lane == 0 && wave == 0 && blk == 0) { // per pair, once: evaluations the CPU loop would do, and those the MFMA sweep does
        if (phase != 2) atomicAdd(&stats[0], (unsigned long long)cnt * (unsigned long long)n);
        atomicAdd(&stats[1], (unsigned long long)(((cnt + 15) / 16) * 16) * (unsigned long long)swept);
    }
}

// ------------------------------------------------------------------------------------------------ bound (fp32)
// Second retirement stage, for the hypotheses k_count let through (at 50 % outliers: the ~6 % with more candidates than
// roughly half of the record's inliers; on outlier-free data: all of them).  One lane per hypothesis sweeps the pair's
// correspondences in fp32 and sums a LOWER bound of every MSAC term (mdrp_math.h bound_r2_32); with its own count of
// k_count as the upper bound of the inlier count, the two record tests of score_models<> are decided for everything except
// true record candidates and near-ties, and only those reach the fp64 sweep.  Branch-free: ~23 fp32 instructions per
// evaluation against ~12 (mostly fp64, divergent) of the exact sweep's survivors, and no second phase.
constexpr int BND_THREADS = 256;
#define MDRP_BND_TILE 256
constexpr int BND_TILE = MDRP_BND_TILE; // correspondences per LDS tile (16 B each, fp32); early-exit test and compaction per tile

// E (or F) of a model as the scoring sweeps see it
template <bool POSE, bool RAWF>
__device__ __forceinline__ void model_matrix(const Model &m, double E[9]) {
    if (RAWF) {
#pragma unroll
        for (int q = 0; q < 9; ++q) E[q] = reinterpret_cast<const double *>(&m)[q];
    } else {
        double R[9], Em[9];
        quat_to_R(m.q, R);
        essential_from_Rt(R, m.t, Em);
        if (POSE) {
#pragma unroll
            for (int q = 0; q < 9; ++q) E[q] = Em[q];
        } else fundamental_from_E(Em, m.f1, m.f2, E);
    }
}

// Both proofs are monotone in the records seen so far (the terms of the score bound are >= 0; unseen records can add at most
// their number to the inlier bound), so after every tile a model whose partial sums already prove it irrelevant leaves; the
// workgroup then packs the models that are still open into as few wavefronts as they need (their state moves through LDS), so
// a wavefront never idles along beside one open lane, and wavefronts without open models only help load tiles.  Outlier-free
// pairs, where every model is good and the bound has to see a large part of the sum: 8.4 -> 5.9 ms per 1024 pairs at 512
// records per tile, 5.6 at 256.  (Measured and dropped: walking the records in the order of their residual under the best
// minimal model so far, largest first — residuals of minimal models are dominated by each model's own error, not by the
// record's noise, so one model's order says little about another's: -6 % on k_bound for a 0.18 ms sort.)
template <bool POSE, bool RAWF = false>
__global__ __launch_bounds__(BND_THREADS, 4) void k_bound(RunParams rp, const PairState *__restrict__ st, const double *__restrict__ pts,
                                                          const Model *__restrict__ models, const uint32_t *__restrict__ tags_in,
                                                          const int32_t *__restrict__ cnt_in, const int32_t *__restrict__ plan,
                                                          uint32_t *__restrict__ tags_out,
                                                          int32_t *__restrict__ cnt_out, unsigned long long *__restrict__ stats,
                                                          double *__restrict__ dbg_lb = nullptr /*unit entry point: the two bounds per model*/,
                                                          int32_t *__restrict__ dbg_cnt_ub = nullptr) {
    __shared__ float4 s_rec[BND_TILE];
    struct Open { uint32_t tag; int32_t sane; float Ef[9], eC, eD, thr_dn, c0, c1; double lb; }; // a model's state while it is still open
    __shared__ Open s_open[BND_THREADS];
    __shared__ int s_wave_open[BND_THREADS / 64];
    const int total = plan[rp.batch];
    const int w = blockIdx.x;
    if (w >= total) return;
    const int pair = plan_find(plan, rp.batch, w);
    const int blk = w - plan[pair];
    const PairState &ps = st[pair];
    const int n = ps.n, cnt = cnt_in[pair];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blk * BND_THREADS + tid;
    bool open = i < cnt; // this lane holds a model that is neither retired nor passed on yet
    const size_t slot_base = (size_t)pair * rp.slot_stride;
    const double thr = ps.sq_thr;
    uint32_t tag = 0;
    float Ef[9], eC = 0.f, eD = 1.f, thr_dn = 0.f;
    bool sane = false;
#pragma unroll
    for (int q = 0; q < 9; ++q) Ef[q] = 0.f;
    if (open) {
        tag = tags_in[slot_base + i];
        const Model m = models[slot_base + (tag & 0xFFFFFFu)];
        double E[9];
        model_matrix<POSE, RAWF>(m, E);
        sane = bound_setup32(E, ps.box, thr, Ef, eC, eD, thr_dn);
    }
    // Two correspondences per step in packed fp32 (every VALU instruction costs 4 cycles per wavefront; v_pk_* carry two
    // values): 16 v_pk_fma for C and den of both, then per element |C| - eC, max, rcp, min.  The inlier-count upper bound is a
    // clamped packed FMA: clamp((thr_cnt - q) * 2^80) is 1 for q < thr_cnt, 0 for q >= thr_cnt (and for NaN), summed in fp32.
    double total_lb = 0.0;
    f32x2 cnt2 = {0.f, 0.f}; // correspondences whose lower bound is below the threshold: an upper bound of the inlier count
    const float thr_cnt = (float)(thr * (1.0 + 1e-5)) * (1.0f + 1e-6f); // q may exceed r^2 by its own roundings (<= 8u)
    const float BIG = 1.2089258e24f; // 2^80
    const f32x2 nbig = {-BIG, -BIG}, kcnt = {thr_cnt * BIG, thr_cnt * BIG};
    const long long rec_cnt = (long long)ps.best_min_cnt;
    const double rec_score = ps.best_min_score < DBL_MAX ? ps.best_min_score * (1.0 + 1e-12) : DBL_MAX;
    float4 *s_pair = s_rec; // two correspondences per 32 B, component-major: (a0 a1 b0 b1)(c0 c1 d0 d1)
    const double *gp = pts + (size_t)pair * rp.n_max * PT_STRIDE;
    for (int t0 = 0; t0 < n; t0 += BND_TILE) {
        const int npts = min(BND_TILE, n - t0);
        __syncthreads();
        for (int j = tid; j < npts; j += BND_THREADS) {
            const double2 *src = reinterpret_cast<const double2 *>(gp + (size_t)(t0 + j) * PT_STRIDE);
            const double2 p0 = src[0], p1 = src[1];
            store_rec32(s_pair, j, p0.x, p0.y, p1.x, p1.y);
        }
        __syncthreads();
        if (__ballot(open)) { // wavefronts without an open model only helped load the tile
            f32x2 Ev[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) Ev[q] = (f32x2){Ef[q], Ef[q]};
            const f32x2 eDv = {eD, eD};
            const int npairs = npts >> 1;
            for (int j0 = 0; j0 < npairs; j0 += 32) { // partial sums of <= 64 terms in fp32, then fp64
                const int je = min(32, npairs - j0);
                f32x2 part = {0.f, 0.f};
#pragma unroll 2
                for (int j = 0; j < je; ++j) {
                    const float4 ab = s_pair[2 * (j0 + j)], cd = s_pair[2 * (j0 + j) + 1];
                    const f32x2 a = {ab.x, ab.y}, b = {ab.z, ab.w}, c = {cd.x, cd.y}, d = {cd.z, cd.w};
                    const f32x2 e0 = __builtin_elementwise_fma(Ev[0], a, __builtin_elementwise_fma(Ev[1], b, Ev[2]));
                    const f32x2 e1 = __builtin_elementwise_fma(Ev[3], a, __builtin_elementwise_fma(Ev[4], b, Ev[5]));
                    const f32x2 e2 = __builtin_elementwise_fma(Ev[6], a, __builtin_elementwise_fma(Ev[7], b, Ev[8]));
                    const f32x2 g0 = __builtin_elementwise_fma(Ev[0], c, __builtin_elementwise_fma(Ev[3], d, Ev[6]));
                    const f32x2 g1 = __builtin_elementwise_fma(Ev[1], c, __builtin_elementwise_fma(Ev[4], d, Ev[7]));
                    const f32x2 C = __builtin_elementwise_fma(c, e0, __builtin_elementwise_fma(d, e1, e2));
                    const f32x2 den = __builtin_elementwise_fma(e0, e0, __builtin_elementwise_fma(e1, e1, __builtin_elementwise_fma(g0, g0, __builtin_elementwise_fma(g1, g1, eDv))));
                    const f32x2 t = {fmaxf(fabsf(C.x) - eC, 0.0f), fmaxf(fabsf(C.y) - eC, 0.0f)};
                    const f32x2 rc = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)}; // v_rcp_f32: 1 ulp, inside BOUND_SLACK
                    const f32x2 q = (t * t) * rc;
                    const f32x2 term = {__builtin_fminf(q.x, thr_dn), __builtin_fminf(q.y, thr_dn)}; // v_min_f32: NaN -> thr, like the reference's `r2 < thr`
                    part += term;
                    f32x2 o;
                    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(o) : "v"(q), "v"(nbig), "v"(kcnt));
                    cnt2 += o;
                }
                total_lb += (double)part.x + (double)part.y;
            }
            if (npts & 1) { // odd tail of the last tile
                const float *fl = reinterpret_cast<const float *>(s_pair) + (npts >> 1) * 8;
                const float q = bound_r2_32(Ef, eC, eD, fl[0], fl[2], fl[4], fl[6]);
                total_lb += (double)__builtin_fminf(q, thr_dn);
                cnt2.x += (q < thr_cnt) ? 1.0f : 0.0f;
            }
        }
        // Monotone exit: unseen records can only add inliers (at most their number) and can only raise the score bound; after the
        // last tile this IS the full test.  A model outside the fp32 range (`sane` false) is never retired.
        const int seen = t0 + npts;
        if (open && sane) {
            const long long cnt_ub = (long long)(cnt2.x + cnt2.y) + 1 + (long long)(n - seen); // + 1: the clamped sum may carry a fraction
            if (cnt_ub <= rec_cnt && total_lb * (1.0 - BOUND_SLACK) >= rec_score) open = false; // its slot keeps k_solve's -2
        }
        if (seen >= n) break;
        // pack the open models into as few wavefronts as they need
        const unsigned long long ob = __ballot(open);
        if (lane == 0) s_wave_open[wave] = __popcll(ob);
        __syncthreads();
        int tot = 0, before = 0, waves_used = 0;
#pragma unroll
        for (int v = 0; v < BND_THREADS / 64; ++v) { const int c = s_wave_open[v]; if (v < wave) before += c; tot += c; waves_used += c > 0; }
        if (tot == 0) break; // (uniform) every model of this workgroup is decided
        if ((tot + 63) / 64 < waves_used) {
            if (open) {
                Open &o = s_open[before + __popcll(ob & ((1ull << lane) - 1ull))];
                o.tag = tag; o.sane = sane; o.eC = eC; o.eD = eD; o.thr_dn = thr_dn; o.c0 = cnt2.x; o.c1 = cnt2.y; o.lb = total_lb;
#pragma unroll
                for (int q = 0; q < 9; ++q) o.Ef[q] = Ef[q];
            }
            __syncthreads();
            open = tid < tot;
            if (open) {
                const Open &o = s_open[tid];
                tag = o.tag; sane = o.sane != 0; eC = o.eC; eD = o.eD; thr_dn = o.thr_dn; cnt2.x = o.c0; cnt2.y = o.c1; total_lb = o.lb;
#pragma unroll
                for (int q = 0; q < 9; ++q) Ef[q] = o.Ef[q];
            }
        }
    }
    // what is still marked open here has seen every record without being retired: on to the exact sweep
    const bool surv = open;
    if (dbg_lb && surv) { // mdrp_bound_models: no records, so every model arrives here with its full sums (a model outside the fp32 range proves nothing)
        dbg_lb[tag & 0xFFFFFFu] = sane ? total_lb * (1.0 - BOUND_SLACK) : 0.0;
        dbg_cnt_ub[tag & 0xFFFFFFu] = sane ? (int32_t)min((long long)n, (long long)(cnt2.x + cnt2.y) + 1) : n;
    }
    const unsigned long long ball = __ballot(surv);
    if (ball) {
        int base = 0;
        const int first = __ffsll((long long)ball) - 1;
        if (lane == first) base = atomicAdd(&cnt_out[pair], __popcll(ball));
        base = __shfl(base, first, 64);
        if (surv) tags_out[slot_base + base + __popcll(ball & ((1ull << lane) - 1ull))] = tag;
    }
    if (stats && tid == 0 && blk == 0) atomicAdd(stats, (unsigned long long)cnt * (unsigned long long)n);
}

// The survivors of k_count, classified and ordered by their candidate density.  One workgroup per pair: key >= DENSE_KEY of
// 64 -> "dense" list (single-pass sweep), written from the BACK of tags_sorted; everything else -> counting sort by key into
// the front (two-phase sweep).  Phase 2 of the sweep costs a wavefront the MAXIMUM candidate count over its lanes, and
// densities differ by 10x between hypotheses, so the sweep wants workgroups of similar hypotheses.  Whole workgroups, not
// wavefronts: the four wavefronts of a workgroup meet at the tile barriers, and one slow wavefront parks the other three in
// their SIMD slots (sorting inside the workgroup made the sweep 3x slower).  Which lane scores a hypothesis does not
// change its result.
// In: surv_count[p] survivors with keys in tags (k_count).  Out: model_count[2p] = sparse, [2p+1] = dense (model_count[2p]
// held the pair's model count until here: k_count has consumed it).
MDRP_GLOBAL __launch_bounds__(256) void k_sort_tags(RunParams rp, const PairState *__restrict__ st, int32_t *__restrict__ model_count,
                                                   const int32_t *__restrict__ surv_count,
                                                   const uint32_t *__restrict__ tags, uint32_t *__restrict__ tags_sorted) {
    const int pair = blockIdx.x, tid = threadIdx.x;
    const PairState &ps = st[pair];
    if (!ps.active) return;
    __shared__ int s_hist[PROBE_PTS + 1], s_pos[PROBE_PTS + 1];
    __shared__ int s_dense;
    if (tid <= PROBE_PTS) s_hist[tid] = 0;
    if (tid == 0) s_dense = 0;
    __syncthreads();
    const int cnt = surv_count[pair];
    const size_t slot_base = (size_t)pair * rp.slot_stride;
    const uint32_t *src = tags + slot_base;
    uint32_t *dst = tags_sorted + slot_base;
    const int dense_min = ps.n >= 8 ? MDRP_DENSE_KEY : PROBE_PTS + 1;
    for (int i = tid; i < cnt; i += 256) atomicAdd(&s_hist[min(src[i] >> 24, (uint32_t)PROBE_PTS)], 1);
    __syncthreads();
    if (tid < 64) { // exclusive prefix over the sparse keys
        const int v = tid < dense_min ? s_hist[tid] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (tid >= o) incl += u;
        }
        s_pos[tid] = incl - v;
        if (tid == 63) s_pos[64] = incl;
    }
    __syncthreads();
    const int cap = rp.slot_stride;
    for (int i = tid; i < cnt; i += 256) {
        const uint32_t t = src[i];
        const int key = (int)min(t >> 24, (uint32_t)PROBE_PTS);
        if (key >= dense_min) dst[cap - 1 - atomicAdd(&s_dense, 1)] = t & 0xFFFFFFu;
        else dst[atomicAdd(&s_pos[key], 1)] = t & 0xFFFFFFu;
    }
    __syncthreads();
    if (tid == 0) { model_count[2 * pair] = cnt - s_dense; model_count[2 * pair + 1] = s_dense; }
}

// Work plan of one sweep launch: the workgroups a pair needs (ceil(count / SCORE_THREADS) per density class).
// One wavefront, 64 pairs per step.  plan[0..B] = prefix sum of blocks per pair, plan[B+1 .. 2B] = sparse blocks of the pair.
MDRP_GLOBAL __launch_bounds__(PLAN_THREADS) void k_plan(int batch, const int32_t *__restrict__ model_count, int32_t *__restrict__ plan,
                                                       int32_t *__restrict__ totals /*[0] dense, [1] dense + sparse, [2] queue head*/) {
    __shared__ int s_w[PLAN_THREADS / 64 + 1];
    int run_d = 0, run_s = 0;
    int32_t *pd = plan, *psp = plan + batch + 1;
    for (int p0 = 0; p0 < batch; p0 += PLAN_THREADS) {
        const int p = p0 + threadIdx.x;
        int bd = 0, bs = 0;
        if (p < batch) {
            bs = (model_count[2 * p] + SCORE_THREADS - 1) / SCORE_THREADS;
            bd = (model_count[2 * p + 1] + SCORE_THREADS - 1) / SCORE_THREADS;
        }
        int tot_d, tot_s;
        const int ed = plan_block_scan(bd, tot_d, s_w), es = plan_block_scan(bs, tot_s, s_w);
        // pair-major order: a pair's sparse blocks then its dense blocks, pairs consecutive (dense workgroups, whose
        // phase 2 reads LDS at per-lane addresses, stay spread out in time instead of saturating every CU's LDS at once)
        if (p < batch) { pd[p] = run_d + run_s + ed + es; psp[p] = bs; }
        run_d += tot_d;
        run_s += tot_s;
    }
    if (threadIdx.x == 0) { pd[batch] = run_d + run_s; psp[batch] = 0; totals[0] = run_d; totals[1] = run_d + run_s; totals[2] = 0; }
}

// Workgroup w handles item w of the plan (grid = an upper bound, surplus workgroups at the END exit at once).  A static
// blockIdx -> (pair, block) map with per-pair padding put the live blocks of every pair on the same few XCDs whenever
// blocks-per-pair shared a factor with 8 (workgroups are dealt round-robin over the 8 XCDs by linear id): a
// 2048-iteration chunk took as long as a 7440-iteration one.  With the compacted plan consecutive workgroups are
// consecutive live items, so every XCD gets the same mix.  (A persistent variant pulling items from an atomic queue
// was 3x slower; a persistent static-stride loop 25 % slower from imbalance.)
template <bool POSE, bool RAWF = false>
__global__ __launch_bounds__(SCORE_THREADS, MDRP_SCORE_MINWAVES) void k_score(RunParams rp, const PairState *__restrict__ st, const double *__restrict__ pts,
                                                         const Model *__restrict__ models, const uint32_t *__restrict__ tags,
                                                         const int32_t *__restrict__ model_count, double *__restrict__ slot_score,
                                                         int32_t *__restrict__ slot_inl, const int32_t *__restrict__ plan,
                                                         int32_t *__restrict__ totals) {
    extern __shared__ double tile[]; // TILE_PTS * PT_STRIDE doubles, then TILE_PTS float4 (fp32 coordinates)
    float4 *tile32 = reinterpret_cast<float4 *>(tile + TILE_PTS * PT_STRIDE);
    const int total = totals[1];
    const int tid = threadIdx.x;
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
    const int pair = plan_find(plan, rp.batch, w);
    const int blk_sparse = plan[rp.batch + 1 + pair];
    const int bi = w - plan[pair];
    const bool dense = bi >= blk_sparse;
    const int blk = dense ? bi - blk_sparse : bi;
    const int cnt_sparse = model_count[2 * pair], cnt_dense = model_count[2 * pair + 1];
    const PairState &ps = st[pair];
    const int n = ps.n;
    const double thr = ps.sq_thr;
    const size_t slot_base = (size_t)pair * rp.slot_stride;
    const int cap = rp.slot_stride;
    bool live;
    uint32_t slot = 0;
    { // sorted tag list of the pair: sparse hypotheses from the front, dense ones from the back (k_sort_tags)
        const int i = blk * SCORE_THREADS + tid;
        live = i < (dense ? cnt_dense : cnt_sparse);
        if (live) slot = tags[slot_base + (dense ? cap - 1 - i : i)];
    }
    double E[9], thr_dmax;
    float tb, Ef[9];
    const Model *mp = models + slot_base + slot;
#pragma unroll
    for (int i = 0; i < 9; ++i) E[i] = 0;
    if (live) {
        const Model m = *mp;
        double R[9], Em[9];
        if (RAWF) {
#pragma unroll
            for (int i = 0; i < 9; ++i) E[i] = reinterpret_cast<const double *>(&m)[i];
        } else {
            quat_to_R(m.q, R);
            essential_from_Rt(R, m.t, Em);
            if (POSE) {
#pragma unroll
                for (int i = 0; i < 9; ++i) E[i] = Em[i];
            } else {
                fundamental_from_E(Em, m.f1, m.f2, E);
            }
        }
    }
    bound_setup(E, ps, thr, Ef, tb, thr_dmax);
    double score = 0;
    int cnt = 0;
    Prune pr;
    pr.rec_cnt = (long long)ps.best_min_cnt;
    pr.rec_score = ps.best_min_score < DBL_MAX ? ps.best_min_score * (1.0 + 1e-12) : DBL_MAX;
    pr.n = n; pr.processed = 0; pr.dead = !live; pr.wave_dead = false;
    const double *gp = pts + (size_t)pair * rp.n_max * PT_STRIDE;
    for (int t0 = 0; t0 < n; t0 += TILE_PTS) {
        const int npts = min(TILE_PTS, n - t0);
        __syncthreads();
        { // cooperative load of the tile: one 48-B record per thread and trip (a wavefront covers 3 KiB contiguous)
            const double2 *src = reinterpret_cast<const double2 *>(gp + (size_t)t0 * PT_STRIDE);
            double2 *dst = reinterpret_cast<double2 *>(tile);
            for (int i = tid; i < npts; i += SCORE_THREADS) {
                const double2 p0 = src[3 * i], p1 = src[3 * i + 1], p2 = src[3 * i + 2];
                dst[3 * i] = p0; dst[3 * i + 1] = p1; dst[3 * i + 2] = p2;
                if (!dense) store_rec32(tile32, i, p0.x, p0.y, p1.x, p1.y); // fp32 coordinates for the phase-1 filter
            }
        }
        __syncthreads();
        if (!pr.wave_dead) {
            if (dense) score_tile_dense<POSE>(tile, npts, E, mp, thr, score, cnt, pr);
            else score_tile_f32<POSE>(tile, tile32, npts, E, Ef, tb, mp, thr, score, cnt, pr);
        }
    }
    if (live) {
        const bool pruned = pr.dead;
        slot_score[slot_base + slot] = pruned ? DBL_MAX : score + thr * (double)(n - cnt);
        slot_inl[slot_base + slot] = pruned ? -2 : cnt;
    }
    } // item loop
}

// ------------------------------------------------------------------------------------------------ score, one wavefront per hypothesis
// Small batches (round 6).  k_score gives every hypothesis a LANE that walks the pair's records in a serial loop: ~0.3 ms for N = 2000 whatever
// the number of hypotheses — with one image pair per call (what /root/reference/eval.py does) the two exact sweeps of a run were 0.6 of its 1.7 ms,
// on a chip that was 99 % idle.  Here a WAVEFRONT owns a hypothesis and its lanes take 64 records per trip (coalesced 48-byte loads, the same
// score_point arithmetic per record).  The MSAC score is a sum in RECORD ORDER (compute_sampson_msac_score @0x4f61d0 adds as it goes, and scores
// are compared with `<`): the inliers' r^2 of a trip are compacted in lane order into LDS and added one after the other, so the sum — and the
// early exit against the records of earlier chunks (struct Prune) — is k_score's bit for bit.  Twice the instructions per hypothesis, none of the
// latency: used where the hypotheses would not fill the chip anyway (mdrp_capi.hip: calls of at most SCORE_WAVE_MAX_PAIRS pairs).
constexpr int SCORE_WAVE_MAX_PAIRS = 128;
constexpr int SCW_THREADS = 256; // four hypotheses per workgroup
template <bool POSE, bool RAWF = false>
__global__ __launch_bounds__(SCW_THREADS) void k_score_w(RunParams rp, const PairState *__restrict__ st, const double *__restrict__ pts,
                                                         const Model *__restrict__ models, const uint32_t *__restrict__ tags /*sorted: k_sort_tags*/,
                                                         const int32_t *__restrict__ model_count, double *__restrict__ slot_score,
                                                         int32_t *__restrict__ slot_inl, const int32_t *__restrict__ plan /*k_count_plan over the survivors, 4 per workgroup*/) {
    __shared__ double s_r2[SCW_THREADS / 64][64];
    const int total = plan[rp.batch];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        const int pair = plan_find(plan, rp.batch, w);
        const int i = (w - plan[pair]) * (SCW_THREADS / 64) + wave; // this wavefront's hypothesis in the pair's sorted list
        const int cnt_sparse = model_count[2 * pair], cnt_dense = model_count[2 * pair + 1];
        if (i >= cnt_sparse + cnt_dense) continue; // (wave-uniform)
        const PairState &ps = st[pair];
        const int n = ps.n;
        const double thr = ps.sq_thr;
        const size_t slot_base = (size_t)pair * rp.slot_stride;
        const uint32_t slot = tags[slot_base + (i < cnt_sparse ? i : rp.slot_stride - 1 - (i - cnt_sparse))];
        const Model m = models[slot_base + slot];
        double E[9], R[9], t[3] = {m.t[0], m.t[1], m.t[2]};
#pragma unroll
        for (int q = 0; q < 9; ++q) R[q] = 0;
        if (RAWF) {
#pragma unroll
            for (int q = 0; q < 9; ++q) E[q] = reinterpret_cast<const double *>(&m)[q];
        } else {
            double Em[9];
            quat_to_R(m.q, R);
            essential_from_Rt(R, m.t, Em);
            if (POSE) {
#pragma unroll
                for (int q = 0; q < 9; ++q) E[q] = Em[q];
            } else fundamental_from_E(Em, m.f1, m.f2, E);
        }
        const long long rec_cnt = (long long)ps.best_min_cnt;
        const double rec_score = ps.best_min_score < DBL_MAX ? ps.best_min_score * (1.0 + 1e-12) : DBL_MAX;
        const double *gp = pts + (size_t)pair * rp.n_max * PT_STRIDE;
        double score = 0;
        int cnt = 0;
        bool pruned = false;
        for (int t0 = 0; t0 < n; t0 += 64) {
            const int r = t0 + lane;
            double s1 = 0;
            int c1 = 0;
            if (r < n) score_point<POSE>(gp + (size_t)r * PT_STRIDE, E, R, t, thr, s1, c1); // s1 = 0 + r^2 of an inlier
            const unsigned long long ball = __ballot(c1 != 0);
            if (ball) {
                if (c1) s_r2[wave][__popcll(ball & lt)] = s1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int k = __popcll(ball);
                for (int j = 0; j < k; ++j) score += s_r2[wave][j]; // in record order, one after the other (every lane the same sum)
                cnt += k;
                __builtin_amdgcn_wave_barrier(); // (the next trip overwrites the buffer)
            }
            // k_score's bail-out (Prune), tested where its dense path tests it: after every 128 records and at the end of a 512-record tile
            const int processed = min(t0 + 64, n);
            if (rec_score < DBL_MAX && ((processed & 127) == 0 || processed == n) &&
                (long long)cnt + (long long)(n - processed) <= rec_cnt && score + thr * (double)(processed - cnt) >= rec_score) { pruned = true; break; }
        }
        if (lane == 0) {
            slot_score[slot_base + slot] = pruned ? DBL_MAX : score + thr * (double)(n - cnt);
            slot_inl[slot_base + slot] = pruned ? -2 : cnt;
        }
    }
}

// ------------------------------------------------------------------------------------------------ scan
// One wave per pair walks the chunk's slots in iteration order, 64 * IPL iterations per step (a lane owns IPL consecutive iterations): per-lane
// local records -> wave exclusive prefix (max count, min score) -> per-lane replay against the true running records.
// IPL (round 6): with one iteration per lane the 154 steps of a 9872-iteration chunk were 154 dependent memory round trips of one wavefront per
// SIMD (0.13 ms, on the critical path in front of the LO launch); four iterations per lane keep four times the loads in flight per step.
template <int MPS = 4, int IPL = 1>
__global__ __launch_bounds__(64) void k_scan(RunParams rp, PairState *__restrict__ st, const double *__restrict__ slot_score,
                                             const int32_t *__restrict__ slot_inl, Trigger *__restrict__ triggers,
                                             int trig_cap, const int32_t *__restrict__ model_count,
                                             unsigned long long *__restrict__ evals) {
    const int pair = blockIdx.x, lane = threadIdx.x;
    PairState &ps = st[pair];
    if (!ps.active) { if (lane == 0) ps.n_triggers = 0; return; }
    if (lane == 0 && evals) atomicAdd(evals, (unsigned long long)(model_count[2 * pair] + model_count[2 * pair + 1]) * (unsigned long long)ps.n); // survivors handed to the fp64 sweep
    long long run_cnt = (long long)ps.best_min_cnt;
    double run_score = ps.best_min_score;
    const double nan_score = (double)ps.n * ps.sq_thr;
    int ntrig = rp.chunk_off > 0 ? ps.n_triggers : 0; // later chunks of a super-chunk append to its trigger list
    static_assert(MPS % 4 == 0, "slots of an iteration are read as 16-byte groups");
    constexpr int NG = MPS / 4;
    const size_t slot_base = (size_t)pair * rp.slot_stride + (size_t)rp.chunk_off * MPS;
    // the slots of the next step are requested before this step is examined, and a step in which no lane beats the running records
    // (most of them) skips the prefix and the replay
    struct Slots { int4 c[IPL][NG]; double2 s01[IPL][NG], s23[IPL][NG]; };
    auto fetch = [&](int it0) {
        Slots r;
#pragma unroll
        for (int j = 0; j < IPL; ++j) {
            const int it = it0 + IPL * lane + j;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (it < rp.chunk_len) {
                    r.c[j][g] = *reinterpret_cast<const int4 *>(slot_inl + slot_base + (size_t)it * MPS + 4 * g);
                    const double2 *sp = reinterpret_cast<const double2 *>(slot_score + slot_base + (size_t)it * MPS + 4 * g);
                    r.s01[j][g] = sp[0]; r.s23[j][g] = sp[1]; // empty slots hold stale scores: masked by the count below
                } else { r.c[j][g] = make_int4(-1, -1, -1, -1); r.s01[j][g] = r.s23[j][g] = make_double2(DBL_MAX, DBL_MAX); }
            }
        }
        return r;
    };
    Slots nxt = fetch(0);
    for (int it0 = 0; it0 < rp.chunk_len; it0 += 64 * IPL) {
        const Slots cur = nxt;
        if (it0 + 64 * IPL < rp.chunk_len) nxt = fetch(it0 + 64 * IPL);
        int c[IPL][MPS];
        double s[IPL][MPS];
        long long lc = -1;
        double ls = DBL_MAX;
#pragma unroll
        for (int j = 0; j < IPL; ++j) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                c[j][4 * g] = cur.c[j][g].x; c[j][4 * g + 1] = cur.c[j][g].y; c[j][4 * g + 2] = cur.c[j][g].z; c[j][4 * g + 3] = cur.c[j][g].w;
                s[j][4 * g] = c[j][4 * g] >= 0 ? cur.s01[j][g].x : DBL_MAX; s[j][4 * g + 1] = c[j][4 * g + 1] >= 0 ? cur.s01[j][g].y : DBL_MAX;
                s[j][4 * g + 2] = c[j][4 * g + 2] >= 0 ? cur.s23[j][g].x : DBL_MAX; s[j][4 * g + 3] = c[j][4 * g + 3] >= 0 ? cur.s23[j][g].y : DBL_MAX;
            }
            if (c[j][0] == -3) { c[j][0] = 0; s[j][0] = nan_score; } // k_solve's NaN model of the iteration (the reference's P3P): no inlier, every residual counts thr
#pragma unroll
            for (int k = 0; k < MPS; ++k) if (c[j][k] >= 0) { lc = max(lc, (long long)c[j][k]); ls = fmin(ls, s[j][k]); }
        }
        if (__all(lc <= run_cnt && !(ls < run_score))) continue; // nothing in these iterations improves a record
        // exclusive prefix over lanes
        long long pc = lc;
        double psn = ls;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long vc = __shfl_up(pc, o, 64);
            const double vs = __shfl_up(psn, o, 64);
            if (lane >= o) { pc = max(pc, vc); psn = fmin(psn, vs); }
        }
        const long long tot_c = __shfl(pc, 63, 64);
        const double tot_s = __shfl(psn, 63, 64);
        long long ec = __shfl_up(pc, 1, 64);
        double es = __shfl_up(psn, 1, 64);
        if (lane == 0) { ec = -1; es = DBL_MAX; }
        long long bc = max(run_cnt, ec);
        double bs = fmin(run_score, es);
        // replay this lane's iterations, and each iteration's models, in order against the true running records
        int k_ref[IPL], k_min[IPL], cnt_min[IPL], cnt_ref[IPL], mine = 0;
        double score_min[IPL];
#pragma unroll
        for (int j = 0; j < IPL; ++j) {
            k_ref[j] = -1; k_min[j] = -1; cnt_min[j] = 0; cnt_ref[j] = 0; score_min[j] = 0;
#pragma unroll
            for (int k = 0; k < MPS; ++k) {
                if (c[j][k] >= 0) {
                    const bool more = (long long)c[j][k] > bc, better = s[j][k] < bs;
                    if (more || better) {
                        if (more) bc = c[j][k];
                        if (better) { bs = s[j][k]; k_min[j] = k; score_min[j] = s[j][k]; cnt_min[j] = c[j][k]; }
                        k_ref[j] = k; cnt_ref[j] = c[j][k];
                    }
                }
            }
            mine += k_ref[j] >= 0;
        }
        // triggers in iteration order: this lane's come after those of the lanes below it
        int before = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(before, o, 64); if (lane >= o) before += v; }
        const int step_total = __shfl(before, 63, 64);
        int pos = ntrig + before - mine;
#pragma unroll
        for (int j = 0; j < IPL; ++j) {
            if (k_ref[j] >= 0) {
                if (pos < trig_cap) {
                    Trigger &tr = triggers[(size_t)pair * trig_cap + pos];
                    tr.iter = (uint32_t)(rp.chunk_off + it0 + IPL * lane + j); tr.k_ref = k_ref[j]; tr.k_min = k_min[j]; tr.cnt_min = cnt_min[j]; tr.score_min = score_min[j];
                    tr.ref_score = DBL_MAX; tr.ref_cnt = 0; tr.cnt_ref = cnt_ref[j];
                }
                ++pos;
            }
        }
        ntrig += step_total;
        run_cnt = max(run_cnt, tot_c);
        run_score = fmin(run_score, tot_s);
    }
    if (lane == 0) {
        ps.best_min_cnt = (uint64_t)(run_cnt < 0 ? 0 : run_cnt);
        ps.best_min_score = run_score;
        ps.n_triggers = min(ntrig, trig_cap);
    }
}

// ------------------------------------------------------------------------------------------------ LM (workgroup)
template <int KIND, bool SHIFT> struct LmTraits;
template <> struct LmTraits<0, false> { static constexpr int NP = 7; };
template <> struct LmTraits<0, true> { static constexpr int NP = 9; };
template <> struct LmTraits<1, false> { static constexpr int NP = 8; };
template <> struct LmTraits<2, false> { static constexpr int NP = 9; };

// active parameter p -> column of the full 11-wide Jacobian
template <int KIND, bool SHIFT>
__device__ __forceinline__ constexpr int lm_col(int p) {
    return p < 7 ? p : (KIND == 0 ? p /*7,8 = shifts*/ : p + 2 /*9,10 = focals*/);
}

struct LmOpt {
    int max_it, loss;
    double loss_scale, grad_tol, step_tol, lambda0, lambda_min, lambda_max;
    double mu = 0.5; // TRUNCATED_LE_ZACH penalty strength of the current LM iteration (lm_refine advances it)
};

// Work-list LM.  With a truncated loss (every LO refinement, and the recommended TRUNCATED_CAUCHY of the final one) a
// correspondence whose three terms all have zero IRLS weight contributes nothing to J'J, but inside a wavefront the
// Jacobian code would still run for all 64 lanes whenever one lane has an inlier.  The cost sweep (which runs anyway for
// every candidate step) therefore also writes the indices of the contributing correspondences, compacted per wavefront
// with a ballot, into an LDS list; the accumulate sweep of an accepted step walks that list with every lane busy.
// Lists are double buffered (current model / candidate).  Each wavefront owns a contiguous segment of the
// correspondences, so list order — and with it the floating-point summation order — is deterministic.
// The lists live in dynamic LDS sized by the host: 2 buffers x stride u16 indices (stride = n_max rounded up to 64;
// 0 = no lists, every correspondence is visited).  T = threads per LM problem: 256 (one workgroup of 4 wavefronts, for
// latency when few problems are in flight) or 64 (one wavefront per problem: no barriers, the serial Cholesky/step part
// is paid once instead of four times — for throughput when problems outnumber SIMDs).
constexpr int LM_LIST_MAX_N = 8192; // u16 indices, <= 32 KiB of dynamic LDS (no opt-in needed)

// The LM state is the same in every lane of the workgroup (one problem per workgroup): pin it to scalar registers.
// The compiler cannot prove uniformity of values that went through vector arithmetic; readfirstlane states it, and the
// 25-34 doubles move from VGPRs to SGPRs, where VOP3 fp64 instructions read them directly.
__device__ __forceinline__ double uniform_f64(double x) {
    const unsigned long long b = __double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void lm_state_uniform(LmState &st) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { st.R[i] = uniform_f64(st.R[i]); st.E[i] = uniform_f64(st.E[i]); st.F[i] = uniform_f64(st.F[i]); }
#pragma unroll
    for (int i = 0; i < 3; ++i) st.t[i] = uniform_f64(st.t[i]);
    st.s = uniform_f64(st.s); st.u = uniform_f64(st.u); st.v = uniform_f64(st.v);
    st.f1 = uniform_f64(st.f1); st.f2 = uniform_f64(st.f2); st.if1 = uniform_f64(st.if1); st.if2 = uniform_f64(st.if2);
}

typedef __attribute__((address_space(3))) uint16_t lds_u16;
__device__ __forceinline__ lds_u16 *lds_cast(uint16_t *p) { return (lds_u16 *)p; }

static __device__ const double g_logtab[MDRP_LOGTAB_N][2] = MDRP_LOGTAB_INIT;
// the workgroup's LDS copy of the table (dst: 2 * MDRP_LOGTAB_N doubles); the caller's next barrier publishes it
__device__ __forceinline__ void lm_logtab_load(double *dst) {
    for (int i = threadIdx.x; i < 2 * MDRP_LOGTAB_N; i += blockDim.x) dst[i] = g_logtab[i >> 1][i & 1];
}

struct LmShared {
    double scratch[4 * MAX_ACC];
    int count[2][4];
    uint16_t *list;  // dynamic LDS, 2 * stride entries (+ stride more when `midx` is set)
    int stride;
    int midx;        // 1: a third list of `stride` entries behind the two work lists holds the indices of the records a record mask lets
                     // through, per wavefront segment (lm_mask_index; the inlier-only final refinement); mcount = entries per wavefront
    int mcount[4];
    const double *logtab;      // LDS copy of the log table of lm_log1p (the Cauchy losses of the final refinements), or null: library log1p
    unsigned long long *stats; // [0] correspondences evaluated by cost sweeps, [1] by accumulate sweeps (or null): bench.py's fp64 roofline
    unsigned long long ev[2];  // ... collected here per problem, flushed by lm_flush_stats
};
// one pair of global atomics per LM problem (thread 0, after the problem's last barrier)
__device__ __forceinline__ void lm_flush_stats(LmShared &sh) {
    if (sh.stats) {
        if (sh.ev[0]) atomicAdd(sh.stats, sh.ev[0]);
        if (sh.ev[1]) atomicAdd(sh.stats + 1, sh.ev[1]);
    }
    sh.ev[0] = 0; sh.ev[1] = 0;
}

#define MDRP_LM_COST_UNROLL 1 // records per lane and trip of the cost sweep (lm_cost)
// IRLS weight of the Sampson row: ws^2 w(r^2) in the calibrated refiner, ws^2 w(ws r^2) in the two focal ones (the COST carries ws rho(r^2) in all
// three) — see lm_accumulate_point.  The work lists of lm_cost and of the list engine (mdrp_lm.h) are built with the same expression.
template <int KIND>
__device__ __forceinline__ double sampson_row_weight(int loss, double lsc, double mu, double ws, double ws2, double rs) {
    return ws2 * loss_weight(loss, lsc, KIND != 0 ? ws * rs : rs, mu);
}
// LOSS: the loss type when the caller knows it at compile time (1 = TRUNCATED: every LO refinement), -1 = o.loss.
// Round 4: the loop body is straight-line — padding lanes evaluate a harmless record and every `if` of the round-3 body (record
// valid, forward / backward depth positive, loss type) is a select on the three cost terms, added in the round-3 order, so the sums
// are bit-identical to it — and it is unrolled by two with ping-pong record buffers, which removes the register rotation of the
// one-trip software pipeline (round 3: 244 instructions per trip for 123 fp64 ones; profiles/r04_*).
template <int KIND, int T, int LOSS = -1>
__device__ __forceinline__ double lm_cost(const Model &m, const double *__restrict__ pts, const double *__restrict__ dep, int n,
                          const uint8_t *__restrict__ mask, double sqrt_sr, double ws, const LmOpt &o, LmShared &sh, int buf) {
    LmState stt;
    lm_state_from_model(m, KIND != 0, stt);
    lm_state_uniform(stt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool use_list = sh.stride > 0;
    // sh.list is a pointer kept IN LDS: read back, the compiler no longer knows it points to LDS and emitted flat_store_short for the list
    // entries — and a FLAT operation may return out of order with the global loads around it, so every wait in the loop became
    // s_waitcnt vmcnt(0): the record prefetch was waited for a few instructions after it was issued and each trip paid a full memory
    // round trip (14.6 cycles per instruction against 5.2 in the normal-equation sweep; found in the ISA in round 4).  Stating the
    // address space gives ds_write_b16 and partial vmcnt waits back.
    lds_u16 *list = lds_cast(sh.list) + (size_t)buf * sh.stride;
    const int seg = ((n + T - 1) / T) * 64; // correspondences per wavefront, multiple of 64
    const int lo = wave * seg;
    // Under a record mask (the inlier-only final refinement: about half of the records) the sweep walks the wavefront's compacted index list
    // (lm_mask_index, built once per refinement) instead of evaluating every record and discarding the masked ones.
    const lds_u16 *midx = (mask && use_list && sh.midx) ? lds_cast(sh.list) + 2 * (size_t)sh.stride : nullptr;
    const int hi = midx ? lo + sh.mcount[wave] : min(n, lo + seg); // end of this wavefront's trips (list positions under a mask index)
    const int loss = LOSS >= 0 ? LOSS : o.loss;
    const double lsc = o.loss_scale, mu = o.mu, t2 = lsc * lsc;
    const double *logtab = sh.logtab;
    const bool ws_nz = ws * ws != 0.0; // the Sampson row's weight carries ws^2 (lm_accumulate_point)
    const unsigned long long lt = (1ull << lane) - 1ull;
    double cost = 0;
    int cnt = 0, evaluated = 0;
    struct Rec { double a, b, c, d, e1, e2; int id; bool ok; };
    const int last = max(n - 1, 0);
    auto fetch = [&](int base) { // unconditional loads from a clamped index: no exec-mask branch around them; `ok` says whether the lane counts
        Rec r;
        const int i = base + lane;
        int ic;
        if (midx) { r.ok = i < hi; ic = r.ok ? (int)midx[i] : last; }
        else { ic = min(i, last); r.ok = (int)(i < hi) & (int)(mask ? mask[ic] != 0 : true); } // (integer AND on purpose, here and below: no short-circuit branches)
        r.id = ic;
        const double2 *P = reinterpret_cast<const double2 *>(pts + (size_t)ic * PT_STRIDE);
        const double2 p01 = P[0], p23 = P[1];
        const double2 dd = *reinterpret_cast<const double2 *>(dep + 2 * (size_t)ic);
        r.a = p01.x; r.b = p01.y; r.c = p23.x; r.d = p23.y; r.e1 = dd.x; r.e2 = dd.y;
        return r;
    };
    // A trip is U records per lane, in two phases.  Phase 1 evaluates the U residual chains in one straight-line block: a chain is ~30
    // dependent fp64 operations deep (rotations, three reciprocals with their Newton steps, the Sampson denominator), a wavefront issues in
    // order, and the branches of the list append used to end the scheduler's block after every record.  Phase 2 adds the terms and appends
    // the list entries in record order: sums and lists are those of the one-record-at-a-time loop bit for bit.
    constexpr int U = MDRP_LM_COST_UNROLL;
    struct Trip { Rec r[U]; };
    auto fetch_trip = [&](int base) {
        Trip t;
#pragma unroll
        for (int u = 0; u < U; ++u) t.r[u] = fetch(base + 64 * u);
        return t;
    };
    auto step_trip = [&](const Trip &t, int base) {
        double vs[U], vf[U], vb[U];
        bool fwd[U], bwd[U], contrib[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const Rec &cur = t.r[u];
            double r[5], zf, zb;
            point_residuals<false, KIND != 0>(stt, sqrt_sr, cur.a, cur.b, cur.c, cur.d, cur.e1, cur.e2, r, zf, zb, nullptr);
            const double rs = r[0] * r[0], rf = r[1] * r[1] + r[2] * r[2], rb = r[3] * r[3] + r[4] * r[4];
            const int oki = cur.ok;
            fwd[u] = oki & (int)(zf > 0); bwd[u] = oki & (int)(zb > 0); // positive depth of the transferred point: NaN depths drop the term like negative ones (reference)
            // The Sampson row's IRLS weight is ws^2 w(r^2) in the calibrated refiner and ws^2 w(ws r^2) in the two focal ones, while the COST
            // carries ws rho(r^2) in all three (lm_accumulate_point): the work list follows the weight, so its argument is `ra`, not `rs`.
            const double ra = KIND != 0 ? ws * rs : rs;
            if (LOSS == 1) { // TRUNCATED: min(r^2, t^2); IRLS weight 1 below the threshold, 0 at and above it (and for NaN)
                // value: std::min(r^2, t^2) as the reference evaluates it — a NaN residual comes back NaN and makes the cost NaN (the LM then accepts
                // nothing).  One predicate serves the value and the work list: a NaN row on the list has weight 0 in the accumulate sweep.
                const bool is = !(rs >= t2), ia = KIND != 0 ? !(ra >= t2) : is, jf = !(rf >= t2), jb = !(rb >= t2);
                vs[u] = ws * (is ? rs : t2); vf[u] = jf ? rf : t2; vb[u] = jb ? rb : t2;
                contrib[u] = ((oki & (int)ia & (int)ws_nz) | ((int)fwd[u] & (int)jf) | ((int)bwd[u] & (int)jb)) != 0;
            } else {
                vs[u] = ws * loss_value_tab(loss, lsc, rs, logtab); vf[u] = loss_value_tab(loss, lsc, rf, logtab); vb[u] = loss_value_tab(loss, lsc, rb, logtab);
                if (LOSS == 3 || LOSS == 4) {
                    // the Cauchy weights 1 / (1 + r^2 / t^2), floored at DBL_MIN, are never zero (NaN included: the floor takes it); TRUNCATED_CAUCHY's is
                    // zero exactly at and beyond the threshold — the same predicate as `loss_weight(...) != 0` without three reciprocal chains
                    const bool ia = LOSS == 3 || ra < t2, jf = LOSS == 3 || rf < t2, jb = LOSS == 3 || rb < t2;
                    contrib[u] = ((oki & (int)ia & (int)ws_nz) | ((int)fwd[u] & (int)jf) | ((int)bwd[u] & (int)jb)) != 0;
                } else
                    contrib[u] = ((oki & (int)(sampson_row_weight<KIND>(loss, lsc, mu, ws, ws * ws, rs) != 0.0)) | ((int)fwd[u] & (int)(loss_weight(loss, lsc, rf, mu) != 0.0)) |
                                  ((int)bwd[u] & (int)(loss_weight(loss, lsc, rb, mu) != 0.0))) != 0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cost += t.r[u].ok ? vs[u] : 0.0; // (+ 0.0 leaves a non-negative sum as it is: the order and the values of round 3's `if`s)
            cost += fwd[u] ? vf[u] : 0.0;
            cost += bwd[u] ? vb[u] : 0.0;
            if (mask && !midx && sh.stats) evaluated += __popcll(__ballot(t.r[u].ok));
            if (use_list) {
                const unsigned long long ball = __ballot(contrib[u]);
                if (contrib[u]) list[lo + cnt + __popcll(ball & lt)] = (uint16_t)t.r[u].id;
                cnt += __popcll(ball);
            }
        }
    };
    if (!mask || midx) evaluated = max(hi - lo, 0);
    // the next trip's records are requested before the current ones are consumed; two trips per loop iteration, the buffers swap roles
    // instead of being copied (a trip past `hi` is all padding: loads from a clamped index, nothing counted).  Requesting two trips ahead
    // (four rotating buffers) was measured 1-2 % slower.
    Trip A = fetch_trip(lo);
    for (int base = lo; base < hi; base += 128 * U) {
        const Trip B = fetch_trip(base + 64 * U);
        step_trip(A, base);
        A = fetch_trip(base + 128 * U);
        if (base + 64 * U < hi) step_trip(B, base + 64 * U);
    }
    if (use_list && lane == 0) sh.count[buf][wave] = cnt;
    if (sh.stats && lane == 0 && evaluated) atomicAdd(&sh.ev[0], (unsigned long long)evaluated);
    double v[1] = {cost};
    block_sum<1, T>(v, sh.scratch);
    return v[0];
}

// JtJ (lower triangle, row-major) and Jtr of one weighted residual row.  ZMASK: columns of the full 11-wide row that are
// structurally zero for this term (no shift / scale / cross-translation dependence); their products are skipped — the
// compiler cannot drop `acc += w * 0.0 * x` on its own under IEEE rules (18-22 % of the normal-equation FMAs).
// UNIT: the weight is known to be exactly 1 (a TRUNCATED loss below its threshold): w * J is J bit for bit, so the NP products are skipped
template <int KIND, bool SHIFT, unsigned ZMASK, bool UNIT = false>
__device__ __forceinline__ void lm_accumulate_row(const double *__restrict__ Jrow, double r, double w, double *acc) {
    constexpr int NP = LmTraits<KIND, SHIFT>::NP;
    double Ja[NP];
    bool nz[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int col = lm_col<KIND, SHIFT>(q);
        Ja[q] = Jrow[col];
        nz[q] = !((ZMASK >> col) & 1u);
        if (KIND == 1 && q == 7) { Ja[q] += Jrow[10]; nz[q] = nz[q] || !((ZMASK >> 10) & 1u); } // shared focal: f1 = f2 = f
    }
    int idx = 0;
#pragma unroll
    for (int a = 0; a < NP; ++a) {
        const double wa = UNIT ? Ja[a] : w * Ja[a];
#pragma unroll
        for (int b = 0; b <= a; ++b, ++idx)
            if (nz[a] && nz[b]) acc[idx] += wa * Ja[b];
    }
#pragma unroll
    for (int a = 0; a < NP; ++a)
        if (nz[a]) acc[NP * (NP + 1) / 2 + a] += UNIT ? Ja[a] * r : w * Ja[a] * r;
}

// One correspondence of the accumulate sweep, term by term: each term's Jacobian rows are folded into the accumulators
// before the next term is computed, so at most two rows need to be live beside the NP (NP + 3) / 2 accumulators.
// A row of weight ZERO is skipped, as the reference's accumulators do (`if (weight == 0.0) continue`): a non-finite Jacobian row of a dropped term (the
// backward term of a model whose scale is NaN, a NaN depth) must not reach the sums as 0 * NaN.  (Rounds 3-5 added `0 * (r + sum of the row)` for such
// rows to reproduce the stalled LM of tests/golden/initial.npz case 11; what stalls the reference there is its NaN COST — lm_cost hands NaN residuals on
// now — and the reference does refine models with a NaN scale, tests/golden/bad_inputs_ref.npz.)
// LOSS == 1 (TRUNCATED, known at compile time: every LO refinement): the weights of the reprojection terms are exactly 0 or 1.  A row
// of weight 1 is accumulated without the NP products by the weight (w J = J bit for bit).  Same sums bit for bit as the general path.
template <int KIND, bool SHIFT, int LOSS = -1>
__device__ __forceinline__ void lm_accumulate_point(const LmState &stt, double2 p01, double2 p23, double2 dd,
                                                    double sqrt_sr, double ws, double ws2 /* ws * ws, uniform */, const LmOpt &o, double *acc) {
    const int loss = LOSS >= 0 ? LOSS : o.loss;
    {
        double r0, J0[LM_NPAR];
        lm_sampson_term<true, KIND != 0>(stt, p01.x, p01.y, p23.x, p23.y, r0, J0);
        // weight_sampson enters the normal equations SQUARED (the cost carries it to the first power), and the focal refiners evaluate the loss
        // weight at ws r^2 where the calibrated one evaluates it at r^2: what the reference binary computes (oracle/orc_refine.c lm_accumulate,
        // fitted against refine_monodepth_*relpose for ws = 0.3 ... 3 and all six losses); every form coincides at ws = 1
        const double w = sampson_row_weight<KIND>(loss, o.loss_scale, o.mu, ws, ws2, r0 * r0);
        if (w != 0.0) lm_accumulate_row<KIND, SHIFT, 0x1C0u>(J0, r0, w, acc); // no scale / shift dependence (ws is a run-time weight: its product stays)
    }
    {
        double r1, r2, zf, J1[LM_NPAR], J2[LM_NPAR];
        lm_forward_term<true, KIND != 0>(stt, sqrt_sr, p01.x, p01.y, p23.x, p23.y, dd.x, r1, r2, zf, J1, J2);
        // positive depth of the transferred point: a NaN depth drops the term like a non-positive one (reference; the cost sweep does the same)
        const double w = !(zf > 0) ? 0.0 : loss_weight(loss, o.loss_scale, r1 * r1 + r2 * r2, o.mu);
        if (w != 0.0) {
            lm_accumulate_row<KIND, SHIFT, 0x150u, LOSS == 1>(J1, r1, w, acc); // t.y, scale, shift2
            lm_accumulate_row<KIND, SHIFT, 0x148u, LOSS == 1>(J2, r2, w, acc); // t.x, scale, shift2
        }
    }
    {
        double r3, r4, zb, J3[LM_NPAR], J4[LM_NPAR];
        lm_backward_term<true, KIND != 0>(stt, sqrt_sr, p01.x, p01.y, p23.x, p23.y, dd.y, r3, r4, zb, J3, J4);
        const double w = !(zb > 0) ? 0.0 : loss_weight(loss, o.loss_scale, r3 * r3 + r4 * r4, o.mu);
        if (w != 0.0) {
            lm_accumulate_row<KIND, SHIFT, 0x080u, LOSS == 1>(J3, r3, w, acc); // shift1
            lm_accumulate_row<KIND, SHIFT, 0x080u, LOSS == 1>(J4, r4, w, acc); // shift1
        }
    }
}

template <int KIND, bool SHIFT, int T, int LOSS = -1>
__device__ __forceinline__ void lm_accumulate(const Model &m, const double *__restrict__ pts, const double *__restrict__ dep, int n,
                              const uint8_t *__restrict__ mask, double sqrt_sr, double ws, double ws2, const LmOpt &o, double *acc, LmShared &sh, int buf) {
    constexpr int NP = LmTraits<KIND, SHIFT>::NP;
    constexpr int NA = NP * (NP + 1) / 2 + NP;
    LmState stt;
    lm_state_from_model(m, KIND != 0, stt);
    lm_state_uniform(stt);
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = ((n + T - 1) / T) * 64;
    const int lo = wave * seg;
    if (sh.stride > 0) {
        const lds_u16 *list = lds_cast(sh.list) + (size_t)buf * sh.stride; // (LDS, stated: see lm_cost)
        const int cnt = sh.count[buf][wave];
        // software-pipelined by one step like the cost sweep: list entry and record of the next trip are requested
        // before the current record's ~700 fp64 ops (an unhidden LDS + L2 round trip was ~30 % of the sweep: PMC SQ_WAIT_ANY)
        double2 n01 = make_double2(0, 0), n23 = n01, ndd = n01;
        auto fetch = [&](int k) {
            if (k < cnt) {
                const size_t i = (size_t)list[lo + k];
                const double2 *P = reinterpret_cast<const double2 *>(pts + i * PT_STRIDE);
                n01 = P[0]; n23 = P[1];
                ndd = *reinterpret_cast<const double2 *>(dep + 2 * i);
            }
        };
        fetch(lane);
        for (int k = lane; k < cnt; k += 64) {
            const double2 c01 = n01, c23 = n23, cdd = ndd;
            fetch(k + 64);
            lm_accumulate_point<KIND, SHIFT, LOSS>(stt, c01, c23, cdd, sqrt_sr, ws, ws2, o, acc);
        }
        if (sh.stats && lane == 0 && cnt) atomicAdd(&sh.ev[1], (unsigned long long)cnt);
    } else {
        const int hi = min(n, lo + seg);
        for (int i = lo + lane; i < hi; i += 64)
            if (!mask || mask[i]) {
                const double2 *P = reinterpret_cast<const double2 *>(pts + (size_t)i * PT_STRIDE);
                lm_accumulate_point<KIND, SHIFT, LOSS>(stt, P[0], P[1], *reinterpret_cast<const double2 *>(dep + 2 * (size_t)i), sqrt_sr, ws, ws2, o, acc);
            }
    }
    block_sum<NA, T>(acc, sh.scratch);
}

// indices of the records a mask lets through, compacted per wavefront segment in record order (the segments of lm_cost / lm_accumulate):
// the third list behind the two work lists.  The inlier-only final refinement sweeps ~half of the records up to 100 times.
template <int T>
__device__ __forceinline__ void lm_mask_index(const uint8_t *__restrict__ mask, int n, LmShared &sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = ((n + T - 1) / T) * 64;
    const int lo = wave * seg, hi = min(n, lo + seg);
    lds_u16 *midx = lds_cast(sh.list) + 2 * (size_t)sh.stride;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int cnt = 0;
    for (int base = lo; base < hi; base += 64) {
        const int i = base + lane;
        const bool in = i < hi && mask[i] != 0;
        const unsigned long long ball = __ballot(in);
        if (in) midx[lo + cnt + __popcll(ball & lt)] = (uint16_t)i;
        cnt += __popcll(ball);
    }
    if (lane == 0) sh.mcount[wave] = cnt;
    __syncthreads();
}

// lm_impl<> loop of the reference (upstream PoseLib convention): executed redundantly and uniformly by every
// thread of the workgroup; only the two sweeps over the correspondences are distributed.
// Round 4: ONE call site each for the cost and the normal-equation sweep (the initial cost is the first trip of the loop), and
// lm_refine itself is inlined into its kernel: every sweep is compiled under the kernel's launch bounds.  (With two call sites the
// sweeps stayed separate functions, compiled without the bounds: 256 VGPRs + 44-70 AGPRs = one wavefront per SIMD.)
template <int KIND, bool SHIFT, int T, int LOSS = -1>
__device__ __forceinline__ void lm_refine(Model &m, const double *__restrict__ pts, const double *__restrict__ dep, int n,
                                          const uint8_t *__restrict__ mask, double scale_reproj, double ws, const LmOpt &o_in, LmShared &sh) {
    constexpr int NP = LmTraits<KIND, SHIFT>::NP;
    constexpr int NT = NP * (NP + 1) / 2;
    const double sqrt_sr = sqrt(scale_reproj);
    const double ws2 = ws * ws; // the Sampson row's weight carries ws^2 (lm_accumulate_point)
    LmOpt o = o_in;
    o.mu = 0.5;
    int cur = 0; // list buffer that belongs to the current model
    double cost = 0;
    double lambda = o.lambda0;
    bool recompute = true, first = true;
    double acc[NT + NP];
    double A[NP * NP], g[NP], sol[NP];
    Model cand = m;
    int it = 0;
    if (mask && sh.stride > 0 && sh.midx) lm_mask_index<T>(mask, n, sh);
#pragma unroll 1
    for (;;) {
        // cost of the model under evaluation: the start model on the first trip (into list buffer `cur`), a candidate step afterwards
        const double cost_new = lm_cost<KIND, T, LOSS>(cand, pts, dep, n, mask, sqrt_sr, ws, o, sh, first ? cur : cur ^ 1);
        if (first) { cost = cost_new; first = false; }
        else {
            if (cost_new < cost) {
                m = cand;
                cur ^= 1;
                lambda = fmax(o.lambda_min, lambda / 10.0);
                cost = cost_new;
                recompute = true;
            } else {
                lambda = fmin(o.lambda_max, lambda * 10.0);
                recompute = false;
            }
            o.mu *= 1.5; // the reference's per-iteration callback of TRUNCATED_LE_ZACH; note the work list of the current model
                         // was built with the previous mu — Le-Zach weights are never zero, so the list holds every record
            ++it;
        }
        if (it >= o.max_it) break;
        if (recompute) {
            lm_accumulate<KIND, SHIFT, T, LOSS>(m, pts, dep, n, mask, sqrt_sr, ws, ws2, o, acc, sh, cur);
            double gn = 0;
            int idx = 0;
#pragma unroll
            for (int a = 0; a < NP; ++a) {
#pragma unroll
                for (int b = 0; b <= a; ++b) A[a * NP + b] = acc[idx++];
            }
#pragma unroll
            for (int a = 0; a < NP; ++a) { g[a] = acc[NT + a]; gn += g[a] * g[a]; }
            if (sqrt(gn) < o.grad_tol) break;
        }
        double Ad[NP * NP];
#pragma unroll
        for (int a = 0; a < NP; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) Ad[a * NP + b] = A[a * NP + b] + (a == b ? lambda : 0.0);
        chol_solve<NP>(Ad, g, sol);
        double sn = 0;
#pragma unroll
        for (int a = 0; a < NP; ++a) { sol[a] = -sol[a]; sn += sol[a] * sol[a]; }
        if (sqrt(sn) < o.step_tol) break;
        double full[LM_NPAR];
#pragma unroll
        for (int q = 0; q < LM_NPAR; ++q) full[q] = 0;
#pragma unroll
        for (int q = 0; q < NP; ++q) full[lm_col<KIND, SHIFT>(q)] = sol[q];
        if (KIND == 1) full[10] = full[9];
        lm_apply_step(m, full, KIND != 0, KIND == 0 && SHIFT, cand);
    }
}

// workgroup-wide exact MSAC score of one model (score_model of the estimators); optional inlier mask output
template <int T>
__device__ __forceinline__ void block_score(int kind, const Model &m, const double *__restrict__ pts, int n, double thr, double *scratch,
                            double &score_out, int &cnt_out, uint8_t *__restrict__ mask_out) {
    double R[9], E[9], Em[9];
    quat_to_R(m.q, R);
    essential_from_Rt(R, m.t, Em);
    if (kind == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) E[i] = Em[i];
    } else fundamental_from_E(Em, m.f1, m.f2, E);
    double score = 0;
    int cnt = 0;
    // one record of look-ahead (all 48 bytes, unconditional loads from a clamped index): round 4 loaded a record, waited for it, and loaded
    // its inverse norms behind the inlier test — two exposed memory round trips per trip, 32 trips per LO problem at one wavefront (found
    // in the ISA in round 5: most of the 55 us a problem spent outside its LM).  Same arithmetic, same order: bit-identical.
    const int last = n > 0 ? n - 1 : 0;
    double2 nx0 = make_double2(0, 0), nx1 = nx0, nx2 = nx0;
    auto fetch = [&](int i) {
        const double2 *P = reinterpret_cast<const double2 *>(pts + (size_t)(i < last ? i : last) * PT_STRIDE);
        nx0 = P[0]; nx1 = P[1]; nx2 = P[2];
    };
    fetch(threadIdx.x);
    for (int i = threadIdx.x; i < n; i += T) {
        const double rec[6] = {nx0.x, nx0.y, nx1.x, nx1.y, nx2.x, nx2.y};
        fetch(i + T);
        double s1 = 0;
        int c1 = 0;
        if (kind == 0) score_point<true>(rec, E, R, m.t, thr, s1, c1);
        else score_point<false>(rec, E, R, m.t, thr, s1, c1);
        score += s1; cnt += c1;
        if (mask_out) mask_out[i] = (uint8_t)c1;
    }
    double v[2] = {score, (double)cnt};
    block_sum<2, T>(v, scratch);
    cnt_out = (int)v[1];
    score_out = v[0] + thr * (double)(n - cnt_out);
}

// ------------------------------------------------------------------------------------------------ LO
// Persistent workgroups pop (pair, trigger) items; each refines the triggering minimal model (refine_model
// @0x4fa550/@0x4fad60/@0x4fb0a0: 25 it, TRUNCATED) and rescoring it.
#define MDRP_LM_MINWAVES 2
// LO work plan of one chunk: the triggers the chunk's scan appended to each pair's list, as a prefix sum over the pairs.
// The plan is frozen when it is built (begin/end per pair), so k_lo of this chunk can run on a second stream while the
// next chunk's scan keeps appending triggers — the LO of the first, short chunk (most of a run's LO work: records fall
// fast at the start) hides behind the second chunk's sweep instead of leaving the chip to the tail of a launch whose
// single problems take ~1 ms on one wavefront.
// (Measured and dropped: longest-first ordering by inlier count — no change.  XCD-affine queues, pair p on XCD p mod 8: 2.8x less HBM
// fetch, same time in round 2; as contiguous eighths of the plan with stealing (lo_take, round 4) 2-5 % of k_lo: the LO is bound by
// instruction issue at 1-2 waves/SIMD, not by bandwidth or order.)
// plan layout: prefix[B+1] | begin[B] | end[B] | total
MDRP_GLOBAL __launch_bounds__(PLAN_THREADS) void k_lo_plan(int batch, const PairState *__restrict__ st, const int32_t *__restrict__ prev_plan /*or null*/,
                                                          int32_t *__restrict__ plan) {
    __shared__ int s_w[PLAN_THREADS / 64 + 1];
    int32_t *prefix = plan, *begin = plan + batch + 1, *end = begin + batch;
    const int32_t *prev_end = prev_plan ? prev_plan + 2 * (size_t)batch + 1 : nullptr;
    int run = 0;
    for (int p0 = 0; p0 < batch; p0 += PLAN_THREADS) {
        const int p = p0 + threadIdx.x;
        int b = 0, e = 0;
        if (p < batch) { b = prev_end ? prev_end[p] : 0; e = st[p].n_triggers; if (e < b) e = b; }
        int tot;
        const int ex = plan_block_scan(e - b, tot, s_w);
        if (p < batch) { prefix[p] = run + ex; begin[p] = b; end[p] = e; }
        run += tot;
    }
    if (threadIdx.x == 0) { prefix[batch] = run; plan[3 * (size_t)batch + 1] = run; }
}

// ------------------------------------------------------------------------------------------------ walk
// One lane per pair replays score_models<> / ransac<> bookkeeping (@0x22ebc0, @0x22f030) over the ordered triggers
// and applies the dynamic stopping rule.
// lo_plans / n_plans / lo_cap: the LM engine refines at most lo_cap triggers of a chunk per pass (its problem table is sized
// before anyone knows how many triggers the scans will find); if a chunk found more, nothing is replayed yet — the flag
// n_active[1] tells the host to run the remaining passes and launch the walk again.
// Replay of one pair (one lane).  Returns true if the pair has stopped; otherwise `need` = iterations it still certainly needs.
// COHERENT: the LO results in the triggers were written by other workgroups of the SAME launch (fused tail): they are read with
// agent-scope atomic loads, which neither the compiler (restrict / invariance reasoning) nor the scalar cache can serve from
// anything older than the acquire that preceded the call.
__device__ __forceinline__ double coherent_f64(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// (uint64_t) of a double as the reference's x86-64 build computes it (comisd against 2^63; cvttsd2si; the high half via d - 2^63 and a flipped top
// bit): what the dynamic iteration bound becomes when it leaves the range — success_prob = 1 gives +inf -> 0 (the search ends right after
// min_iterations), NaN (success_prob > 1) gives 2^63 (never), a negative bound wraps.  v_cvt saturates instead.  tests/golden/edge_options_ref.npz
__device__ __forceinline__ uint64_t f64_to_u64_x86(double d) {
    const double t63 = 9223372036854775808.0;
    if (d >= t63) {
        const double e = d - t63;
        return (e < t63 ? (uint64_t)(int64_t)e : 0x8000000000000000ull) ^ 0x8000000000000000ull;
    }
    return d >= -t63 ? (uint64_t)(int64_t)d : 0x8000000000000000ull; // (NaN fails both comparisons: cvttsd2si's "integer indefinite")
}
template <bool COHERENT = false>
__device__ bool walk_pair(const RunParams &rp, PairState &ps, const Model *__restrict__ models, const Trigger *trig /*of this pair*/,
                          size_t slot_base, uint64_t &need) {
    need = 0;
    if (!ps.active) return true;
    // max_iterations = 0: the reference's loop head ends the search before a sample is drawn (ransac<>: iterations < max_iterations) — no record, no
    // LO; the closing LO then runs on the reset identity model (tests/golden/edge_options_ref.npz)
    if (rp.max_iterations == 0) { ps.iterations = 0; ps.active = 0; return true; }
    const uint64_t c0 = rp.chunk_start, c1 = rp.chunk_start + (uint64_t)rp.super_len;
    uint64_t it = c0; // iterations completed so far
    bool stopped = false;
    // stop test applied after each completed iteration: it >= max -> stop; it > min && it > dyn -> stop
    auto first_stop = [&](uint64_t lo /*first candidate value of it*/) -> uint64_t {
        uint64_t s = lo;
        if (s < ps.dyn_max_iter + 1) s = ps.dyn_max_iter + 1;
        if (s < rp.min_iterations + 1) s = rp.min_iterations + 1;
        if (s > rp.max_iterations) s = rp.max_iterations;
        if (lo > s) s = lo;
        return s;
    };
    for (int k = 0; k < ps.n_triggers && !stopped; ++k) {
        const Trigger &tr = trig[k];
        const uint64_t ti = c0 + tr.iter; // absolute index of the triggering iteration
        // iterations it .. ti-1 complete without bookkeeping changes; would the loop stop at a value in (it, ti] ?
        if (ti > it) {
            const uint64_t s = first_stop(it + 1);
            if (s <= ti) { it = s; stopped = true; break; }
        }
        // execute iteration ti
        if (tr.k_min >= 0 && tr.score_min < ps.model_score) {
            ps.model_score = tr.score_min;
            ps.best = models[slot_base + (size_t)tr.iter * rp.mps + tr.k_min];
            ps.num_inliers = (uint64_t)tr.cnt_min;
        }
        ps.refinements++;
        const double ref_score = COHERENT ? coherent_f64(&tr.ref_score) : tr.ref_score;
        if (ref_score < ps.model_score) {
            ps.model_score = ref_score;
            if (COHERENT) {
                ps.num_inliers = (uint64_t)__hip_atomic_load(&tr.ref_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                static_assert(sizeof(Model) % 8 == 0, "Model is copied as doubles");
                const double *src = reinterpret_cast<const double *>(&tr.refined);
                double *dst = reinterpret_cast<double *>(&ps.best);
#pragma unroll
                for (int q = 0; q < (int)(sizeof(Model) / 8); ++q) dst[q] = coherent_f64(src + q);
            } else {
                ps.num_inliers = (uint64_t)tr.ref_cnt;
                ps.best = tr.refined;
            }
        }
        ps.inlier_ratio = (double)ps.num_inliers / (double)ps.n;
        if (ps.inlier_ratio >= 0.9999) ps.dyn_max_iter = rp.min_iterations;
        else if (ps.inlier_ratio <= 0.0001) ps.dyn_max_iter = rp.max_iterations;
        else {
            // pow(ratio, sample size) of the reference: x * x * x for 3 (pinned against the binary), libm pow otherwise
            const double prob_outlier = 1.0 - (rp.sample_sz == 3 ? ps.inlier_ratio * ps.inlier_ratio * ps.inlier_ratio : pow(ps.inlier_ratio, (double)rp.sample_sz));
            ps.dyn_max_iter = f64_to_u64_x86(ceil(rp.log_prob_missing / log(prob_outlier) * rp.dyn_mult));
        }
        it = ti + 1;
        if (it >= rp.max_iterations || (it > rp.min_iterations && it > ps.dyn_max_iter)) { stopped = true; break; }
    }
    if (!stopped) {
        // remaining iterations of the chunk
        if (c1 > it) {
            const uint64_t s = first_stop(it + 1);
            if (s <= c1) { it = s; stopped = true; } else it = c1;
        }
    }
    ps.iterations = it;
    if (stopped) { ps.active = 0; return true; }
    need = first_stop(it + 1) - it; // iterations still certainly needed with the current dyn_max_iter
    return false;
}

MDRP_GLOBAL void k_walk(RunParams rp, PairState *__restrict__ st, const Model *__restrict__ models, const Trigger *__restrict__ triggers,
                       int trig_cap, int32_t *__restrict__ n_active, unsigned long long *__restrict__ max_needed,
                       const int32_t *__restrict__ lo_plans, int n_plans, int plan_stride, int lo_cap) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= rp.batch) return;
    for (int c = 0; c < n_plans; ++c)
        if (lo_plans[(size_t)c * plan_stride + 3 * (size_t)rp.batch + 1] > lo_cap) {
            if (pair == 0) n_active[1] = 1;
            return;
        }
    PairState &ps = st[pair];
    if (!ps.active) return;
    uint64_t need;
    if (!walk_pair<false>(rp, ps, models, triggers + (size_t)pair * trig_cap, (size_t)pair * rp.slot_stride, need)) {
        atomicAdd(n_active, 1);
        atomicMax(max_needed, (unsigned long long)need);
    }
}

// Fused tail (the LAST LO launch of a run whose end is known): the workgroup that refines the last open trigger of a
// pair replays the pair (walk_pair, instead of a k_walk launch) and appends it to `ready`.  When the LO queue is empty - its last
// problems are in flight, most wavefront slots are already free - k_gate lets the final refinements start on another stream:
// workgroup i of k_final takes the i-th ready pair.  The few whose pair is not ready yet wait for problems that resident LO
// workgroups hold (nobody needs their slots: the queue is empty), so the wait cannot block anything.
struct FuseTail {
    int32_t *done_cnt;   // [batch] refined triggers of this launch per pair (zeroed)
    int32_t *ready;      // [batch] pairs in the order they became ready (-1 = not yet)
    int32_t *ctl;        // [0] entries of `ready`, [1] LO workgroups that have started (their trigger-less pairs are published)   (zeroed)
    PairState *st;       // mutable alias of the pair states (only the walking lane writes)
};
// one lane: returns when every workgroup of the LO launch has started and every problem has been taken from its queue - from then
// on whatever a final refinement may wait for is in the hands of a resident workgroup.
// Both this wait and the final refinements' wait for their pair are BOUNDED (`ticks` of the 100 MHz clock): where kernels of
// different streams cannot run side by side (rocprofv3 --pmc serialises dispatches, debuggers do) the LO launch may be stuck behind
// the very kernels that wait for it.  Then the gate gives up, the final workgroups give up and leave their pairs undone, the LO
// launch runs, and the ordinary k_final pass behind it (`skip`) refines what is left: slower, never stuck.
// LO queue with XCD affinity.  The problems of a launch are sorted by pair (k_lo_plan) and every LM sweep re-reads its pair's records
// (96 KB at N = 2000, 240 KB at N = 5000; ~18 sweeps per problem): with one queue the ~5 problems of a pair run on five different XCDs and
// every sweep comes over the fabric.  Eight queues, one per XCD, each over a contiguous eighth of the plan: an XCD's wavefronts work on ~46
// consecutive pairs at a time and the problems of a pair share its L2.  The LO is bound by instruction issue, not by the memory side
// (DESIGN.md 4): this buys 2-5 % of k_lo, no more.  A workgroup whose own queue is empty takes from the others (the tail stays balanced);
// which problem runs where changes nothing but speed.  `head` counts the tickets handed out (k_gate waits on it); xheads == null: one queue.
constexpr int LO_XCD_STRIDE = 16; // int32 between two queue heads (their own 64-byte blocks)
__device__ __forceinline__ int lo_xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7;
}
__device__ __forceinline__ int lo_take(int32_t *__restrict__ head, int32_t *__restrict__ xheads, int total) {
    if (!xheads) return atomicAdd(head, 1);
    const int x = lo_xcc_id();
    for (int k = 0; k < 8; ++k) {
        const int y = (x + k) & 7;
        const int b = (int)((long long)total * y / 8), len = (int)((long long)total * (y + 1) / 8) - b;
        if (len <= 0 || __hip_atomic_load(xheads + y * LO_XCD_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= len) continue;
        const int t = atomicAdd(xheads + y * LO_XCD_STRIDE, 1);
        if (t < len) { atomicAdd(head, 1); return b + t; }
    }
    return total;
}

MDRP_GLOBAL void k_gate(const int32_t *__restrict__ lo_head, const int32_t *__restrict__ plan_total, const int32_t *__restrict__ ctl, int lo_blocks,
                       unsigned long long ticks, unsigned long long *__restrict__ timeouts /*[0] gate, [1] final waits: mdrp_stats.fuse_*_timeouts*/) {
    const int total = *plan_total;
    const unsigned long long t0 = wall_clock64();
    bool open = false;
    while (!(open = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= lo_blocks &&
                    __hip_atomic_load(lo_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= total) && wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(64);
    if (!open && timeouts) atomicAdd(timeouts, 1ull);
}
__device__ __forceinline__ void fuse_publish(const FuseTail &fz, const RunParams &rp, int pair, const Model *__restrict__ models,
                                             const Trigger *__restrict__ triggers, int trig_cap) {
    uint64_t need;
    walk_pair<true>(rp, fz.st[pair], models, triggers + (size_t)pair * trig_cap, (size_t)pair * rp.slot_stride, need);
    __threadfence();
    const int t = atomicAdd(fz.ctl, 1);
    __hip_atomic_store(fz.ready + t, pair, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// LO of item w of the launch's plan (refine_model + score_model of the refined model), by the whole workgroup
template <int KIND, bool SHIFT, int T>
__device__ void lo_problem(const RunParams &rp, const PairState *__restrict__ st, const double *__restrict__ pts, const double *__restrict__ dep,
                           const Model *__restrict__ models, Trigger *__restrict__ triggers, int trig_cap, const int32_t *__restrict__ plan, int w,
                           LmShared &sh, const FuseTail &fz) {
    const int32_t *prefix = plan, *begin = plan + rp.batch + 1, *end = begin + rp.batch;
    const int pair = plan_find(prefix, rp.batch, w);
    const int pos = begin[pair] + (w - prefix[pair]);
    const PairState &ps = st[pair];
    Trigger &tr = triggers[(size_t)pair * trig_cap + pos];
    const size_t slot_base = (size_t)pair * rp.slot_stride;
    Model m = models[slot_base + (size_t)tr.iter * rp.mps + tr.k_ref];
    LmOpt o;
    o.max_it = 25; o.loss = 1; o.loss_scale = ps.lo_loss_scale;
    o.grad_tol = 1e-10; o.step_tol = 1e-8; o.lambda0 = 1e-3; o.lambda_min = 1e-10; o.lambda_max = 1e10;
    const double *pp = pts + (size_t)pair * rp.n_max * PT_STRIDE;
    const double *dd = dep + (size_t)pair * rp.n_max * 2;
    // (a NaN model — the reference's P3P, k_solve — has a NaN cost: no step can be accepted, it comes back as it is)
    if (m.q[0] == m.q[0]) lm_refine<KIND, SHIFT, T, 1>(m, pp, dd, ps.n, nullptr, ps.scale_reproj, rp.weight_sampson, o, sh); // refine_model: always TRUNCATED
    double sc;
    int cn;
    block_score<T>(KIND, m, pp, ps.n, ps.sq_thr, sh.scratch, sc, cn, nullptr);
    if (threadIdx.x == 0) {
        tr.refined = m; tr.ref_score = sc; tr.ref_cnt = cn; lm_flush_stats(sh);
        if (fz.ready) {
            __threadfence(); // this trigger's results before the count
            if (atomicAdd(fz.done_cnt + pair, 1) + 1 == end[pair] - begin[pair]) {
                __threadfence(); // the other triggers' results (written on other CUs / XCDs) before the replay reads them
                fuse_publish(fz, rp, pair, models, triggers, trig_cap);
            }
        }
    }
}

template <int KIND, bool SHIFT, int T>
__global__ __launch_bounds__(T, MDRP_LM_MINWAVES) void k_lo(RunParams rp, const PairState *__restrict__ st, const double *__restrict__ pts,
                                                   const double *__restrict__ dep, const Model *__restrict__ models,
                                                   Trigger *__restrict__ triggers, int trig_cap, const int32_t *__restrict__ plan,
                                                   int32_t *__restrict__ head /*zeroed*/, int32_t *__restrict__ xheads /*zeroed, or null: lo_take*/,
                                                   int list_stride, unsigned long long *__restrict__ lm_stats, FuseTail fz /*ready == null: off*/) {
    extern __shared__ uint16_t lm_dyn_list[];
    __shared__ LmShared sh;
    __shared__ int s_item;
    if (threadIdx.x == 0) { sh.list = lm_dyn_list; sh.stride = list_stride; sh.midx = 0; sh.stats = lm_stats; sh.ev[0] = 0; sh.ev[1] = 0; sh.logtab = nullptr; }
    __syncthreads();
    const int total = plan[3 * (size_t)rp.batch + 1];
    if (fz.ready && threadIdx.x == 0) { // pairs without a trigger in this launch are ready as they are (earlier LO launches have ended: stream order)
        const int32_t *begin = plan + rp.batch + 1, *end = begin + rp.batch;
        for (int p = blockIdx.x; p < rp.batch; p += gridDim.x)
            if (end[p] == begin[p]) fuse_publish(fz, rp, p, models, triggers, trig_cap);
        atomicAdd(fz.ctl + 1, 1);
    }
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_item = lo_take(head, xheads, total);
        __syncthreads();
        const int w = s_item;
        if (w >= total) break;
        lo_problem<KIND, SHIFT, T>(rp, st, pts, dep, models, triggers, trig_cap, plan, w, sh, fz);
    }
}

// ------------------------------------------------------------------------------------------------ final
// ransac<> tail (@0x22f1d0-0x22f295) + get_inliers (@0x4f7a10/@0x4f77f0) + the estimator's inlier-only refinement
// (@0x2247c3 / @0x223815) + focal un-normalisation.
struct ResultDev {
    Model model;
    uint64_t refinements, iterations, num_inliers;
    double inlier_ratio, model_score;
};

// FLOSS: the user's loss type of the inlier-only refinement as a compile-time constant (rp.final_loss; -1 = read o.loss at run time).
// Round 5: the two refinements of ransac<>'s tail are two call sites of lm_refine, each with its loss (and whether a record mask
// exists) known to the compiler — the LO with TRUNCATED exactly as k_lo compiles it (unit-weight rows, no products by the weight), the
// inlier-only refinement without the six-way loss switch in its sweeps.  Nothing but the model under refinement is live across an LM:
// the round-4 kernel carried `best`, `m`, `x` and the result record (100 VGPRs) through both refinements and spilled 201-449 VGPRs.
template <int KIND, bool SHIFT, int T, int FLOSS>
__device__ __forceinline__ void final_pair(const RunParams &rp, const PairState &ps, const double *__restrict__ pts, const double *__restrict__ dep,
                                           uint8_t *__restrict__ mask_all, ResultDev *__restrict__ results, int pair, LmShared &sh) {
    double *scratch = sh.scratch;
    uint8_t *mask = mask_all + (size_t)pair * rp.n_max;
    if (ps.n < 3) {
        for (int i = threadIdx.x; i < rp.n_max; i += T) mask[i] = 0;
        if (threadIdx.x == 0) {
            ResultDev res;
            res.model = ps.best;
            res.refinements = ps.refinements; res.iterations = ps.iterations; res.num_inliers = ps.num_inliers;
            res.inlier_ratio = ps.inlier_ratio; res.model_score = ps.model_score;
            results[pair] = res;
        }
        return;
    }
    const double *pp = pts + (size_t)pair * rp.n_max * PT_STRIDE;
    const double *dd = dep + (size_t)pair * rp.n_max * 2;
    // ransac<>'s last LO from the best model: 25 iterations, TRUNCATED, all records
    Model x = ps.best;
    {
        LmOpt o;
        o.max_it = 25; o.loss = 1; o.loss_scale = ps.lo_loss_scale;
        o.grad_tol = 1e-10; o.step_tol = 1e-8; o.lambda0 = 1e-3; o.lambda_min = 1e-10; o.lambda_max = 1e10;
        if (x.q[0] == x.q[0]) lm_refine<KIND, SHIFT, T, 1>(x, pp, dd, ps.n, nullptr, ps.scale_reproj, rp.weight_sampson, o, sh); // (NaN model: lo_problem)
    }
    uint64_t num_inliers = ps.num_inliers;
    {
        double sc;
        int cn;
        block_score<T>(KIND, x, pp, ps.n, ps.sq_thr, scratch, sc, cn, nullptr);
        if (sc < ps.model_score) num_inliers = (uint64_t)cn; // the refined model is adopted; score / ratio NOT updated (reference)
        else x = ps.best;                                    // (re-read: no second model is kept live across the LM)
        for (int i = ps.n + threadIdx.x; i < rp.n_max; i += T) mask[i] = 0;
        block_score<T>(KIND, x, pp, ps.n, ps.sq_thr, scratch, sc, cn, mask); // get_inliers of the winner
        __syncthreads();
    }
    // the estimator's inlier-only refinement with the user's BundleOptions: with more than 3 inliers in the calibrated (@0x224434) and shared-focal
    // (@0x2235e6) wrappers, with more than 7 in the varying-focal one (@0x223d16)
    if (num_inliers > (KIND == 2 ? 7u : 3u)) {
        LmOpt o;
        o.max_it = rp.final_max_it; o.loss = FLOSS >= 0 ? FLOSS : rp.final_loss; o.loss_scale = ps.final_loss_scale;
        o.grad_tol = rp.grad_tol; o.step_tol = rp.step_tol; o.lambda0 = rp.lambda0; o.lambda_min = rp.lambda_min; o.lambda_max = rp.lambda_max;
        lm_refine<KIND, SHIFT, T, FLOSS>(x, pp, dd, ps.n, mask, ps.scale_reproj, rp.weight_sampson, o, sh);
    }
    if (KIND != 0) { x.f1 *= ps.norm; x.f2 *= ps.norm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ResultDev res;
        res.model = x;
        res.refinements = ps.refinements + 1; res.iterations = ps.iterations; res.num_inliers = num_inliers;
        res.inlier_ratio = ps.inlier_ratio; res.model_score = ps.model_score;
        results[pair] = res;
        if (rp.inl_stat) { // what this pair would have liked as the run's first chunk: ~6 outlier-free samples expected in it (first_chunk_wish)
            atomicAdd(&rp.inl_stat[0], first_chunk_wish((double)num_inliers / (double)ps.n, 3));
            atomicAdd(&rp.inl_stat[1], 1);
        }
        lm_flush_stats(sh);
    }
}

template <int KIND, bool SHIFT, int T, int FLOSS>
__global__ __launch_bounds__(T, MDRP_LM_MINWAVES) void k_final(RunParams rp, PairState *__restrict__ st, const double *__restrict__ pts,
                                                      const double *__restrict__ dep, uint8_t *__restrict__ mask_all,
                                                      ResultDev *__restrict__ results, int list_stride, int mask_index /*1: 3 * list_stride entries of dynamic LDS*/,
                                                      unsigned long long *__restrict__ lm_stats,
                                                      const int32_t *__restrict__ ready /*or null: pair = blockIdx.x*/,
                                                      int32_t *__restrict__ fin_done /*fused: set per refined pair; unfused: pairs to skip, or null*/,
                                                      unsigned long long ticks, unsigned long long *__restrict__ timeouts /*fused: expired bounded waits*/) {
    extern __shared__ uint16_t lm_dyn_list[];
    __shared__ LmShared sh;
    __shared__ int s_pair;
    __shared__ __attribute__((aligned(16))) unsigned int s_ps[(sizeof(PairState) + 3) / 4];
    __shared__ __attribute__((aligned(16))) double s_logtab[(FLOSS == 3 || FLOSS == 4 || FLOSS < 0) ? 2 * MDRP_LOGTAB_N : 2];
    constexpr bool use_logtab = FLOSS == 3 || FLOSS == 4 || FLOSS < 0;
    if (use_logtab) lm_logtab_load(s_logtab);
    if (threadIdx.x == 0) {
        sh.list = lm_dyn_list; sh.stride = list_stride; sh.midx = mask_index; sh.stats = lm_stats; sh.ev[0] = 0; sh.ev[1] = 0;
        sh.logtab = use_logtab ? s_logtab : nullptr;
        int p = blockIdx.x;
        if (ready) { // fused tail: the blockIdx-th pair to become ready (bounded wait, see k_gate)
            const unsigned long long t0 = wall_clock64();
            // polled RELAXED, one acquire fence when the pair is there: an acquire LOAD is followed by an invalidate of the XCD's L2, and the few
            // hundred workgroups waiting here did that every microsecond while the last LO problems were still sweeping.  (Found with an experimental
            // queue-driven k_final whose idle workgroups polled the same way: the pairs still running took twice as long per iteration.)
            while ((p = __hip_atomic_load(ready + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0 && wall_clock64() - t0 < ticks)
                __builtin_amdgcn_s_sleep(16);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (p < 0 && timeouts) atomicAdd(timeouts, 1ull); // gave up: the pass behind the LO launch refines this pair
            __threadfence(); // the replayed pair state (written on another CU / XCD) before anyone of this workgroup reads it
        } else if (fin_done && fin_done[p]) p = -1; // the pass behind a fused tail: only what that left undone
        s_pair = p;
    }
    __syncthreads();
    if (s_pair < 0) return;
    // The pair state is read ONCE, with vector loads, into LDS.  Fused: it was written DURING this launch (by the LO workgroup that replayed the
    // pair); this kernel never writes `st`, so the compiler may fetch st[pair] through the scalar cache, which an acquire does not invalidate and
    // which can hold the cache line that the previous pair's last field shares with this pair's head from BEFORE the replay.  Unfused: the same
    // copy keeps ONE call site of final_pair (round 4 compiled the 17 k-instruction body twice) and the pair's fields re-readable from LDS.
    {
        const volatile unsigned int *src = reinterpret_cast<const volatile unsigned int *>(st + s_pair);
        for (int i = threadIdx.x; i < (int)(sizeof(PairState) / 4); i += T) s_ps[i] = src[i];
    }
    __syncthreads();
    final_pair<KIND, SHIFT, T, FLOSS>(rp, *reinterpret_cast<const PairState *>(s_ps), pts, dep, mask_all, results, s_pair, sh);
    if (ready && threadIdx.x == 0) fin_done[s_pair] = 1;
}

// ------------------------------------------------------------------------------------------------ unit-parity kernels
MDRP_GLOBAL void k_solver_unit(int solver, int count, const double *__restrict__ x1h, const double *__restrict__ x2h,
                              const double *__restrict__ d1, const double *__restrict__ d2, Model *__restrict__ out, int32_t *__restrict__ n_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Sample3 s;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s.x1[k][0] = x1h[9 * i + 3 * k]; s.x1[k][1] = x1h[9 * i + 3 * k + 1];
        s.x2[k][0] = x2h[9 * i + 3 * k]; s.x2[k][1] = x2h[9 * i + 3 * k + 1];
        s.d1[k] = d1[3 * i + k]; s.d2[k] = d2[3 * i + k];
    }
    Model m[4];
    const int n = run_solver(solver, s, m);
    n_out[i] = n;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < n) out[4 * i + k] = m[k];
}

// coordinate box of ONE pair's records into st[0].box (unit sweep path)
MDRP_GLOBAL __launch_bounds__(256) void k_box_unit(int n, const double *__restrict__ pts, PairState *__restrict__ st) {
    __shared__ double red[4][4];
    double bx[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bx[q] = fmax(bx[q], fabs(pts[(size_t)i * PT_STRIDE + q]));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = bx[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) st[0].box[q] = fmax(fmax(red[0][q], red[1][q]), fmax(red[2][q], red[3][q]));
    }
}

// MFMA A fragments of ONE pair's packed records (unit path of k_count)
MDRP_GLOBAL void k_frag_unit(int n, const double *__restrict__ pts, uint4 *__restrict__ rfrag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int g_end = ((n + 15) / 16) * 16;
    if (i < n) {
        const double *p = pts + (size_t)i * PT_STRIDE;
        store_record_fragment(rfrag, i, p[0], p[1], p[2], p[3]);
    } else if (i < g_end) clear_record_fragment(rfrag, i);
}

// pack raw normalised correspondences of ONE pair into pts records (for mdrp_score_models / mdrp_refine_models)
MDRP_GLOBAL void k_pack_unit(int n, const double *__restrict__ x1, const double *__restrict__ x2, const double *__restrict__ d1,
                            const double *__restrict__ d2, double *__restrict__ pts, double *__restrict__ dep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x1[2 * i], b = x1[2 * i + 1], c = x2[2 * i], d = x2[2 * i + 1];
    double *p = pts + (size_t)i * PT_STRIDE;
    p[0] = a; p[1] = b; p[2] = c; p[3] = d;
    p[4] = 1.0 / sqrt(a * a + b * b + 1.0);
    p[5] = 1.0 / sqrt(c * c + d * d + 1.0);
    if (dep) { dep[2 * i] = d1 ? d1[i] : 0.0; dep[2 * i + 1] = d2 ? d2[i] : 0.0; }
}

template <int KIND, bool SHIFT, int T>
__global__ __launch_bounds__(T) void k_refine_unit(int count, Model *__restrict__ models,
                                                            const double *__restrict__ pts, const double *__restrict__ dep, int n,
                                                            double scale_reproj, double ws, LmOpt o, double *__restrict__ final_cost, int list_stride) {
    extern __shared__ uint16_t lm_dyn_list[];
    __shared__ LmShared sh;
    __shared__ __attribute__((aligned(16))) double s_logtab[2 * MDRP_LOGTAB_N];
    lm_logtab_load(s_logtab);
    if (threadIdx.x == 0) { sh.list = lm_dyn_list; sh.stride = list_stride; sh.midx = 0; sh.stats = nullptr; sh.ev[0] = 0; sh.ev[1] = 0; sh.logtab = s_logtab; }
    __syncthreads();
    const int i = blockIdx.x;
    if (i >= count) return;
    Model m = models[i];
    lm_refine<KIND, SHIFT, T>(m, pts, dep, n, nullptr, scale_reproj, ws, o, sh);
    const double c = lm_cost<KIND, T>(m, pts, dep, n, nullptr, sqrt(scale_reproj), ws, o, sh, 0);
    if (threadIdx.x == 0) { models[i] = m; if (final_cost) final_cost[i] = c; }
}

} // namespace mdrp
