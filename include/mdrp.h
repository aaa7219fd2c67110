/* mdrp.h — C ABI of the MI355X-native RePoseD RANSAC hot path (libmdrp_hip.so).
 *
 * Drop-in boundary.  The reference reaches this path through pybind11
 * (wheel poselib/_core.pyi:446-501) into PoseLib's C++:
 *     estimate_monodepth_relative_pose                @0x224170   (README.md:86, make_pair.py:111, make_video.py:284)
 *     estimate_shared_focal_monodepth_relative_pose   @0x223300   (README.md:90)
 *     estimate_varying_focal_monodepth_relative_pose  @0x223a40   (README.md:96)
 * one image pair per call.  The entry points below bind the same three estimators, batched: B image pairs per
 * call, each pair an independent unit (that is how the reference itself parallelises, eval.py:355-359).
 * mdrp_amd/_capi.py is the ctypes binding; INTEGRATION.md shows the stub a PoseLib maintainer would add.
 *
 * Conventions: plain pointers + sizes, caller owns every buffer, the library never frees caller memory,
 * int return codes (0 = ok), no exceptions cross the boundary.  One HIP stream per handle.  Threading (the reference
 * releases the GIL around its estimators, wrapper @0x8ad01, and is re-entrant): calls on DIFFERENT handles run
 * concurrently from different host threads; calls on the SAME handle are serialised by a lock inside the handle.
 * Every entry point runs on the handle's device and restores the caller's current HIP device before it returns.
 */
#ifndef MDRP_H
#define MDRP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MDRP_CALIB = 0, MDRP_SHARED_FOCAL = 1, MDRP_VARYING_FOCAL = 2,
       /* non-monodepth baselines of the same binary on the same kernels (d1 = d2 = NULL):
        * estimate_relative_pose (wheel _core.pyi:504-529; 5-point, cameras as for MDRP_CALIB; model: q, t) and
        * estimate_fundamental (_core.pyi:309-323; 7-point; model: F row-major in the first nine doubles of mdrp_model). */
       MDRP_RELPOSE_5PT = 3, MDRP_FUNDAMENTAL_7PT = 5,
       /* estimate_shared_focal_relative_pose (wheel _core.pyi: 6-point, one unknown focal length shared by both images;
        * /root/reference/eval_shared_f.py:161).  Pixels relative to nothing: the principal point travels in cam1[i].params[0..1]
        * (cam2 is not read and may be NULL); model: q, t, f1 = f2 = f in pixels */
       MDRP_SHARED_6PT = 4 };

enum { /* return codes */
    MDRP_OK = 0,
    MDRP_ERR_INVALID = 1,   /* bad argument */
    MDRP_ERR_HIP = 2,       /* a HIP runtime call failed; mdrp_last_error() has the text */
    MDRP_ERR_NO_DEVICE = 3, /* no usable gfx950 device */
    MDRP_ERR_UNSUPPORTED = 4 /* ABI 0.4: a RansacOptions switch of the reference that selects behaviour this library does not build
                              * (progressive_sampling = PROSAC; real_focal_check on the 6- / 7-point baselines).  Refused, never ignored:
                              * the reference would switch samplers / drop models and return different results. */
};

enum { /* where the caller's buffers live */
    MDRP_MEM_HOST = 0,
    MDRP_MEM_DEVICE = 1
};

/* MonoDepthTwoViewGeometry + the two focals of MonoDepthImagePair (wheel _core.pyi:134-204):
 * q = (w,x,y,z);  R (d1+shift1) K1^-1 x1 + t = scale (d2+shift2) K2^-1 x2 */
typedef struct {
    double q[4];
    double t[3];
    double scale, shift1, shift2;
    double f1, f2; /* 1.0 for the calibrated estimator */
} mdrp_model;

/* RansacOptions (wheel METADATA:72-91; defaults as the pybind wrapper @0x8ab59-0x8ac44) */
typedef struct {
    uint64_t max_iterations;   /* 100000 */
    uint64_t min_iterations;   /* 1000 */
    double dyn_num_trials_mult; /* 3.0 */
    double success_prob;        /* 0.9999 */
    double max_reproj_error;    /* 12.0 (pixels) */
    double max_epipolar_error;  /* 1.0 (pixels) */
    uint64_t seed;              /* 0 */
    int32_t monodepth_estimate_shift; /* calibrated estimator only; ignored elsewhere exactly like the reference */
    float monodepth_weight_sampson;   /* 1.0.  A float in the reference too (+0x4c); the wrappers hand max(ws, 0) on.  Away from 1 the reference's
                                       * refiners are not self-consistent and the library reproduces them as they are: the Sampson term enters
                                       * the LM COST as ws rho(r^2) but the normal equations as ws^2 w(.) J'J, with w evaluated at r^2 in the
                                       * calibrated refiner and at ws r^2 in the two focal ones (tests/golden/refine_ws.npz). */
    int32_t score_initial_model;      /* 0.  RansacOptions +0x49, set by the binding when an initial pose is passed with it.  What the
                                       * reference then scores first is NOT the caller's pose: ransac_*_relpose reset it to the identity
                                       * (black-box: any initial pose gives the same result).  The reset model has E = 0: no inliers,
                                       * score N eps^2; its LO changes nothing.  Reproduced as that state: records start at
                                       * (0, N eps^2) and `refinements` at 1 (tests/golden/initial.npz). */
    /* ---- ABI 0.4: the remaining RansacOptions fields of the reference (+0x38 progressive_sampling,
     * +0x40 max_prosac_iterations, +0x48 real_focal_check), so that a host can hand its options over unabridged.
     * PROSAC (RandomSampler::initialize_prosac @0x4f8a20) is not built — every caller in the reference passes
     * progressive_sampling = False (eval.py:99, make_video.py:192): a non-zero value is refused with MDRP_ERR_UNSUPPORTED
     * on every estimator.  (This field sits where ABI 0.3 had a zero `reserved_` word.) */
    int32_t progressive_sampling;    /* 0 */
    uint64_t max_prosac_iterations;  /* 100000; read only with progressive_sampling, i.e. never */
    int32_t real_focal_check;        /* 0.  Only the 6- / 7-point baselines look at it in the reference; refused there when set */
    int32_t reserved_;
} mdrp_ransac_opt;

/* BundleOptions (wheel METADATA:94-106) */
typedef struct {
    uint64_t max_iterations; /* 100 */
    int32_t loss_type;       /* 0 TRIVIAL 1 TRUNCATED 2 HUBER 3 CAUCHY 4 TRUNCATED_CAUCHY 5 TRUNCATED_LE_ZACH */
    double loss_scale;       /* 1.0.  Final refinement of the shared- / varying-focal estimators and of the 5- / 6- / 7-point baselines: divided by the
                              * normalisation scale.  The calibrated monodepth estimator IGNORES it, as the reference does
                              * (estimate_monodepth_relative_pose @0x224704): its final loss scale is half the normalised epipolar threshold,
                              * (1/f1 + 1/f2) * max_epipolar_error / 4 — the same number as loss_scale = 1 at max_epipolar_error = 2, the
                              * reference's own setting (tests/golden/options_ref.npz). */
    double gradient_tol;     /* 1e-10 */
    double step_tol;         /* 1e-8 */
    double initial_lambda;   /* 1e-3 */
    double min_lambda;       /* 1e-10 */
    double max_lambda;       /* 1e10 */
} mdrp_bundle_opt;

/* Pinhole intrinsics of one image (Camera, _core.pyi:76-132; only the models the reference callers use):
 * model_id 0 SIMPLE_PINHOLE params {f,cx,cy,-} ; 1 PINHOLE params {fx,fy,cx,cy} */
typedef struct {
    int32_t model_id;
    int32_t pad_;
    double params[4];
} mdrp_camera;

/* One record per image pair: the estimator's return value + RansacStats.  As in the reference, the model of MDRP_CALIB without the shift flag can be a
 * NaN pose (num_inliers 0, model_score N * eps^2) when no sample of the run gave a real pose: the reference's p3p() emits NaN poses for ~3 % of the
 * samples, and such a model is the record until a real one is scored. */
typedef struct {
    mdrp_model model;
    uint64_t refinements, iterations, num_inliers;
    double inlier_ratio, model_score;
} mdrp_result;

typedef struct mdrp_handle mdrp_handle;

/* Library/handle management.  device = HIP device ordinal.  stream = a hipStream_t created by the caller or NULL to
 * let the handle create its own (non-blocking) stream.
 * mdrp_create / mdrp_create_on_stream are MACROS over the exported mdrp_create_ / mdrp_create_on_stream_ (the zlib deflateInit pattern): they hand the
 * ABI version and the size of mdrp_ransac_opt of the header the HOST was compiled against to the library, which refuses a mismatch
 * (MDRP_ERR_INVALID, mdrp_last_error() names both versions) instead of reading option fields past the end of a smaller struct.  A host built
 * against ABI 0.4 or older, which imported the plain symbols, no longer loads against this library: it must be recompiled. */
int mdrp_create_(int device, void *stream, mdrp_handle **out, int abi_version, int ransac_opt_bytes);
/* Same, but `stream` is used exactly as given: NULL means the device's legacy default (null) stream — which is what
 * torch.cuda.current_stream().cuda_stream is (0) unless the caller switched streams.  Work of the handle is then
 * ordered with everything else queued on that stream (inputs produced by earlier kernels, consumers of the mask). */
int mdrp_create_on_stream_(int device, void *stream, mdrp_handle **out, int abi_version, int ransac_opt_bytes);
#define mdrp_create(device, stream, out) mdrp_create_((device), (stream), (out), MDRP_ABI_VERSION, (int)sizeof(mdrp_ransac_opt))
#define mdrp_create_on_stream(device, stream, out) mdrp_create_on_stream_((device), (stream), (out), MDRP_ABI_VERSION, (int)sizeof(mdrp_ransac_opt))
void mdrp_destroy(mdrp_handle *h);
const char *mdrp_last_error(void);
/* "mdrp-hip <ver> (gfx950) MDRP_SRC_HASH=<16 hex digits>": the hash covers mdrp_capi.hip, mdrp_kernels.h, mdrp_math.h,
 * mdrp_classic.h, mdrp_classic_math.h and this header as they were when the library was built (mdrp_amd/build.py source_hash()) */
const char *mdrp_version(void);
/* ABI 0.5: (major << 16) | minor of the structs and entry points in this header = 0x00000005.  mdrp_ransac_opt grew from 72 to 88 bytes in
 * 0.4; 0.5 makes the version check involuntary: handles are created through mdrp_create_ / mdrp_create_on_stream_, which take the host's
 * MDRP_ABI_VERSION and sizeof(mdrp_ransac_opt) (the macros above pass them) and refuse another ABI. */
#define MDRP_ABI_VERSION 0x00000005
int mdrp_abi_version(void);
/* HIP_VERSION (major * 10^7 + minor * 10^5 + patch) of the toolchain the library was compiled with.  The library carries no HIP
 * runtime of its own (it binds to the host process's libamdhip64 when it is loaded): a host compares this with
 * hipRuntimeGetVersion() of the runtime it links; mdrp_amd/_capi.py refuses a different major. */
int mdrp_hip_build_version(void);
/* block the calling thread until all work queued on the handle's stream is done */
int mdrp_synchronize(mdrp_handle *h);

/* Batched estimators.  x1,x2: [B][n_max][2] pixel coordinates; d1,d2: [B][n_max] depths; n_per_pair: [B] valid
 * correspondences per pair (host memory always; NULL = all n_max).  cam1/cam2: [B] cameras (host memory; calibrated
 * estimator only — the focal estimators take principal-point-centred pixels, README.md:88-96).  out: [B] results
 * (host memory).  inlier_mask: [B][n_max] bytes or NULL (same memory space as the inputs).
 * Pairs with fewer than 3 correspondences return zeroed stats with model_score = DBL_MAX and the identity model,
 * like ransac<> @0x22f087.  The call is synchronous with respect to `out`. */
int mdrp_estimate_batch(mdrp_handle *h, int kind, int mem_space, const double *x1, const double *x2, const double *d1,
                        const double *d2, int batch, int n_max, const int32_t *n_per_pair, const mdrp_camera *cam1,
                        const mdrp_camera *cam2, const mdrp_ransac_opt *ropt, const mdrp_bundle_opt *bopt,
                        mdrp_result *out, uint8_t *inlier_mask);

/* Same work on buffers that already live on the device, and nothing is copied back: results stay in the handle's device
 * buffers until mdrp_fetch_results / mdrp_copy_results_device.  Used by bench.py so that the timed region holds device work
 * only (inputs resident in HBM).  "async" is relative to the RESULTS, not to the host: the call queues every kernel of a
 * super-chunk without waiting, but the LO-RANSAC stop rule needs one 48-byte progress record per super-chunk on the host
 * (one short hipStreamSynchronize each; a single one when max_iterations == min_iterations). */
int mdrp_estimate_batch_async(mdrp_handle *h, int kind, const double *x1_dev, const double *x2_dev, const double *d1_dev,
                              const double *d2_dev, int batch, int n_max, const int32_t *n_per_pair_host,
                              const mdrp_camera *cam1_host, const mdrp_camera *cam2_host, const mdrp_ransac_opt *ropt,
                              const mdrp_bundle_opt *bopt, uint8_t *inlier_mask_dev);
int mdrp_fetch_results(mdrp_handle *h, mdrp_result *out_host, int batch);
/* The same records into DEVICE memory (e.g. a torch tensor that goes straight into the RCCL all-gather of the poses,
 * SURVEY.md 8e) — nothing crosses PCIe.  Returns after the handle's stream has drained. */
int mdrp_copy_results_device(mdrp_handle *h, void *dst_dev, int batch);

/* ---- unit-parity entry points (the reference exposes the same pieces: _core.pyi:614-619, 871-876, 914-919) ---- */
/* Minimal solvers on `count` independent 3-point problems (host memory).  x1h,x2h: [count][3][3] homogeneous points
 * (z = 1), d1,d2: [count][3].  out: [count][4] models, n_out: [count] number of valid models.
 * solver: 0 calibrated P3P path (shift off), 1 calibrated with shifts, 2 shared focal, 3 varying focal. */
int mdrp_solver_batch(mdrp_handle *h, int solver, const double *x1h, const double *x2h, const double *d1,
                      const double *d2, int count, mdrp_model *out, int32_t *n_out);

/* The baselines' minimal solvers (relpose_5pt @0x14ae80, relpose_7pt @0x4ff2e0) on `count` independent problems, host memory.
 * x1h, x2h: [count][K][3] unit bearings, K = 5 / 7.  out: [count][M] models, M = 10 / 3 (solutions in the reference's order);
 * n_out: [count]. */
int mdrp_classic_solver_batch(mdrp_handle *h, int kind, const double *x1h, const double *x2h, int count, mdrp_model *out,
                              int32_t *n_out);

/* Sampson/MSAC sweep only (compute_sampson_msac_score @0x4f61d0 / @0x4f65d0): `num_models` models against the n
 * normalised correspondences of ONE pair.  kind selects pose scoring with cheirality (MDRP_CALIB, MDRP_RELPOSE_5PT), F built
 * from pose and focals (focal estimators), or the raw F of MDRP_FUNDAMENTAL_7PT models.
 * All pointers in `mem_space`.  scores: [num_models], counts: [num_models]. */
int mdrp_score_models(mdrp_handle *h, int kind, int mem_space, const mdrp_model *models, int num_models,
                      const double *x1, const double *x2, int n, double sq_threshold, double *scores, int32_t *counts);

/* Candidate counts of the MFMA pre-pass alone (k_count): for every model an UPPER bound on its inlier count against the n
 * normalised correspondences of one pair — the number of correspondences the conservative bf16-split filter cannot prove
 * to be outliers.  Host memory.  candidates: [num_models]. */
int mdrp_count_candidates(mdrp_handle *h, int kind, const mdrp_model *models, int num_models, const double *x1,
                          const double *x2, int n, double sq_threshold, int32_t *candidates);

/* The fp32 stage between the two (k_bound) alone: for every model a LOWER bound of its MSAC score and an UPPER bound of its
 * inlier count against the n normalised correspondences of one pair, from packed-fp32 arithmetic with explicit error terms
 * (DESIGN.md 4).  A hypothesis is retired without the exact fp64 sweep when these two bounds prove that it cannot break a
 * record — the tests assert score_lb <= exact score and count_ub >= exact count on adversarial inputs.  A model whose
 * coefficients leave the fp32 range proves nothing: it reports (0, n).  Host memory.  score_lb, count_ub: [num_models]. */
int mdrp_bound_models(mdrp_handle *h, int kind, const mdrp_model *models, int num_models, const double *x1, const double *x2,
                      int n, double sq_threshold, double *score_lb, int32_t *count_ub);

/* Hybrid LM refinement of `count` models, each over the correspondences of ONE pair (refine_monodepth_*relpose
 * @0x261030/@0x2592e0/@0x260fa0).  Host memory.  models in/out.  For MDRP_RELPOSE_5PT / MDRP_FUNDAMENTAL_7PT: the Sampson-only
 * refine_relpose @0x258f50 / refine_fundamental @0x2590d0 (d1, d2, scale_reproj, weight_sampson, estimate_shift ignored). */
int mdrp_refine_models(mdrp_handle *h, int kind, mdrp_model *models, int count, const double *x1, const double *x2,
                       const double *d1, const double *d2, int n, double scale_reproj, double weight_sampson,
                       const mdrp_bundle_opt *opt, int estimate_shift, double *final_cost /*[count] or NULL*/);

/* Timing of the last mdrp_estimate_batch* call on this handle, measured with HIP events on the handle's stream:
 * total milliseconds in the scoring-sweep kernel, number of its launches, and (model x correspondence) evaluations. */
int mdrp_last_sweep_stats(mdrp_handle *h, double *sweep_ms, int64_t *launches, int64_t *evaluations);

/* The same with the two scoring kernels apart (HIP events on the handle's stream around every launch of each kernel):
 * k_count — candidate counts of all hypotheses on the matrix cores (v_mfma_f32_16x16x32_bf16) — and k_score — the exact
 * fp64 sweep of the hypotheses k_count could not retire. */
typedef struct {
    double count_ms;           /* total time in k_count */
    int64_t count_launches;
    double sweep_ms;           /* total time in k_score */
    int64_t sweep_launches;
    int64_t evals_algorithmic; /* (model x correspondence) evaluations the CPU loop does: sum over pairs of models * n */
    int64_t evals_mfma;        /* evaluations executed by k_count (16 x 16 tiles, padding included) */
    int64_t evals_fp64;        /* evaluations handed to k_score (survivors * n) */
    int64_t evals_bound;       /* evaluations executed by k_bound in fp32 (k_count's survivors * n) */
    /* LM refinements (refine_monodepth_*relpose @0x261030 / @0x2592e0 / @0x260fa0): HIP events around every launch of the LO
     * kernel (on the stream it runs on) and of the final-refinement kernel, and the correspondences their sweeps evaluated */
    double lo_ms;              /* total time in k_lo (or the LM engine's LO phases) */
    int64_t lo_launches;
    double final_ms;           /* total time in k_final (or the LM engine's final phase) */
    int64_t final_launches;
    double bound_ms;           /* total time in k_bound */
    int64_t bound_launches;
    double solve_ms;           /* total time in the minimal-solver kernel */
    int64_t solve_launches;
    int64_t lm_cost_evals;     /* LO kernel: correspondences evaluated by its cost sweeps (residuals only) */
    int64_t lm_accum_evals;    /* LO kernel: correspondences evaluated by its normal-equation sweeps (residuals + Jacobians + J'J) */
    int64_t final_cost_evals;  /* the same two counters of the final-refinement kernel */
    int64_t final_accum_evals;
    /* ---- appended in ABI 0.3 (mdrp_last_stats_sized only) ----
     * Fused tail (the last LO launch and the final refinements overlap on two streams): bounded waits that expired in the
     * last call.  Non-zero means kernels of the handle's streams did not run side by side (serialising profiler / debugger,
     * AMD_SERIALIZE_KERNEL, a busy shared GPU): results are unaffected, the call was slower, final_ms includes the waits, and the
     * handle runs unfused for its next 64 calls (MDRP_FUSE_RETRY_CALLS), then tries again. */
    int64_t fuse_gate_timeouts;
    int64_t fuse_wait_timeouts;
    /* ---- appended in ABI 0.5 (mdrp_last_stats_sized only) ----
     * Iterations of the first chunk of the last call — the part of a run that is scored exactly in full because nothing has set a bar yet.  The
     * monodepth estimators size it from the inlier ratios of the results of the handle's PREVIOUS call with the same estimator (a pair whose first
     * chunk holds no outlier-free sample has no bar for the rest of its run: 256 iterations where half of the correspondences are inliers, 1024
     * where one in seven is, 128 where all are), 256 without one; MDRP_CHUNKS overrides.  Results do not depend on it. */
    int64_t first_chunk;
} mdrp_stats;
/* writes min(out_size, sizeof(mdrp_stats)) bytes: pass sizeof(mdrp_stats) of the header the caller was compiled against */
int mdrp_last_stats_sized(mdrp_handle *h, mdrp_stats *out, size_t out_size);
/* the ABI 0.2 entry point: writes the ABI 0.2 struct (everything before fuse_gate_timeouts), never more.
 * With the fused tail, lo_ms and final_ms are overlapping intervals on two streams and final_ms includes the final refinements' wait
 * for their pairs: read them as one phase (lo_ms + final_ms is an upper bound of it), not as two kernel durations. */
int mdrp_last_stats(mdrp_handle *h, mdrp_stats *out);

#ifdef __cplusplus
}
#endif
#endif
