/* mdrp_oracle.h — CPU restatement of the RePoseD RANSAC hot path (PoseLib 2.0.5 monodepth estimators).
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported baseline — never as the product path.
 *
 * The algorithm lives in a third-party dependency that is NOT vendored under /root/reference:
 * PoseLib 2.0.5, kocurvik/PoseLib@pr-mdrp (pinned by /root/reference/README.md:52-55 and
 * demo/reposed_demo.ipynb cell 4).  Only its compiled binary ships (demo/poselib-2.0.5-cp312-*.whl).
 * Every function here restates the published/observed algorithm of one exported symbol of that
 * binary (ELF addresses as in SURVEY.md §2/§8a) and is PINNED against outputs of that very binary
 * run in the build container (oracle/refshim + tests/tools/gen_golden.py -> tests/golden/ fixtures).
 */
#ifndef MDRP_ORACLE_H
#define MDRP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* MonoDepthTwoViewGeometry (+ the two focals of MonoDepthImagePair): q = (w,x,y,z), x_cam2 = R x_cam1 + t,
 * R (d1+shift1) x1 + t = scale (d2+shift2) x2   (wheel METADATA:264-272). */
typedef struct {
    double q[4];
    double t[3];
    double scale, shift1, shift2;
    double f1, f2;
} orc_model;

typedef struct { /* RansacOptions, wheel METADATA:72-91 */
    uint64_t max_iterations, min_iterations;
    double dyn_num_trials_mult, success_prob, max_reproj_error, max_epipolar_error;
    uint64_t seed;
    int estimate_shift; /* monodepth_estimate_shift */
    double weight_sampson; /* monodepth_weight_sampson (float in the reference) */
    int score_initial_model; /* RansacOptions +0x49: score (and LO-refine) the model handed in before the first iteration.
                              * ransac_*_relpose reset its POSE to the identity first (black-box: any initial pose gives the same
                              * result), so what is scored is (identity, initial scale / shifts). */
} orc_ransac_opt;

typedef struct { /* BundleOptions, wheel METADATA:94-106 */
    uint64_t max_iterations;
    int loss_type; /* 0 TRIVIAL 1 TRUNCATED 2 HUBER 3 CAUCHY 4 TRUNCATED_CAUCHY 5 TRUNCATED_LE_ZACH */
    double loss_scale, gradient_tol, step_tol, initial_lambda, min_lambda, max_lambda;
} orc_bundle_opt;

typedef struct { uint64_t refinements, iterations, num_inliers; double inlier_ratio, model_score; } orc_ransac_stats;
typedef struct { uint64_t iterations; double initial_cost, cost, lambda; uint64_t invalid_steps; double step_norm, grad_norm; } orc_bundle_stats;

enum { ORC_CALIB = 0, ORC_SHARED = 1, ORC_VARYING = 2 };

/* a-3 sampler */
int32_t orc_random_int(uint64_t *state);
int orc_p3p_reference_nan(const double xb[9], const double X[9]); /* does the reference's P3P return NaN poses for this sample? (orc_solvers.c) */
void orc_draw_sample(uint64_t n, uint64_t *state, uint64_t out[3]);
/* (uint64_t)d as the reference's x86-64 (gcc) build computes it, spelled out so that the oracle does not depend on the compiler it is built with:
 * d >= 2^63: cvttsd2si(d - 2^63) with the top bit flipped (+inf and d >= 2^64 -> 0); otherwise cvttsd2si(d) (NaN and d < -2^63 -> 2^63, negative d
 * wraps).  Decides what the dynamic iteration bound is for success_prob >= 1 (tests/golden/edge_options_ref.npz). */
static inline uint64_t orc_f64_to_u64(double d) {
    const double t63 = 9223372036854775808.0;
    if (d >= t63) {
        const double e = d - t63;
        return (e < t63 ? (uint64_t)(int64_t)e : 0x8000000000000000ull) ^ 0x8000000000000000ull;
    }
    return d >= -t63 ? (uint64_t)(int64_t)d : 0x8000000000000000ull;
}

/* geometry helpers */
void orc_quat_to_rotmat(const double q[4], double R[9] /*row-major*/);
void orc_rotmat_to_quat(const double R[9], double q[4]);
void orc_essential(const orc_model *m, double E[9] /*row-major*/);
void orc_fundamental(const orc_model *m, double F[9] /*row-major: diag(1,1,f2) E diag(1,1,f1)*/);

/* a-7 / a-9 scoring */
int orc_check_cheirality(const orc_model *m, const double x1[3], const double x2[3], double min_depth);
double orc_msac_pose(const orc_model *m, const double *x1, const double *x2, int n, double sq_thr, uint64_t *cnt);
double orc_msac_F(const double F[9], const double *x1, const double *x2, int n, double sq_thr, uint64_t *cnt);
int orc_inliers_pose(const orc_model *m, const double *x1, const double *x2, int n, double sq_thr, uint8_t *mask);
int orc_inliers_F(const double F[9], const double *x1, const double *x2, int n, double sq_thr, uint8_t *mask);

/* a-4..a-6' minimal solvers; inputs 3x3 row-major point arrays; return number of models written (<=4) */
int orc_p3p(const double x[9] /*unit bearings*/, const double X[9], orc_model out[4]);
int orc_solver_calib_shift(const double x1h[9], const double x2h[9], const double d1[3], const double d2[3], orc_model out[4]);
int orc_solver_calib_p3p(const double x1h[9], const double x2h[9], const double d1[3], const double d2[3], orc_model out[4]);
int orc_solver_shared(const double x1h[9], const double x2h[9], const double d1[3], const double d2[3], orc_model out[4]);
int orc_solver_varying(const double x1h[9], const double x2h[9], const double d1[3], const double d2[3], orc_model out[4]);

/* a-8 refinement (hybrid Sampson + forward/backward reprojection LM).  weights may be NULL. */
orc_bundle_stats orc_refine(int kind, const double *x1, const double *x2, const double *d1, const double *d2, int n,
                            orc_model *m, double scale_reproj, double weight_sampson, const orc_bundle_opt *opt,
                            int estimate_shift, const double *weights);

/* a-2 LO-RANSAC on normalised inputs, a-1 full estimators */
orc_ransac_stats orc_ransac(int kind, const double *x1, const double *x2, const double *d1, const double *d2, int n,
                            const orc_ransac_opt *opt, orc_model *best, uint8_t *mask);
/* cam: {model_id, nparams, params...} only for ORC_CALIB (SIMPLE_PINHOLE=0 [f,cx,cy], PINHOLE=1 [fx,fy,cx,cy]) */
orc_ransac_stats orc_estimate(int kind, const double *x1, const double *x2, const double *d1, const double *d2, int n,
                              const double *cam1, const double *cam2, const orc_ransac_opt *ropt,
                              const orc_bundle_opt *bopt, orc_model *best, uint8_t *mask);

/* ---- non-monodepth baselines of the same binary, orc_classic.c.  Bearings row-major k x 3. */
void orc_fullpiv_qr_Q(double *A /*9 x m col-major, destroyed*/, int m, double *Q /*9 x 9 col-major*/);
int orc_real_roots(const double *coef /*ascending powers*/, int degree, double *roots);
int orc_relpose_5pt_E(const double *x1h, const double *x2h, double *E_out /*<=10 x 9 row-major*/);
int orc_relpose_5pt(const double *x1h, const double *x2h, orc_model *out /*<=40*/);
int orc_motion_from_essential(const double *E /*row-major*/, const double *x1h, const double *x2h, int npts, orc_model *out /*<=4*/);
int orc_relpose_7pt(const double *x1h, const double *x2h, double *F_out /*<=3 x 9 row-major*/);
/* 6-point relative pose with one shared unknown focal length (orc_sixpt.c): models with f1 = f2 = f, by ascending f */
int orc_relpose_6pt(const double *x1h, const double *x2h, orc_model *out /*<=60*/);
int orc_eigenvalues(double *a /*n x n row-major, destroyed*/, int n, double *wr, double *wi);
/* kind 3 = relative pose (blob: q, t), 4 = shared focal (q, t, f1 = f2 = f), 5 = fundamental (F row-major in the first 9 doubles) */
enum { ORC_RELPOSE = 3, ORC_SHARED_RELPOSE = 4, ORC_FUNDAMENTAL = 5 };
orc_bundle_stats orc_refine_classic(int kind, const double *x1, const double *x2, int n, orc_model *blob, const orc_bundle_opt *opt,
                                    const double *weights);
orc_ransac_stats orc_ransac_classic(int kind, const double *x1, const double *x2, int n, const orc_ransac_opt *opt, orc_model *best,
                                    uint8_t *mask);
orc_ransac_stats orc_estimate_classic(int kind, const double *x1, const double *x2, int n, const double *cam1, const double *cam2,
                                      const double *pp, const orc_ransac_opt *ropt, const orc_bundle_opt *bopt, orc_model *best,
                                      uint8_t *mask);

#ifdef __cplusplus
}
#endif
#endif
