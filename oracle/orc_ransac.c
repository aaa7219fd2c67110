#include <stdio.h>
/* orc_ransac.c — LO-RANSAC driver and the three estimate_* wrappers.  TEST INFRASTRUCTURE (see mdrp_oracle.h).
 *
 * Restates (reference binary):
 *   ransac<Estimator,Model> @0x22f030/0x230e10/0x2321f0 + score_models<> @0x22ebc0/0x2306b0/0x231a90
 *   ransac_monodepth_relpose @0x228c20, ransac_shared_focal_monodepth_relpose @0x2298e0, ..._varying_ @0x22a3a0
 *   estimate_monodepth_relative_pose @0x224170, estimate_shared_focal_... @0x223300, estimate_varying_focal_... @0x223a40
 * This is the SEQUENTIAL form exactly as the reference runs it (LO interleaved with sampling).
 */
#include "mdrp_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int kind, n;
    const double *x1, *x2, *d1, *d2;
    const orc_ransac_opt *opt;
    double sq_thr, scale_reproj;
    uint64_t rng;
} estimator;

static int generate_models(estimator *e, orc_model out[4]) {
    uint64_t s[3];
    orc_draw_sample((uint64_t)e->n, &e->rng, s);
    double x1h[9], x2h[9], d1[3], d2[3];
    for (int k = 0; k < 3; ++k) {
        x1h[3 * k] = e->x1[2 * s[k]]; x1h[3 * k + 1] = e->x1[2 * s[k] + 1]; x1h[3 * k + 2] = 1.0;
        x2h[3 * k] = e->x2[2 * s[k]]; x2h[3 * k + 1] = e->x2[2 * s[k] + 1]; x2h[3 * k + 2] = 1.0;
        d1[k] = e->d1[s[k]]; d2[k] = e->d2[s[k]];
    }
    switch (e->kind) {
    case ORC_CALIB:
        if (e->opt->estimate_shift) return orc_solver_calib_shift(x1h, x2h, d1, d2, out);
        {
            /* the reference's P3P returns NaN poses where its degenerate conic is a point conic (orc_p3p_reference_nan): four of them, each scoring
             * N * thr with no inlier — one is enough to reproduce what they do to the records */
            double X[9], xb[9];
            for (int i = 0; i < 3; ++i) {
                const double nrm = sqrt(x2h[3 * i] * x2h[3 * i] + x2h[3 * i + 1] * x2h[3 * i + 1] + x2h[3 * i + 2] * x2h[3 * i + 2]);
                for (int k = 0; k < 3; ++k) { X[3 * i + k] = d1[i] * x1h[3 * i + k]; xb[3 * i + k] = x2h[3 * i + k] / nrm; }
            }
            if (orc_p3p_reference_nan(xb, X)) {
                memset(&out[0], 0, sizeof out[0]);
                out[0].q[0] = out[0].q[1] = out[0].q[2] = out[0].q[3] = out[0].t[0] = out[0].t[1] = out[0].t[2] = out[0].scale = NAN;
                out[0].f1 = out[0].f2 = 1.0;
                return 1;
            }
        }
        return orc_solver_calib_p3p(x1h, x2h, d1, d2, out);
    case ORC_SHARED: return orc_solver_shared(x1h, x2h, d1, d2, out);
    default: return orc_solver_varying(x1h, x2h, d1, d2, out);
    }
}

static double score_model(const estimator *e, const orc_model *m, uint64_t *cnt) {
    if (e->kind == ORC_CALIB) return orc_msac_pose(m, e->x1, e->x2, e->n, e->sq_thr, cnt);
    double F[9];
    orc_fundamental(m, F);
    return orc_msac_F(F, e->x1, e->x2, e->n, e->sq_thr, cnt);
}

/* LO step (refine_model @0x4fa550/0x4fad60/0x4fb0a0): 25 iterations, TRUNCATED loss, all N correspondences.
 * BundleOptions as built in the binary: gradient_tol 1e-10, step_tol 1e-8, lambda 1e-3 in [1e-10,1e10];
 * loss_scale = max_epipolar_error for the calibrated (@0x4fa5da) and shared-focal (@0x4fadf2) estimators, but
 * the varying-focal estimator leaves loss_scale at its default 1.0 (@0x4fb0e1 loads {1.0,1e-10}) — on
 * scale-normalised points that makes its TRUNCATED loss effectively untruncated.  Reproduced as is. */
static void refine_model(const estimator *e, orc_model *m) {
    orc_bundle_opt b;
    b.max_iterations = 25; b.loss_type = 1;
    b.loss_scale = e->kind == ORC_VARYING ? 1.0 : e->opt->max_epipolar_error;
    b.gradient_tol = 1e-10; b.step_tol = 1e-8; b.initial_lambda = 1e-3; b.min_lambda = 1e-10; b.max_lambda = 1e10;
    const orc_bundle_stats st = orc_refine(e->kind, e->x1, e->x2, e->d1, e->d2, e->n, m, e->scale_reproj, e->opt->weight_sampson, &b,
                                           e->kind == ORC_CALIB && e->opt->estimate_shift, NULL);
    if (getenv("ORC_TRACE_LM")) fprintf(stderr, "[orc] LO  LM iterations %llu invalid %llu\n", (unsigned long long)st.iterations, (unsigned long long)st.invalid_steps);
}

orc_ransac_stats orc_ransac(int kind, const double *x1, const double *x2, const double *d1, const double *d2, int n,
                            const orc_ransac_opt *opt, orc_model *best, uint8_t *mask) {
    orc_ransac_stats stats;
    memset(&stats, 0, sizeof stats);
    stats.model_score = DBL_MAX;
    if (mask) memset(mask, 0, (size_t)(n > 0 ? n : 0));
    if (n < 3) return stats;
    estimator e;
    e.kind = kind; e.n = n; e.x1 = x1; e.x2 = x2; e.d1 = d1; e.d2 = d2; e.opt = opt;
    e.sq_thr = opt->max_epipolar_error * opt->max_epipolar_error;
    e.scale_reproj = opt->max_reproj_error > 0.0
                         ? (opt->max_epipolar_error * opt->max_epipolar_error) / (opt->max_reproj_error * opt->max_reproj_error)
                         : 0.0;
    e.rng = opt->seed;

    uint64_t best_min_cnt = 0;
    double best_min_score = DBL_MAX;
    uint64_t dynamic_max_iter = opt->max_iterations;
    const double log_prob_missing = log(1.0 - opt->success_prob);
    orc_model models[4];
    /* ransac_*_relpose @0x228c20 / 0x2298e0 / 0x22a3a0 reset the pose of the caller's model (black-box probe: results do not
     * depend on it; with nothing found the caller gets identity + its own scale back) */
    best->q[0] = 1.0; best->q[1] = best->q[2] = best->q[3] = 0.0;
    best->t[0] = best->t[1] = best->t[2] = 0.0;
    if (kind != ORC_CALIB) { best->f1 = 1.0; best->f2 = 1.0; }
    int pending_initial = opt->score_initial_model != 0; /* ransac<> branch @0x22f2c8: score_models on {*best} first */
    for (;;) {
        int nm;
        if (!pending_initial && stats.iterations >= opt->max_iterations) break; /* max_iterations = 0: no sample is drawn (ransac<> loop head) */
        if (pending_initial) { models[0] = *best; nm = 1; }
        else nm = generate_models(&e, models);
        int best_ind = -1;
        for (int i = 0; i < nm; ++i) {
            uint64_t cnt;
            const double score = score_model(&e, &models[i], &cnt);
            const int more = cnt > best_min_cnt, better = score < best_min_score;
            if (more || better) {
                if (more) best_min_cnt = cnt;
                if (better) best_min_score = score;
                best_ind = i;
                if (score < stats.model_score) {
                    stats.model_score = score;
                    *best = models[i];
                    stats.num_inliers = cnt;
                }
            }
        }
        if (best_ind >= 0) {
            orc_model refined = models[best_ind];
            refine_model(&e, &refined);
            stats.refinements++;
            uint64_t cnt;
            const double score = score_model(&e, &refined, &cnt);
            if (score < stats.model_score) {
                stats.model_score = score;
                stats.num_inliers = cnt;
                *best = refined;
            }
            stats.inlier_ratio = (double)stats.num_inliers / (double)n;
            if (stats.inlier_ratio >= 0.9999) dynamic_max_iter = opt->min_iterations;
            else if (stats.inlier_ratio <= 0.0001) dynamic_max_iter = opt->max_iterations;
            else {
                const double prob_outlier = 1.0 - pow(stats.inlier_ratio, 3.0);
                dynamic_max_iter = orc_f64_to_u64(ceil(log_prob_missing / log(prob_outlier) * opt->dyn_num_trials_mult));
            }
        }
        if (pending_initial) { pending_initial = 0; continue; } /* not an iteration */
        ++stats.iterations;
        if (stats.iterations >= opt->max_iterations) break;
        if (stats.iterations <= opt->min_iterations) continue;
        if (stats.iterations > dynamic_max_iter) break;
    }
    /* final refinement of the winner: adopts the model and its inlier count, not the score / ratio */
    {
        orc_model refined = *best;
        refine_model(&e, &refined);
        stats.refinements++;
        uint64_t cnt;
        const double score = score_model(&e, &refined, &cnt);
        if (score < stats.model_score) {
            *best = refined;
            stats.num_inliers = cnt;
        }
    }
    if (mask) {
        if (kind == ORC_CALIB) orc_inliers_pose(best, x1, x2, n, e.sq_thr, mask);
        else { double F[9]; orc_fundamental(best, F); orc_inliers_F(F, x1, x2, n, e.sq_thr, mask); }
    }
    return stats;
}

static double cam_focal(const double *cam) { /* Camera::focal(): SIMPLE_PINHOLE f ; PINHOLE mean(fx,fy) */
    return ((int)cam[0] == 1) ? 0.5 * (cam[2] + cam[3]) : cam[2];
}
static void cam_unproject(const double *cam, const double *x, double *o) {
    if ((int)cam[0] == 1) { o[0] = (x[0] - cam[4]) / cam[2]; o[1] = (x[1] - cam[5]) / cam[3]; }
    else { o[0] = (x[0] - cam[3]) / cam[2]; o[1] = (x[1] - cam[4]) / cam[2]; }
}

orc_ransac_stats orc_estimate(int kind, const double *x1, const double *x2, const double *d1, const double *d2, int n,
                              const double *cam1, const double *cam2, const orc_ransac_opt *ropt,
                              const orc_bundle_opt *bopt, orc_model *best, uint8_t *mask) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    double *a1 = (double *)malloc(sizeof(double) * 2 * nn), *a2 = (double *)malloc(sizeof(double) * 2 * nn);
    orc_ransac_opt ro = *ropt;
    orc_bundle_opt bo = *bopt;
    double norm = 1.0;
    /* RansacOptions::monodepth_weight_sampson is a float (+0x4c); the wrappers hand max(ws, 0) on as a double (@0x2246e6) */
    ro.weight_sampson = (double)(float)ropt->weight_sampson;
    if (!(ro.weight_sampson > 0.0)) ro.weight_sampson = 0.0;
    if (kind == ORC_CALIB) {
        for (int k = 0; k < n; ++k) { cam_unproject(cam1, x1 + 2 * k, a1 + 2 * k); cam_unproject(cam2, x2 + 2 * k, a2 + 2 * k); }
        const double k = 0.5 * (1.0 / cam_focal(cam1) + 1.0 / cam_focal(cam2));
        ro.max_epipolar_error *= k; ro.max_reproj_error *= k;
        /* the calibrated wrapper does NOT scale the caller's BundleOptions::loss_scale: it overwrites it with half the
         * normalised epipolar threshold, (1/f1 + 1/f2) * (max_epipolar_error * 0.25) (@0x224704 .. 0x2247a8); the two focal
         * wrappers do divide the caller's value by the normalisation scale.  With the reference's own settings
         * (max_epipolar_error 2, loss_scale 1: eval.py) the two readings coincide. */
        bo.loss_scale = (1.0 / cam_focal(cam2) + 1.0 / cam_focal(cam1)) * (ropt->max_epipolar_error * 0.25);
    } else {
        /* normalize_points(..., normalize_scale=1, normalize_centroid=0, shared_scale=1) @0x4f6ae0 */
        double acc = 0.0;
        for (int k = 0; k < n; ++k)
            acc += sqrt(x1[2 * k] * x1[2 * k] + x1[2 * k + 1] * x1[2 * k + 1]) + sqrt(x2[2 * k] * x2[2 * k] + x2[2 * k + 1] * x2[2 * k + 1]);
        norm = acc / (sqrt(2.0) * (double)(n > 0 ? n : 1));
        for (int k = 0; k < 2 * n; ++k) { a1[k] = x1[k] / norm; a2[k] = x2[k] / norm; }
        ro.max_epipolar_error /= norm; ro.max_reproj_error /= norm; bo.loss_scale /= norm;
    }
    uint8_t *m8 = mask ? mask : (uint8_t *)malloc(nn);
    orc_ransac_stats stats = orc_ransac(kind, a1, a2, d1, d2, n, &ro, best, m8);
    /* inlier-only refinement: more than 3 inliers in the calibrated (cmp $3 @0x224434) and shared-focal (@0x2235e6) wrappers, more than 7
     * in the varying-focal one (cmp $7 @0x223d16) */
    if (stats.num_inliers > (kind == ORC_VARYING ? 7u : 3u)) {
        int ni = 0;
        double *i1 = (double *)malloc(sizeof(double) * 2 * nn), *i2 = (double *)malloc(sizeof(double) * 2 * nn);
        double *e1 = (double *)malloc(sizeof(double) * nn), *e2 = (double *)malloc(sizeof(double) * nn);
        for (int k = 0; k < n; ++k)
            if (m8[k]) {
                i1[2 * ni] = a1[2 * k]; i1[2 * ni + 1] = a1[2 * k + 1];
                i2[2 * ni] = a2[2 * k]; i2[2 * ni + 1] = a2[2 * k + 1];
                e1[ni] = d1[k]; e2[ni] = d2[k];
                ++ni;
            }
        const double sr = ro.max_reproj_error > 0.0
                              ? (ro.max_epipolar_error * ro.max_epipolar_error) / (ro.max_reproj_error * ro.max_reproj_error)
                              : 0.0;
        const orc_bundle_stats fst = orc_refine(kind, i1, i2, e1, e2, ni, best, sr, ro.weight_sampson, &bo, kind == ORC_CALIB && ro.estimate_shift, NULL);
        if (getenv("ORC_TRACE_LM")) fprintf(stderr, "[orc] final LM iterations %llu invalid %llu (n = %d)\n", (unsigned long long)fst.iterations, (unsigned long long)fst.invalid_steps, ni);
        free(i1); free(i2); free(e1); free(e2);
    }
    if (kind != ORC_CALIB) { best->f1 *= norm; best->f2 *= norm; }
    if (!mask) free(m8);
    free(a1); free(a2);
    return stats;
}
