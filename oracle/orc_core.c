/* orc_core.c — sampler, pose algebra, Sampson/MSAC scoring.  TEST INFRASTRUCTURE (see mdrp_oracle.h).
 * Restates (reference binary demo/poselib-2.0.5-cp312-*.whl!poselib/_core*.so):
 *   random_int @0x4f87a0, draw_sample @0x4f87f0                    (a-3)
 *   essential_from_motion @0x1dcb60, check_cheirality @0x1dce00     (a-7)
 *   compute_sampson_msac_score(CameraPose) @0x4f61d0, (Matrix3d) @0x4f65d0   (a-7)
 *   get_inliers(CameraPose) @0x4f7a10, get_inliers(Matrix3d) @0x4f77f0       (a-9)
 */
#include "mdrp_oracle.h"
#include <math.h>
#include <string.h>

/* splitmix64 step, truncated to int32 (a-3) */
int32_t orc_random_int(uint64_t *state) {
    *state += 0x9e3779b97f4a7c15ULL;
    uint64_t z = *state;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return (int32_t)(z ^ (z >> 31));
}

/* three distinct indices: sign-extend the int32 then unsigned modulo n; redraw on duplicate (a-3) */
void orc_draw_sample(uint64_t n, uint64_t *state, uint64_t out[3]) {
    for (int i = 0; i < 3; ++i) {
        int dup;
        do {
            out[i] = (uint64_t)(int64_t)orc_random_int(state) % n;
            dup = 0;
            for (int j = 0; j < i; ++j) dup |= (out[j] == out[i]);
        } while (dup);
    }
}

/* Eigen::Quaterniond(w,x,y,z).toRotationMatrix() operation order */
void orc_quat_to_rotmat(const double q[4], double R[9]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* Eigen::Quaterniond(Matrix3d) followed by normalisation */
void orc_rotmat_to_quat(const double R[9], double q[4]) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t;
        q[2] = (R[2] - R[6]) * t;
        q[3] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
        q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * t;
        q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
    double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; ++a) q[a] /= nrm;
}

/* E = [t]x R */
void orc_essential(const orc_model *m, double E[9]) {
    double R[9];
    orc_quat_to_rotmat(m->q, R);
    const double *t = m->t;
    for (int c = 0; c < 3; ++c) {
        E[0 + c] = -t[2] * R[3 + c] + t[1] * R[6 + c];
        E[3 + c] = t[2] * R[0 + c] - t[0] * R[6 + c];
        E[6 + c] = -t[1] * R[0 + c] + t[0] * R[3 + c];
    }
}

/* F = K2^-1' E K1^-1 up to scale with K = diag(f,f,1):  diag(1,1,f2) E diag(1,1,f1)  (score_model @0x4fac60) */
void orc_fundamental(const orc_model *m, double F[9]) {
    orc_essential(m, F);
    F[2] *= m->f1; F[5] *= m->f1; F[8] *= m->f1;
    F[6] *= m->f2; F[7] *= m->f2; F[8] *= m->f2;
}

static void quat_rotate(const double q[4], const double p[3], double out[3]) {
    const double q1 = q[0], q2 = q[1], q3 = q[2], q4 = q[3];
    const double p1 = p[0], p2 = p[1], p3 = p[2];
    const double px1 = -p1 * q2 - p2 * q3 - p3 * q4;
    const double px2 = p1 * q1 - p2 * q4 + p3 * q3;
    const double px3 = p2 * q1 + p1 * q4 - p3 * q2;
    const double px4 = p2 * q2 - p1 * q3 + p3 * q1;
    out[0] = px2 * q1 - px1 * q2 - px3 * q4 + px4 * q3;
    out[1] = px3 * q1 - px1 * q3 + px2 * q4 - px4 * q2;
    out[2] = px3 * q2 - px2 * q3 - px1 * q4 + px4 * q1;
}

/* unit bearings x1,x2; depths from [1 a; a 1][l1;l2] = [b1;b2], factor 1/(1-a^2) dropped (a-7) */
int orc_check_cheirality(const orc_model *m, const double x1[3], const double x2[3], double min_depth) {
    double u[3];
    quat_rotate(m->q, x1, u);
    const double a = -(u[0] * x2[0] + u[1] * x2[1] + u[2] * x2[2]);
    const double b1 = -(u[0] * m->t[0] + u[1] * m->t[1] + u[2] * m->t[2]);
    const double b2 = x2[0] * m->t[0] + x2[1] * m->t[1] + x2[2] * m->t[2];
    const double l1 = b1 - a * b2;
    const double l2 = -a * b1 + b2;
    min_depth = min_depth * (1 - a * a);
    return l1 > min_depth && l2 > min_depth;
}

static inline double sampson_sq(const double E[9], double a, double b, double c, double d) {
    const double Ex1_0 = E[0] * a + E[1] * b + E[2];
    const double Ex1_1 = E[3] * a + E[4] * b + E[5];
    const double Ex1_2 = E[6] * a + E[7] * b + E[8];
    const double Ex2_0 = E[0] * c + E[3] * d + E[6];
    const double Ex2_1 = E[1] * c + E[4] * d + E[7];
    const double C = c * Ex1_0 + d * Ex1_1 + Ex1_2;
    const double Cx = Ex1_0 * Ex1_0 + Ex1_1 * Ex1_1;
    const double Cy = Ex2_0 * Ex2_0 + Ex2_1 * Ex2_1;
    return C * C / (Cx + Cy);
}

static int cheir_px(const orc_model *m, const double *x1, const double *x2) {
    double n1 = sqrt(x1[0] * x1[0] + x1[1] * x1[1] + 1.0), n2 = sqrt(x2[0] * x2[0] + x2[1] * x2[1] + 1.0);
    double b1[3] = {x1[0] / n1, x1[1] / n1, 1.0 / n1}, b2[3] = {x2[0] / n2, x2[1] / n2, 1.0 / n2};
    return orc_check_cheirality(m, b1, b2, 0.01);
}

double orc_msac_pose(const orc_model *m, const double *x1, const double *x2, int n, double sq_thr, uint64_t *cnt) {
    double E[9], score = 0.0;
    orc_essential(m, E);
    *cnt = 0;
    for (int k = 0; k < n; ++k) {
        const double r2 = sampson_sq(E, x1[2 * k], x1[2 * k + 1], x2[2 * k], x2[2 * k + 1]);
        if (r2 < sq_thr && cheir_px(m, x1 + 2 * k, x2 + 2 * k)) {
            score += r2;
            (*cnt)++;
        } else {
            score += sq_thr;
        }
    }
    return score;
}

double orc_msac_F(const double F[9], const double *x1, const double *x2, int n, double sq_thr, uint64_t *cnt) {
    double score = 0.0;
    *cnt = 0;
    for (int k = 0; k < n; ++k) {
        const double r2 = sampson_sq(F, x1[2 * k], x1[2 * k + 1], x2[2 * k], x2[2 * k + 1]);
        if (r2 < sq_thr) {
            score += r2;
            (*cnt)++;
        } else {
            score += sq_thr;
        }
    }
    return score;
}

int orc_inliers_pose(const orc_model *m, const double *x1, const double *x2, int n, double sq_thr, uint8_t *mask) {
    double E[9];
    int c = 0;
    orc_essential(m, E);
    for (int k = 0; k < n; ++k) {
        const double r2 = sampson_sq(E, x1[2 * k], x1[2 * k + 1], x2[2 * k], x2[2 * k + 1]);
        mask[k] = (uint8_t)(r2 < sq_thr && cheir_px(m, x1 + 2 * k, x2 + 2 * k));
        c += mask[k];
    }
    return c;
}

int orc_inliers_F(const double F[9], const double *x1, const double *x2, int n, double sq_thr, uint8_t *mask) {
    int c = 0;
    for (int k = 0; k < n; ++k) {
        const double r2 = sampson_sq(F, x1[2 * k], x1[2 * k + 1], x2[2 * k], x2[2 * k + 1]);
        mask[k] = (uint8_t)(r2 < sq_thr);
        c += mask[k];
    }
    return c;
}
