/* orc_refine.c — hybrid Sampson + forward/backward reprojection Levenberg-Marquardt (a-8).
 * TEST INFRASTRUCTURE (see mdrp_oracle.h).
 *
 * Restates refine_monodepth_relpose @0x261030, refine_monodepth_shared_focal_relpose @0x2592e0,
 * refine_monodepth_varying_focal_relpose @0x260fa0 and the lm_impl<> loop they instantiate (reference binary
 * only).  Cost per correspondence, pinned against the binary's `initial_cost` to 1e-14 relative:
 *     ws * rho(Sampson^2) + rho(sr * |pi(R (d1+u) b1 + t) - x2|^2) + rho(sr * |pi(R'(s (d2+v) b2 - t)) - x1|^2)
 * with b = (x/f, y/f, 1), pi(X) = f X.xy / X.z, rho the robust loss with threshold loss_scale.
 * LM loop (upstream PoseLib lm_impl convention): IRLS weights rho'(r^2), damping lambda added to the diagonal,
 * Cholesky solve, accept if the cost decreases (lambda/10) else reject (lambda*10), stop on
 * |J'r| < gradient_tol or |step| < step_tol.
 */
#include "mdrp_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* probing switches (kept so tools/probe_lm.py can re-run the discrimination; defaults = what the binary does) */
int orc_lm_rot_pre = 0;   /* 0: R <- R exp([w]x)   1: R <- exp([w]x) R */
int orc_lm_t_mode = 0;    /* 0: t <- t + dt        1: t <- t + R dt */
int orc_lm_s_mode = 0;    /* 0: s <- s + ds        1: s <- s exp(ds) */

#define NPAR 11 /* rot(3) t(3) s u v f1 f2 */

/* the truncated losses are std::min(r2, t2) = (t2 < r2) ? t2 : r2 in the reference: a NaN residual (E = 0: the reset identity model, a NaN
 * model) makes the COST NaN, no LM step is ever accepted and the model comes back unchanged (black-box: refine_* on the identity model return it
 * with cost NaN under every loss), while the IRLS weight of such a row is 0 */
static double loss_value(int type, double thr, double r2) {
    const double t2 = thr * thr;
    switch (type) {
    case 0: return r2;
    case 1: return t2 < r2 ? t2 : r2;
    case 2: { const double r = sqrt(r2); return r <= thr ? r2 : thr * (2.0 * r - thr); }
    case 3: return t2 * log1p(r2 / t2);
    case 4: return t2 * log1p((t2 < r2 ? t2 : r2) / t2);
    case 5: return t2 < r2 ? t2 : r2;
    }
    return r2;
}
static double lz_mu = 0.5; /* TRUNCATED_LE_ZACH penalty strength: starts at 0.5, x1.5 after every LM iteration */
static double loss_weight(int type, double thr, double r2) {
    const double t2 = thr * thr;
    switch (type) {
    case 0: return 1.0;
    case 1: return r2 < t2 ? 1.0 : 0.0;
    case 2: { const double r = sqrt(r2); return r <= thr ? 1.0 : thr / r; }
    case 3: { const double w = 1.0 / (1.0 + r2 / t2); return w > DBL_MIN ? w : DBL_MIN; }
    case 4: { if (!(r2 < t2)) return 0.0; const double w = 1.0 / (1.0 + r2 / t2); return w > DBL_MIN ? w : DBL_MIN; }
    case 5: { /* Le & Zach, 3DV 2021 (upstream PoseLib TruncatedLossLeZach::weight) */
        const double r2h = r2 / t2;
        if (r2h < 1.0) return 0.5;
        const double mu = lz_mu, zstar = 1.0, r2m1 = r2h - 1.0;
        const double rho = (2.0 * r2m1 + sqrt(4.0 * r2m1 * r2m1 * mu * mu + 2.0 * mu * r2m1)) / mu;
        const double a = (r2h + mu * rho * zstar - 0.5 * rho) / (1.0 + mu * rho);
        const double zbar = a < 0.0 ? 0.0 : (a > 1.0 ? 1.0 : a);
        return (zstar - zbar) / rho;
    }
    }
    return 1.0;
}

static void mat3_vec(const double *R, const double *x, double *y) {
    for (int i = 0; i < 3; ++i) y[i] = R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2];
}
static void mat3t_vec(const double *R, const double *x, double *y) {
    for (int i = 0; i < 3; ++i) y[i] = R[i] * x[0] + R[3 + i] * x[1] + R[6 + i] * x[2];
}

typedef struct {
    double R[9], t[3], s, u, v, f1, f2;
    double E[9], F[9];
} lm_state;

static void state_from_model(const orc_model *m, int kind, lm_state *st) {
    orc_quat_to_rotmat(m->q, st->R);
    memcpy(st->t, m->t, sizeof st->t);
    st->s = m->scale; st->u = m->shift1; st->v = m->shift2;
    st->f1 = kind == ORC_CALIB ? 1.0 : m->f1;
    st->f2 = kind == ORC_CALIB ? 1.0 : m->f2;
    const double *t = st->t, *R = st->R;
    for (int c = 0; c < 3; ++c) {
        st->E[0 + c] = -t[2] * R[3 + c] + t[1] * R[6 + c];
        st->E[3 + c] = t[2] * R[0 + c] - t[0] * R[6 + c];
        st->E[6 + c] = -t[1] * R[0 + c] + t[0] * R[3 + c];
    }
    memcpy(st->F, st->E, sizeof st->F);
    st->F[2] *= st->f1; st->F[5] *= st->f1; st->F[8] *= st->f1;
    st->F[6] *= st->f2; st->F[7] *= st->f2; st->F[8] *= st->f2;
}

/* residuals r[0..4] = {sampson, fwd.x, fwd.y, bwd.x, bwd.y} (reprojection ones already times sqrt(sr));
 * r[5], r[6] = depth of the forward / backward transferred point (terms with negative depth are skipped by the
 * callers, as upstream PoseLib's reprojection accumulators do: "if (Z(2) < 0) continue");  J[5][NPAR] if J != NULL */
static void point_residuals(const lm_state *st, double sqrt_sr, const double *x1, const double *x2, double d1, double d2,
                            double r[7], double (*J)[NPAR]) {
    const double *R = st->R, *t = st->t, *F = st->F;
    const double h1[3] = {x1[0], x1[1], 1.0}, h2[3] = {x2[0], x2[1], 1.0};
    double Fh1[3], Fth2[3];
    mat3_vec(F, h1, Fh1);
    mat3t_vec(F, h2, Fth2);
    const double C = h2[0] * Fh1[0] + h2[1] * Fh1[1] + Fh1[2];
    const double den = Fh1[0] * Fh1[0] + Fh1[1] * Fh1[1] + Fth2[0] * Fth2[0] + Fth2[1] * Fth2[1];
    const double isd = 1.0 / sqrt(den);
    r[0] = C * isd;
    /* forward */
    const double b1[3] = {x1[0] / st->f1, x1[1] / st->f1, 1.0}, b2[3] = {x2[0] / st->f2, x2[1] / st->f2, 1.0};
    const double dd1 = d1 + st->u, dd2 = d2 + st->v;
    const double X1[3] = {dd1 * b1[0], dd1 * b1[1], dd1};
    double Z[3];
    mat3_vec(R, X1, Z);
    Z[0] += t[0]; Z[1] += t[1]; Z[2] += t[2];
    const double iz = 1.0 / Z[2];
    r[1] = sqrt_sr * (st->f2 * Z[0] * iz - x2[0]);
    r[2] = sqrt_sr * (st->f2 * Z[1] * iz - x2[1]);
    /* backward */
    const double X2[3] = {st->s * dd2 * b2[0], st->s * dd2 * b2[1], st->s * dd2};
    const double Y[3] = {X2[0] - t[0], X2[1] - t[1], X2[2] - t[2]};
    double W[3];
    mat3t_vec(R, Y, W);
    const double iw = 1.0 / W[2];
    r[3] = sqrt_sr * (st->f1 * W[0] * iw - x1[0]);
    r[4] = sqrt_sr * (st->f1 * W[1] * iw - x1[1]);
    r[5] = Z[2];
    r[6] = W[2];
    if (!J) return;
    memset(J, 0, sizeof(double) * 5 * NPAR);

    /* ---- Sampson: G = d r / d F ---- */
    double G[9];
    const double k = C * isd * isd * isd;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double g = h2[i] * h1[j] * isd;
            if (i < 2) g -= k * Fh1[i] * h1[j];
            if (j < 2) g -= k * Fth2[j] * h2[i];
            G[3 * i + j] = g;
        }
    /* dr/dE_ij = G_ij * (f2 if i==2) * (f1 if j==2) */
    double GE[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) GE[3 * i + j] = G[3 * i + j] * (i == 2 ? st->f2 : 1.0) * (j == 2 ? st->f1 : 1.0);
    const double *E = st->E;
    for (int a = 0; a < 3; ++a) {
        /* rotation: post  dE = E [e_a]x ;  pre  dE = [t]x [e_a]x R */
        double dE[9];
        if (!orc_lm_rot_pre) {
            /* (E [e_a]x)_ij = sum_k E_ik [e_a]x_kj ; [e_a]x_kj = -eps_{a k j} */
            const int b = (a + 1) % 3, c = (a + 2) % 3;
            for (int i = 0; i < 3; ++i) { dE[3 * i + a] = 0.0; dE[3 * i + b] = -E[3 * i + c]; dE[3 * i + c] = E[3 * i + b]; }
            /* check: [e_a]x has (c,b)=+1? [e_x]x = [[0,0,0],[0,0,-1],[0,1,0]] -> (k=2,j=1)=+1,(k=1,j=2)=-1.
               column j=b=1: E_i2*(+1)  -> dE[:,b] = +E[:,c];  column j=c=2: E_i1*(-1) -> dE[:,c] = -E[:,b] */
            for (int i = 0; i < 3; ++i) { dE[3 * i + b] = E[3 * i + c]; dE[3 * i + c] = -E[3 * i + b]; }
        } else {
            /* [e_a]x R : row b of result = -R row c ; row c = +R row b ; row a = 0.   ([e_x]x R)_1j = -R_2j, _2j = R_1j */
            double M[9];
            const int b = (a + 1) % 3, c = (a + 2) % 3;
            for (int j = 0; j < 3; ++j) { M[3 * a + j] = 0.0; M[3 * b + j] = -R[3 * c + j]; M[3 * c + j] = R[3 * b + j]; }
            for (int j = 0; j < 3; ++j) {
                dE[0 + j] = -t[2] * M[3 + j] + t[1] * M[6 + j];
                dE[3 + j] = t[2] * M[0 + j] - t[0] * M[6 + j];
                dE[6 + j] = -t[1] * M[0 + j] + t[0] * M[3 + j];
            }
        }
        double acc = 0;
        for (int i = 0; i < 9; ++i) acc += GE[i] * dE[i];
        J[0][a] = acc;
    }
    for (int a = 0; a < 3; ++a) {
        /* translation direction: e_a (mode 0) or R e_a = column a of R (mode 1);  dE = [dir]x R */
        double dir[3] = {0, 0, 0};
        if (orc_lm_t_mode == 0) dir[a] = 1.0; else { dir[0] = R[a]; dir[1] = R[3 + a]; dir[2] = R[6 + a]; }
        double acc = 0;
        for (int j = 0; j < 3; ++j) {
            acc += GE[0 + j] * (-dir[2] * R[3 + j] + dir[1] * R[6 + j]);
            acc += GE[3 + j] * (dir[2] * R[0 + j] - dir[0] * R[6 + j]);
            acc += GE[6 + j] * (-dir[1] * R[0 + j] + dir[0] * R[3 + j]);
        }
        J[0][3 + a] = acc;
    }
    { /* focals: F_ij = E_ij f2^[i==2] f1^[j==2] */
        double a1 = 0, a2 = 0;
        for (int i = 0; i < 3; ++i) a1 += G[3 * i + 2] * E[3 * i + 2] * (i == 2 ? st->f2 : 1.0);
        for (int j = 0; j < 3; ++j) a2 += G[6 + j] * E[6 + j] * (j == 2 ? st->f1 : 1.0);
        J[0][9] = a1; J[0][10] = a2;
    }

    /* ---- forward reprojection ---- */
    {
        const double Pz[6] = {st->f2 * iz, 0.0, -st->f2 * Z[0] * iz * iz, 0.0, st->f2 * iz, -st->f2 * Z[1] * iz * iz};
        double dZ[3][NPAR];
        memset(dZ, 0, sizeof dZ);
        for (int a = 0; a < 3; ++a) {
            double v3[3];
            if (!orc_lm_rot_pre) { /* d(R X1) = R (e_a x X1) */
                double ex[3] = {0, 0, 0}, cr[3];
                ex[a] = 1.0;
                cr[0] = ex[1] * X1[2] - ex[2] * X1[1]; cr[1] = ex[2] * X1[0] - ex[0] * X1[2]; cr[2] = ex[0] * X1[1] - ex[1] * X1[0];
                mat3_vec(R, cr, v3);
            } else { /* e_a x (R X1) */
                const double RX[3] = {Z[0] - t[0], Z[1] - t[1], Z[2] - t[2]};
                double ex[3] = {0, 0, 0};
                ex[a] = 1.0;
                v3[0] = ex[1] * RX[2] - ex[2] * RX[1]; v3[1] = ex[2] * RX[0] - ex[0] * RX[2]; v3[2] = ex[0] * RX[1] - ex[1] * RX[0];
            }
            for (int i = 0; i < 3; ++i) dZ[i][a] = v3[i];
            for (int i = 0; i < 3; ++i) dZ[i][3 + a] = orc_lm_t_mode == 0 ? (i == a ? 1.0 : 0.0) : R[3 * i + a];
        }
        double Rb1[3];
        mat3_vec(R, b1, Rb1);
        for (int i = 0; i < 3; ++i) dZ[i][7] = Rb1[i];
        const double dX1f[3] = {-dd1 * x1[0] / (st->f1 * st->f1), -dd1 * x1[1] / (st->f1 * st->f1), 0.0};
        double v3[3];
        mat3_vec(R, dX1f, v3);
        for (int i = 0; i < 3; ++i) dZ[i][9] = v3[i];
        for (int p = 0; p < NPAR; ++p) {
            J[1][p] = sqrt_sr * (Pz[0] * dZ[0][p] + Pz[2] * dZ[2][p]);
            J[2][p] = sqrt_sr * (Pz[4] * dZ[1][p] + Pz[5] * dZ[2][p]);
        }
        J[1][10] += sqrt_sr * Z[0] * iz;
        J[2][10] += sqrt_sr * Z[1] * iz;
    }
    /* ---- backward reprojection ---- */
    {
        const double Pw[6] = {st->f1 * iw, 0.0, -st->f1 * W[0] * iw * iw, 0.0, st->f1 * iw, -st->f1 * W[1] * iw * iw};
        double dW[3][NPAR];
        memset(dW, 0, sizeof dW);
        for (int a = 0; a < 3; ++a) {
            double v3[3];
            if (!orc_lm_rot_pre) { /* R' <- (I - [w]x) R' : dW = -e_a x W = W x e_a */
                double ex[3] = {0, 0, 0};
                ex[a] = 1.0;
                v3[0] = W[1] * ex[2] - W[2] * ex[1]; v3[1] = W[2] * ex[0] - W[0] * ex[2]; v3[2] = W[0] * ex[1] - W[1] * ex[0];
            } else { /* R' (I - [w]x) Y : dW = R' (Y x e_a) */
                double ex[3] = {0, 0, 0}, cr[3];
                ex[a] = 1.0;
                cr[0] = Y[1] * ex[2] - Y[2] * ex[1]; cr[1] = Y[2] * ex[0] - Y[0] * ex[2]; cr[2] = Y[0] * ex[1] - Y[1] * ex[0];
                mat3t_vec(R, cr, v3);
            }
            for (int i = 0; i < 3; ++i) dW[i][a] = v3[i];
            /* translation: dW = -R' dir */
            for (int i = 0; i < 3; ++i) dW[i][3 + a] = orc_lm_t_mode == 0 ? -R[3 * a + i] : (i == a ? -1.0 : 0.0);
        }
        double v3[3];
        const double ds_dir[3] = {dd2 * b2[0], dd2 * b2[1], dd2};
        mat3t_vec(R, ds_dir, v3);
        for (int i = 0; i < 3; ++i) dW[i][6] = (orc_lm_s_mode == 0 ? 1.0 : st->s) * v3[i];
        const double dv_dir[3] = {st->s * b2[0], st->s * b2[1], st->s};
        mat3t_vec(R, dv_dir, v3);
        for (int i = 0; i < 3; ++i) dW[i][8] = v3[i];
        const double df_dir[3] = {-st->s * dd2 * x2[0] / (st->f2 * st->f2), -st->s * dd2 * x2[1] / (st->f2 * st->f2), 0.0};
        mat3t_vec(R, df_dir, v3);
        for (int i = 0; i < 3; ++i) dW[i][10] = v3[i];
        for (int p = 0; p < NPAR; ++p) {
            J[3][p] = sqrt_sr * (Pw[0] * dW[0][p] + Pw[2] * dW[2][p]);
            J[4][p] = sqrt_sr * (Pw[4] * dW[1][p] + Pw[5] * dW[2][p]);
        }
        J[3][9] += sqrt_sr * W[0] * iw;
        J[4][9] += sqrt_sr * W[1] * iw;
    }
}

typedef struct {
    int kind, n, np, estimate_shift, loss_type;
    const double *x1, *x2, *d1, *d2, *weights;
    double sqrt_sr, ws, thr;
    int idx[NPAR]; /* active parameter -> column(s) of the full Jacobian (shared focal folds 9+10) */
} lm_problem;

static double lm_cost(const lm_problem *pb, const orc_model *m) {
    lm_state st;
    state_from_model(m, pb->kind, &st);
    double cost = 0;
    for (int k = 0; k < pb->n; ++k) {
        double r[7];
        point_residuals(&st, pb->sqrt_sr, pb->x1 + 2 * k, pb->x2 + 2 * k, pb->d1[k], pb->d2[k], r, NULL);
        const double w = pb->weights ? pb->weights[k] : 1.0;
        cost += w * pb->ws * loss_value(pb->loss_type, pb->thr, r[0] * r[0]);
        /* a reprojection term takes part only with a POSITIVE depth of the transferred point: a NaN depth (NaN in d1 / d2) drops the term like a negative
         * one does (black-box: refine_* with NaN depths have the finite cost of the same data with those depths negated), it does not poison the cost */
        if (r[5] > 0) cost += w * loss_value(pb->loss_type, pb->thr, r[1] * r[1] + r[2] * r[2]);
        if (r[6] > 0) cost += w * loss_value(pb->loss_type, pb->thr, r[3] * r[3] + r[4] * r[4]);
    }
    return cost;
}

static void lm_accumulate(const lm_problem *pb, const orc_model *m, double *JtJ, double *Jtr) {
    lm_state st;
    state_from_model(m, pb->kind, &st);
    const int np = pb->np;
    memset(JtJ, 0, sizeof(double) * np * np);
    memset(Jtr, 0, sizeof(double) * np);
    for (int k = 0; k < pb->n; ++k) {
        double r[7], Jf[5][NPAR], J[5][NPAR];
        point_residuals(&st, pb->sqrt_sr, pb->x1 + 2 * k, pb->x2 + 2 * k, pb->d1[k], pb->d2[k], r, Jf);
        for (int row = 0; row < 5; ++row)
            for (int p = 0; p < np; ++p) {
                J[row][p] = Jf[row][pb->idx[p]];
                if (pb->kind == ORC_SHARED && pb->idx[p] == 9) J[row][p] += Jf[row][10];
            }
        const double pw = pb->weights ? pb->weights[k] : 1.0;
        /* weight_sampson enters the normal equations SQUARED while lm_cost() above carries it to the first power, and the two focal
         * refiners evaluate the loss weight at ws * r^2 where the calibrated one evaluates it at r^2.  Neither is what a derivation
         * from the cost gives; both are what the reference binary computes (fitted against refine_monodepth_relpose /
         * refine_monodepth_{shared,varying}_focal_relpose for ws in {0.3, 0.5, 0.7, 1.3, 2, 3}, all six losses, with and without
         * per-point weights: tests/test_refshim_parity.py::test_weight_sampson_in_the_refiners).  At ws = 1 all forms coincide. */
        const double ws2 = pb->ws * pb->ws, rs2 = r[0] * r[0];
        const double wS = pw * ws2 * loss_weight(pb->loss_type, pb->thr, pb->kind == ORC_CALIB ? rs2 : pb->ws * rs2);
        const double wF = !(r[5] > 0) ? 0.0 : pw * loss_weight(pb->loss_type, pb->thr, r[1] * r[1] + r[2] * r[2]);
        const double wB = !(r[6] > 0) ? 0.0 : pw * loss_weight(pb->loss_type, pb->thr, r[3] * r[3] + r[4] * r[4]);
        const double wr[5] = {wS, wF, wF, wB, wB};
        for (int row = 0; row < 5; ++row) {
            if (wr[row] == 0.0) continue;
            for (int a = 0; a < np; ++a) {
                Jtr[a] += wr[row] * J[row][a] * r[row];
                for (int b = 0; b <= a; ++b) JtJ[a * np + b] += wr[row] * J[row][a] * J[row][b];
            }
        }
    }
}

/* solve (lower-stored SPD) A x = b by Cholesky; returns 0 on breakdown */
static int chol_solve(const double *A, const double *b, double *x, int n) {
    double L[NPAR * NPAR];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            if (i == j) { L[i * n + i] = sqrt(s); }
            else L[i * n + j] = s / L[j * n + j];
        }
    double y[NPAR];
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k];
        y[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
    return 1;
}

static void quat_mul(const double a[4], const double b[4], double o[4]) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
static void quat_exp(const double w[3], double q[4]) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    if (th > 1e-6) {
        const double re = cos(0.5 * th), im = sin(0.5 * th) / th;
        q[0] = re; q[1] = im * w[0]; q[2] = im * w[1]; q[3] = im * w[2];
    } else { /* series, as PoseLib quat_exp */
        const double re = 1.0 - th2 / 8.0, im = 0.5 - th2 / 48.0;
        double nq = sqrt(re * re + im * im * th2);
        q[0] = re / nq; q[1] = im * w[0] / nq; q[2] = im * w[1] / nq; q[3] = im * w[2] / nq;
    }
}

static void lm_step(const lm_problem *pb, const orc_model *m, const double *delta, orc_model *o) {
    *o = *m;
    double full[NPAR];
    memset(full, 0, sizeof full);
    for (int p = 0; p < pb->np; ++p) full[pb->idx[p]] = delta[p];
    if (pb->kind == ORC_SHARED) full[10] = full[9];
    double dq[4], R[9];
    quat_exp(full, dq);
    if (!orc_lm_rot_pre) quat_mul(m->q, dq, o->q); else quat_mul(dq, m->q, o->q);
    if (orc_lm_t_mode == 0) { for (int i = 0; i < 3; ++i) o->t[i] = m->t[i] + full[3 + i]; }
    else {
        orc_quat_to_rotmat(m->q, R);
        for (int i = 0; i < 3; ++i) o->t[i] = m->t[i] + R[3 * i] * full[3] + R[3 * i + 1] * full[4] + R[3 * i + 2] * full[5];
    }
    o->scale = orc_lm_s_mode == 0 ? m->scale + full[6] : m->scale * exp(full[6]);
    if (pb->kind == ORC_CALIB && pb->estimate_shift) {
        o->shift1 = m->shift1 + full[7];
        o->shift2 = m->shift2 + full[8];
    } else { /* black-box: the reference's step() rebuilds the geometry with zero shifts when they are not estimated */
        o->shift1 = 0.0;
        o->shift2 = 0.0;
    }
    if (pb->kind != ORC_CALIB) { o->f1 = m->f1 + full[9]; o->f2 = m->f2 + full[10]; }
}

orc_bundle_stats orc_refine(int kind, const double *x1, const double *x2, const double *d1, const double *d2, int n,
                            orc_model *m, double scale_reproj, double weight_sampson, const orc_bundle_opt *opt,
                            int estimate_shift, const double *weights) {
    lm_problem pb;
    memset(&pb, 0, sizeof pb);
    pb.kind = kind; pb.n = n; pb.x1 = x1; pb.x2 = x2; pb.d1 = d1; pb.d2 = d2; pb.weights = weights;
    pb.sqrt_sr = sqrt(scale_reproj); pb.ws = weight_sampson; pb.thr = opt->loss_scale; pb.loss_type = opt->loss_type;
    pb.estimate_shift = estimate_shift;
    int np = 0;
    for (int p = 0; p < 7; ++p) pb.idx[np++] = p;
    if (kind == ORC_CALIB && estimate_shift) { pb.idx[np++] = 7; pb.idx[np++] = 8; }
    if (kind == ORC_SHARED) pb.idx[np++] = 9;
    if (kind == ORC_VARYING) { pb.idx[np++] = 9; pb.idx[np++] = 10; }
    pb.np = np;

    orc_bundle_stats stats;
    memset(&stats, 0, sizeof stats);
    stats.cost = lm_cost(&pb, m);
    stats.initial_cost = stats.cost;
    stats.grad_norm = -1; stats.step_norm = -1; stats.invalid_steps = 0;
    stats.lambda = opt->initial_lambda;
    double JtJ[NPAR * NPAR], Jtr[NPAR], sol[NPAR];
    int recompute = 1;
    lz_mu = 0.5;
    for (stats.iterations = 0; stats.iterations < opt->max_iterations; ++stats.iterations) {
        if (recompute) {
            lm_accumulate(&pb, m, JtJ, Jtr);
            double g = 0;
            for (int p = 0; p < np; ++p) g += Jtr[p] * Jtr[p];
            stats.grad_norm = sqrt(g);
            if (stats.grad_norm < opt->gradient_tol) break;
        }
        for (int p = 0; p < np; ++p) JtJ[p * np + p] += stats.lambda;
        chol_solve(JtJ, Jtr, sol, np);
        double sn = 0;
        for (int p = 0; p < np; ++p) { sol[p] = -sol[p]; sn += sol[p] * sol[p]; }
        stats.step_norm = sqrt(sn);
        if (stats.step_norm < opt->step_tol) break;
        orc_model cand;
        lm_step(&pb, m, sol, &cand);
        const double cost_new = lm_cost(&pb, &cand);
        if (cost_new < stats.cost) {
            *m = cand;
            stats.lambda = fmax(opt->min_lambda, stats.lambda / 10.0);
            stats.cost = cost_new;
            recompute = 1;
        } else {
            stats.invalid_steps++;
            for (int p = 0; p < np; ++p) JtJ[p * np + p] -= stats.lambda;
            stats.lambda = fmin(opt->max_lambda, stats.lambda * 10.0);
            recompute = 0;
        }
        lz_mu *= 1.5; /* the reference's per-iteration callback */
    }
    return stats;
}

/* finite-difference check helper for tests: full residual vector and Jacobian of one correspondence */
void orc_debug_point(int kind, const orc_model *m, double scale_reproj, const double *x1, const double *x2, double d1,
                     double d2, double r[7], double J[5 * NPAR]) {
    lm_state st;
    state_from_model(m, kind, &st);
    point_residuals(&st, sqrt(scale_reproj), x1, x2, d1, d2, r, (double(*)[NPAR])J);
}
void orc_debug_step(int kind, const orc_model *m, const double full_delta[NPAR], orc_model *o) {
    lm_problem pb;
    memset(&pb, 0, sizeof pb);
    pb.kind = kind;
    pb.np = NPAR;
    pb.estimate_shift = 1;
    for (int p = 0; p < NPAR; ++p) pb.idx[p] = p;
    /* for the shared kind lm_step would overwrite full[10]; use VARYING semantics for raw steps */
    const int k0 = kind;
    if (kind == ORC_SHARED) pb.kind = ORC_VARYING;
    lm_step(&pb, m, full_delta, o);
    if (k0 != ORC_CALIB) { o->shift1 = m->shift1 + full_delta[7]; o->shift2 = m->shift2 + full_delta[8]; }
}
