/* orc_sixpt.c — CPU oracle (TEST INFRASTRUCTURE ONLY, never linked into the product): the 6-point relative pose solver with one
 * unknown shared focal length, relpose_6pt_shared_focal of the reference binary (PoseLib 2.0.5, @0x1a6b00 region; used by
 * SharedFocalRelativePoseEstimator::generate_models, /root/reference/eval_shared_f.py:161 estimate_shared_focal_relative_pose).
 *
 * The binary solves the problem with a generated elimination template and the eigenvectors of a 15 x 15 action matrix
 * (Eigen::EigenSolver); that template exists only as machine code.  The SOLUTION SET is a property of the polynomial system,
 * not of the template, so this file restates the published formulation instead (Stewenius, Nister, Kahl, Schaffalitzky 2005;
 * as a polynomial eigenvalue problem: Kukelova, Bujnak, Pajdla BMVC 2008):
 *   F = x N0 + y N1 + N2 on the 3-dimensional null space of the six epipolar constraints, Q = diag(1, 1, w), w = 1 / f^2,
 *   2 F Q F' Q F - tr(F Q F' Q) F = 0 (nine cubics in x, y; quadratic in w) and det F = 0:
 *   (M0 + w M1 + w^2 M2) v = 0 with v the ten monomials of degree <= 3 in (x, y).
 * In u = 1 / w = f^2 the leading matrix M0 is regular and the problem is the standard eigenvalue problem of the 20 x 20
 * companion matrix [[0, I], [-M0^-1 M2, -M0^-1 M1]] (five eigenvalues are the spurious u = 0).  Real u > 0 -> f = sqrt(u), the
 * null vector of M(w) -> (x, y) -> F -> E = diag(1,1,1/f) F diag(1,1,1/f) -> motion_from_essential with the cheirality of all six
 * points.  What is pinned against the binary: the solution SETS (tests/golden/classic.npz sixpt_*); the ORDER of the binary's
 * solutions is the order of Eigen's eigenvalues and is not reproduced (solutions come out by ascending focal length). */
#include "mdrp_oracle.h"
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ real eigenvalues, n <= 20
 * Reduction to Hessenberg form by stabilised elementary similarity transformations, then the shifted QR iteration on the
 * Hessenberg matrix (the EISPACK pair elmhes / hqr, restated).  a: n x n row-major, destroyed.  Returns 0 on convergence. */
#define EN 20
/* The five spurious eigenvalues u = 0 (w = infinity) form one defective cluster; rounding splits it into a ring of radius
 * ~eps^(1/5) |C|: u ~ 1e-7 on scale-normalised points.  A focal length below 0.003 in units of the mean point radius is not a
 * camera: everything below is the cluster. */
#define SIX_MIN_U 1e-5
static double sgn_of(double a, double b) { return b >= 0.0 ? fabs(a) : -fabs(a); }

int orc_eigenvalues(double *a, int n, double *wr, double *wi) {
#define A(i, j) a[(i) * n + (j)]
    for (int m = 1; m < n - 1; ++m) {
        double x = 0.0;
        int piv = m;
        for (int j = m; j < n; ++j)
            if (fabs(A(j, m - 1)) > fabs(x)) { x = A(j, m - 1); piv = j; }
        if (piv != m) {
            for (int j = m - 1; j < n; ++j) { const double t = A(piv, j); A(piv, j) = A(m, j); A(m, j) = t; }
            for (int j = 0; j < n; ++j) { const double t = A(j, piv); A(j, piv) = A(j, m); A(j, m) = t; }
        }
        if (x != 0.0)
            for (int i = m + 1; i < n; ++i) {
                double y = A(i, m - 1);
                if (y != 0.0) {
                    y /= x;
                    A(i, m - 1) = y;
                    for (int j = m; j < n; ++j) A(i, j) -= y * A(m, j);
                    for (int j = 0; j < n; ++j) A(j, m) += y * A(j, i);
                }
            }
    }
    for (int i = 2; i < n; ++i)
        for (int j = 0; j < i - 1; ++j) A(i, j) = 0.0;

    int nn = n - 1, its = 0;
    double t = 0.0, anorm = 0.0, p = 0, q = 0, r = 0, s, w, x, y, z;
    for (int i = 0; i < n; ++i)
        for (int j = (i > 0 ? i - 1 : 0); j < n; ++j) anorm += fabs(A(i, j));
    while (nn >= 0) {
        int l;
        for (l = nn; l >= 1; --l) {
            s = fabs(A(l - 1, l - 1)) + fabs(A(l, l));
            if (s == 0.0) s = anorm;
            if (fabs(A(l, l - 1)) + s == s) { A(l, l - 1) = 0.0; break; }
        }
        x = A(nn, nn);
        if (l == nn) { wr[nn] = x + t; wi[nn] = 0.0; --nn; its = 0; continue; }
        y = A(nn - 1, nn - 1);
        w = A(nn, nn - 1) * A(nn - 1, nn);
        if (l == nn - 1) {
            p = 0.5 * (y - x);
            q = p * p + w;
            z = sqrt(fabs(q));
            x += t;
            if (q >= 0.0) {
                z = p + sgn_of(z, p);
                wr[nn - 1] = wr[nn] = x + z;
                if (z != 0.0) wr[nn] = x - w / z;
                wi[nn - 1] = wi[nn] = 0.0;
            } else {
                wr[nn - 1] = wr[nn] = x + p;
                wi[nn - 1] = z; wi[nn] = -z;
            }
            nn -= 2; its = 0;
            continue;
        }
        if (its == 60) return 1;
        if (its == 10 || its == 20 || its == 30 || its == 40) { /* exceptional shift */
            t += x;
            for (int i = 0; i <= nn; ++i) A(i, i) -= x;
            s = fabs(A(nn, nn - 1)) + fabs(A(nn - 1, nn - 2));
            y = x = 0.75 * s;
            w = -0.4375 * s * s;
        }
        ++its;
        int m;
        for (m = nn - 2; m >= l; --m) {
            z = A(m, m);
            r = x - z; s = y - z;
            p = (r * s - w) / A(m + 1, m) + A(m, m + 1);
            q = A(m + 1, m + 1) - z - r - s;
            r = A(m + 2, m + 1);
            s = fabs(p) + fabs(q) + fabs(r);
            p /= s; q /= s; r /= s;
            if (m == l) break;
            const double u = fabs(A(m, m - 1)) * (fabs(q) + fabs(r));
            const double v = fabs(p) * (fabs(A(m - 1, m - 1)) + fabs(z) + fabs(A(m + 1, m + 1)));
            if (u + v == v) break;
        }
        for (int i = m + 2; i <= nn; ++i) { A(i, i - 2) = 0.0; if (i != m + 2) A(i, i - 3) = 0.0; }
        for (int k = m; k <= nn - 1; ++k) {
            if (k != m) {
                p = A(k, k - 1); q = A(k + 1, k - 1); r = 0.0;
                if (k != nn - 1) r = A(k + 2, k - 1);
                if ((x = fabs(p) + fabs(q) + fabs(r)) != 0.0) { p /= x; q /= x; r /= x; }
            }
            if ((s = sgn_of(sqrt(p * p + q * q + r * r), p)) != 0.0) {
                if (k == m) { if (l != m) A(k, k - 1) = -A(k, k - 1); }
                else A(k, k - 1) = -s * x;
                p += s; x = p / s; y = q / s; z = r / s; q /= p; r /= p;
                for (int j = k; j <= nn; ++j) {
                    p = A(k, j) + q * A(k + 1, j);
                    if (k != nn - 1) { p += r * A(k + 2, j); A(k + 2, j) -= p * z; }
                    A(k + 1, j) -= p * y; A(k, j) -= p * x;
                }
                const int mmin = nn < k + 3 ? nn : k + 3;
                for (int i = l; i <= mmin; ++i) {
                    p = x * A(i, k) + y * A(i, k + 1);
                    if (k != nn - 1) { p += z * A(i, k + 2); A(i, k + 2) -= p * r; }
                    A(i, k + 1) -= p * q; A(i, k) -= p;
                }
            }
        }
    }
    return 0;
#undef A
}

/* ------------------------------------------------------------------------------------------------ polynomials in (x, y), degree <= 3
 * monomial order: x3, x2y, xy2, y3, x2, xy, y2, x, y, 1 */
static const int SIX_MON[10][2] = {{3, 0}, {2, 1}, {1, 2}, {0, 3}, {2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};
static int six_idx(int a, int b) {
    for (int i = 0; i < 10; ++i) if (SIX_MON[i][0] == a && SIX_MON[i][1] == b) return i;
    return -1;
}
typedef struct { double c[10]; } spoly;
static void sp_mul_add(const spoly *a, const spoly *b, double s, spoly *out) { /* out += s a b (degrees above 3 do not occur) */
    for (int i = 0; i < 10; ++i) {
        if (a->c[i] == 0.0) continue;
        for (int j = 0; j < 10; ++j) {
            if (b->c[j] == 0.0) continue;
            const int k = six_idx(SIX_MON[i][0] + SIX_MON[j][0], SIX_MON[i][1] + SIX_MON[j][1]);
            if (k >= 0) out->c[k] += s * a->c[i] * b->c[j];
        }
    }
}

/* null vector of a 10 x 10 matrix (row-major, destroyed): Gaussian elimination with complete pivoting; the last pivot is taken
 * as zero.  Returns the ratio |last pivot| / |first pivot| (a residual of the eigenvalue it was built from). */
static double six_null_vector(double *M, double *v) {
    int cp[10];
    for (int i = 0; i < 10; ++i) cp[i] = i;
    double first = 0.0, last = 0.0;
    for (int k = 0; k < 10; ++k) {
        int pr = k, pc = k;
        double best = -1.0;
        for (int i = k; i < 10; ++i)
            for (int j = k; j < 10; ++j)
                if (fabs(M[i * 10 + j]) > best) { best = fabs(M[i * 10 + j]); pr = i; pc = j; }
        if (k == 0) first = best;
        if (k == 9) { last = best; break; }
        if (pr != k) for (int j = 0; j < 10; ++j) { const double t = M[pr * 10 + j]; M[pr * 10 + j] = M[k * 10 + j]; M[k * 10 + j] = t; }
        if (pc != k) { for (int i = 0; i < 10; ++i) { const double t = M[i * 10 + pc]; M[i * 10 + pc] = M[i * 10 + k]; M[i * 10 + k] = t; } const int t = cp[pc]; cp[pc] = cp[k]; cp[k] = t; }
        const double piv = M[k * 10 + k];
        if (piv == 0.0) break;
        for (int i = k + 1; i < 10; ++i) {
            const double f = M[i * 10 + k] / piv;
            if (f != 0.0) for (int j = k; j < 10; ++j) M[i * 10 + j] -= f * M[k * 10 + j];
        }
    }
    double y[10];
    y[9] = 1.0;
    for (int i = 8; i >= 0; --i) {
        double s = 0.0;
        for (int j = i + 1; j < 10; ++j) s += M[i * 10 + j] * y[j];
        y[i] = M[i * 10 + i] != 0.0 ? -s / M[i * 10 + i] : 0.0;
    }
    for (int i = 0; i < 10; ++i) v[cp[i]] = y[i];
    return first > 0.0 ? last / first : 1.0;
}

/* Gauss-Newton on the ten equations in (x, y, w): the eigenvalue and the null vector are good to ~1e-10 on well-conditioned
 * samples and worse on others; three steps bring the residual to rounding level. */
static void six_polish(const double *M0, const double *M1, const double *M2, double *px, double *py, double *pw) {
    for (int it = 0; it < 3; ++it) {
        const double x = *px, y = *py, w = *pw;
        double mono[10], dmx[10], dmy[10];
        for (int e = 0; e < 10; ++e) {
            const int a = SIX_MON[e][0], b = SIX_MON[e][1];
            mono[e] = pow(x, a) * pow(y, b);
            dmx[e] = a > 0 ? a * pow(x, a - 1) * pow(y, b) : 0.0;
            dmy[e] = b > 0 ? b * pow(x, a) * pow(y, b - 1) : 0.0;
        }
        double JtJ[9] = {0}, Jtr[3] = {0};
        for (int r = 0; r < 10; ++r) {
            double g = 0, gx = 0, gy = 0, gw = 0;
            for (int e = 0; e < 10; ++e) {
                const double c = M0[r * 10 + e] + w * (M1[r * 10 + e] + w * M2[r * 10 + e]);
                g += c * mono[e]; gx += c * dmx[e]; gy += c * dmy[e];
                gw += (M1[r * 10 + e] + 2.0 * w * M2[r * 10 + e]) * mono[e];
            }
            const double J[3] = {gx, gy, gw};
            for (int a = 0; a < 3; ++a) { Jtr[a] += J[a] * g; for (int b = 0; b < 3; ++b) JtJ[3 * a + b] += J[a] * J[b]; }
        }
        const double c00 = JtJ[4] * JtJ[8] - JtJ[5] * JtJ[7], c01 = JtJ[5] * JtJ[6] - JtJ[3] * JtJ[8], c02 = JtJ[3] * JtJ[7] - JtJ[4] * JtJ[6];
        const double det = JtJ[0] * c00 + JtJ[1] * c01 + JtJ[2] * c02;
        if (!(fabs(det) > 0.0)) return;
        const double id = 1.0 / det;
        const double dx = (c00 * Jtr[0] + (JtJ[2] * JtJ[7] - JtJ[1] * JtJ[8]) * Jtr[1] + (JtJ[1] * JtJ[5] - JtJ[2] * JtJ[4]) * Jtr[2]) * id;
        const double dy = (c01 * Jtr[0] + (JtJ[0] * JtJ[8] - JtJ[2] * JtJ[6]) * Jtr[1] + (JtJ[2] * JtJ[3] - JtJ[0] * JtJ[5]) * Jtr[2]) * id;
        const double dw = (c02 * Jtr[0] + (JtJ[1] * JtJ[6] - JtJ[0] * JtJ[7]) * Jtr[1] + (JtJ[0] * JtJ[4] - JtJ[1] * JtJ[3]) * Jtr[2]) * id;
        *px = x - dx; *py = y - dy; *pw = w - dw;
    }
}

/* x1h, x2h: six homogeneous image points each (the estimator passes unit vectors of principal-point-centred, scale-normalised
 * pixels).  out: up to 60 models (pose, f1 = f2 = f), by ascending f.  Returns their number. */
int orc_relpose_6pt(const double *x1h, const double *x2h, orc_model *out) {
    /* null space of the 6 x 9 constraint matrix: the last three columns of Q */
    double A[9 * 6], Qm[81], N[3][9];
    for (int p = 0; p < 6; ++p)
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) A[p * 9 + 3 * i + j] = x2h[3 * p + i] * x1h[3 * p + j]; /* entry 3 i + j multiplies F(i, j): x2' F x1 */
    orc_fullpiv_qr_Q(A, 6, Qm);
    for (int k = 0; k < 3; ++k)
        for (int e = 0; e < 9; ++e) N[k][e] = Qm[(6 + k) * 9 + e];
    spoly F[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            memset(&F[i][j], 0, sizeof(spoly));
            F[i][j].c[7] = N[0][3 * i + j]; F[i][j].c[8] = N[1][3 * i + j]; F[i][j].c[9] = N[2][3 * i + j];
        }
    /* S = F Q F' = S0 + w S1 */
    spoly S0[3][3], S1[3][3];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
            memset(&S0[i][k], 0, sizeof(spoly)); memset(&S1[i][k], 0, sizeof(spoly));
            sp_mul_add(&F[i][0], &F[k][0], 1.0, &S0[i][k]); sp_mul_add(&F[i][1], &F[k][1], 1.0, &S0[i][k]);
            sp_mul_add(&F[i][2], &F[k][2], 1.0, &S1[i][k]);
        }
    spoly tr0, tr1, tr2;
    memset(&tr0, 0, sizeof tr0); memset(&tr1, 0, sizeof tr1); memset(&tr2, 0, sizeof tr2);
    for (int e = 0; e < 10; ++e) {
        tr0.c[e] = S0[0][0].c[e] + S0[1][1].c[e];
        tr1.c[e] = S1[0][0].c[e] + S1[1][1].c[e] + S0[2][2].c[e];
        tr2.c[e] = S1[2][2].c[e];
    }
    double M0[100], M1[100], M2[100];
    memset(M0, 0, sizeof M0); memset(M1, 0, sizeof M1); memset(M2, 0, sizeof M2);
    for (int i = 0; i < 3; ++i)
        for (int l = 0; l < 3; ++l) {
            spoly t0, t1, t2;
            memset(&t0, 0, sizeof t0); memset(&t1, 0, sizeof t1); memset(&t2, 0, sizeof t2);
            sp_mul_add(&S0[i][0], &F[0][l], 2.0, &t0); sp_mul_add(&S0[i][1], &F[1][l], 2.0, &t0); sp_mul_add(&tr0, &F[i][l], -1.0, &t0);
            sp_mul_add(&S1[i][0], &F[0][l], 2.0, &t1); sp_mul_add(&S1[i][1], &F[1][l], 2.0, &t1); sp_mul_add(&S0[i][2], &F[2][l], 2.0, &t1);
            sp_mul_add(&tr1, &F[i][l], -1.0, &t1);
            sp_mul_add(&S1[i][2], &F[2][l], 2.0, &t2); sp_mul_add(&tr2, &F[i][l], -1.0, &t2);
            for (int e = 0; e < 10; ++e) { M0[(3 * i + l) * 10 + e] = t0.c[e]; M1[(3 * i + l) * 10 + e] = t1.c[e]; M2[(3 * i + l) * 10 + e] = t2.c[e]; }
        }
    { /* det F */
        spoly m, d;
        memset(&d, 0, sizeof d);
        memset(&m, 0, sizeof m); sp_mul_add(&F[1][1], &F[2][2], 1.0, &m); sp_mul_add(&F[1][2], &F[2][1], -1.0, &m); sp_mul_add(&m, &F[0][0], 1.0, &d);
        memset(&m, 0, sizeof m); sp_mul_add(&F[1][0], &F[2][2], 1.0, &m); sp_mul_add(&F[1][2], &F[2][0], -1.0, &m); sp_mul_add(&m, &F[0][1], -1.0, &d);
        memset(&m, 0, sizeof m); sp_mul_add(&F[1][0], &F[2][1], 1.0, &m); sp_mul_add(&F[1][1], &F[2][0], -1.0, &m); sp_mul_add(&m, &F[0][2], 1.0, &d);
        for (int e = 0; e < 10; ++e) M0[90 + e] = d.c[e];
    }
    /* X = M0^-1 [M2 | M1]: LU with partial pivoting on a copy of M0, 20 right-hand sides */
    double L[100], B[200];
    memcpy(L, M0, sizeof L);
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 10; ++j) { B[i * 20 + j] = M2[i * 10 + j]; B[i * 20 + 10 + j] = M1[i * 10 + j]; }
    for (int k = 0; k < 10; ++k) {
        int piv = k;
        for (int i = k + 1; i < 10; ++i) if (fabs(L[i * 10 + k]) > fabs(L[piv * 10 + k])) piv = i;
        if (L[piv * 10 + k] == 0.0) return 0;
        if (piv != k) {
            for (int j = 0; j < 10; ++j) { const double t = L[piv * 10 + j]; L[piv * 10 + j] = L[k * 10 + j]; L[k * 10 + j] = t; }
            for (int j = 0; j < 20; ++j) { const double t = B[piv * 20 + j]; B[piv * 20 + j] = B[k * 20 + j]; B[k * 20 + j] = t; }
        }
        for (int i = k + 1; i < 10; ++i) {
            const double f = L[i * 10 + k] / L[k * 10 + k];
            if (f == 0.0) continue;
            for (int j = k; j < 10; ++j) L[i * 10 + j] -= f * L[k * 10 + j];
            for (int j = 0; j < 20; ++j) B[i * 20 + j] -= f * B[k * 20 + j];
        }
    }
    for (int i = 9; i >= 0; --i)
        for (int j = 0; j < 20; ++j) {
            double s = B[i * 20 + j];
            for (int k = i + 1; k < 10; ++k) s -= L[i * 10 + k] * B[k * 20 + j];
            B[i * 20 + j] = s / L[i * 10 + i];
        }
    double C[EN * EN], wr[EN], wi[EN];
    memset(C, 0, sizeof C);
    for (int i = 0; i < 10; ++i) {
        C[i * EN + 10 + i] = 1.0;
        for (int j = 0; j < 20; ++j) C[(10 + i) * EN + j] = -B[i * 20 + j];
    }
    if (orc_eigenvalues(C, EN, wr, wi)) return 0;
    /* real u > 0, ascending */
    double us[EN];
    int nu = 0;
    for (int i = 0; i < EN; ++i)
        if (fabs(wi[i]) <= 1e-9 * (fabs(wr[i]) + 1e-300) && wr[i] > SIX_MIN_U) us[nu++] = wr[i];
    for (int i = 1; i < nu; ++i) { const double t = us[i]; int j = i - 1; while (j >= 0 && us[j] > t) { us[j + 1] = us[j]; --j; } us[j + 1] = t; }
    int n_out = 0;
    for (int s = 0; s < nu && n_out < 56; ++s) {
        const double w = 1.0 / us[s];
        double Mw[100], v[10];
        for (int e = 0; e < 100; ++e) Mw[e] = M0[e] + w * (M1[e] + w * M2[e]);
        const double res = six_null_vector(Mw, v);
        if (!(res < 1e-6) || !(fabs(v[9]) > 0.0)) continue;
        double x = v[7] / v[9], y = v[8] / v[9], wv = w;
        six_polish(M0, M1, M2, &x, &y, &wv);
        if (!(wv > 0.0)) continue;
        const double f = sqrt(1.0 / wv);
        double E[9], nrm = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const double Fij = x * N[0][3 * i + j] + y * N[1][3 * i + j] + N[2][3 * i + j];
                E[3 * i + j] = Fij * (i == 2 ? 1.0 / f : 1.0) * (j == 2 ? 1.0 / f : 1.0);
                nrm += E[3 * i + j] * E[3 * i + j];
            }
        nrm = sqrt(nrm);
        for (int e = 0; e < 9; ++e) E[e] /= nrm;
        /* bearings K^-1 x, unit length */
        double b1[18], b2[18];
        for (int p = 0; p < 6; ++p) {
            double a[3] = {x1h[3 * p] / f, x1h[3 * p + 1] / f, x1h[3 * p + 2]}, b[3] = {x2h[3 * p] / f, x2h[3 * p + 1] / f, x2h[3 * p + 2]};
            const double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
            for (int c = 0; c < 3; ++c) { b1[3 * p + c] = a[c] / na; b2[3 * p + c] = b[c] / nb; }
        }
        orc_model poses[4];
        const int np = orc_motion_from_essential(E, b1, b2, 6, poses);
        for (int k = 0; k < np; ++k) { poses[k].f1 = poses[k].f2 = f; out[n_out++] = poses[k]; }
    }
    return n_out;
}
