/* orc_classic.c — the non-monodepth baselines of the same binary: 5-point relative pose, 7-point
 * fundamental matrix.  TEST INFRASTRUCTURE (see mdrp_oracle.h).
 *
 * Restates (reference binary demo/poselib-2.0.5-cp312-*.whl!poselib/_core*.so; upstream PoseLib 2.0.5 algorithms):
 *   relpose_5pt (Matrix3d) @0x145900, (CameraPose) @0x14ae80     Nistér 2004: null space, 10 cubic constraints, Gauss-Jordan,
 *                                                               degree-10 polynomial in z, real roots in ascending order
 *   relpose_7pt @0x4ff2e0                                        null space, det(F) = 0 cubic
 *   motion_from_essential @0x1dd540                              closed-form factorisation + cheirality of all sample points
 * The null-space basis is the one Eigen's FullPivHouseholderQR produces (last columns of matrixQ()): the parametrisation —
 * and with it the ORDER of the solutions, which the RANSAC trajectory depends on — follows from it.
 */
#include "mdrp_oracle.h"
#include <float.h>
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ null space
 * Eigen::FullPivHouseholderQR<Matrix<double,9,m>>::computeInPlace + matrixQ().  A: 9 x m column-major (destroyed).
 * Q: 9 x 9 column-major. */
void orc_fullpiv_qr_Q(double *A, int m, double *Q) {
    const int rows = 9, cols = m, size = m;
    double tau[9];
    int rt[9];
    const double prec = DBL_EPSILON * (double)size;
    double biggest = 0.0;
    int k;
    for (k = 0; k < size; ++k) {
        int br = k, bc = k;
        double big = -1.0;
        for (int c = k; c < cols; ++c)
            for (int r = k; r < rows; ++r)
                if (fabs(A[c * 9 + r]) > big) { big = fabs(A[c * 9 + r]); br = r; bc = c; }
        if (k == 0) biggest = big;
        if (big <= biggest * prec) {
            for (int i = k; i < size; ++i) { rt[i] = i; tau[i] = 0.0; }
            break;
        }
        rt[k] = br;
        if (br != k)
            for (int c = k; c < cols; ++c) { const double t = A[c * 9 + k]; A[c * 9 + k] = A[c * 9 + br]; A[c * 9 + br] = t; }
        if (bc != k)
            for (int r = 0; r < rows; ++r) { const double t = A[k * 9 + r]; A[k * 9 + r] = A[bc * 9 + r]; A[bc * 9 + r] = t; }
        double tail = 0.0;
        for (int r = k + 1; r < rows; ++r) tail += A[k * 9 + r] * A[k * 9 + r];
        const double c0 = A[k * 9 + k];
        double beta;
        if (tail <= DBL_MIN) {
            tau[k] = 0.0; beta = c0;
            for (int r = k + 1; r < rows; ++r) A[k * 9 + r] = 0.0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            for (int r = k + 1; r < rows; ++r) A[k * 9 + r] /= (c0 - beta);
            tau[k] = (beta - c0) / beta;
        }
        A[k * 9 + k] = beta;
        for (int c = k + 1; c < cols; ++c) {
            double tmp = 0.0;
            for (int r = k + 1; r < rows; ++r) tmp += A[k * 9 + r] * A[c * 9 + r];
            tmp += A[c * 9 + k];
            A[c * 9 + k] -= tau[k] * tmp;
            for (int r = k + 1; r < rows; ++r) A[c * 9 + r] -= tau[k] * A[k * 9 + r] * tmp;
        }
    }
    for (int i = 0; i < 81; ++i) Q[i] = 0.0;
    for (int i = 0; i < 9; ++i) Q[i * 9 + i] = 1.0;
    for (k = size - 1; k >= 0; --k) {
        for (int c = k; c < rows; ++c) {
            double tmp = 0.0;
            for (int r = k + 1; r < rows; ++r) tmp += A[k * 9 + r] * Q[c * 9 + r];
            tmp += Q[c * 9 + k];
            Q[c * 9 + k] -= tau[k] * tmp;
            for (int r = k + 1; r < rows; ++r) Q[c * 9 + r] -= tau[k] * A[k * 9 + r] * tmp;
        }
        if (rt[k] != k)
            for (int c = 0; c < rows; ++c) { const double t = Q[c * 9 + k]; Q[c * 9 + k] = Q[c * 9 + rt[k]]; Q[c * 9 + rt[k]] = t; }
    }
}

/* epipolar constraint rows kron(x1, x2): entry 3 j + i multiplies E(i, j) (x2' E x1 = 0) */
static void epipolar_matrix(const double *x1h, const double *x2h, int m, double *A /*9 x m col-major*/) {
    for (int p = 0; p < m; ++p)
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) A[p * 9 + 3 * j + i] = x1h[3 * p + j] * x2h[3 * p + i];
}

/* ------------------------------------------------------------------------------------------------ polynomials in x, y, z
 * 20 monomials of degree <= 3, Nistér's order: the first ten are eliminated, the last ten are [x, y, 1] (x) powers of z */
static const int MONO[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {1, 1, 1}, {1, 1, 0}, {0, 2, 1}, {0, 2, 0},
                                {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
static int mono_index(int a, int b, int c) {
    for (int i = 0; i < 20; ++i)
        if (MONO[i][0] == a && MONO[i][1] == b && MONO[i][2] == c) return i;
    return -1;
}
typedef struct { double c[20]; } poly3;
static void p_zero(poly3 *p) { memset(p, 0, sizeof *p); }
static void p_mul_add(const poly3 *a, const poly3 *b, double s, poly3 *out) { /* out += s a b (terms above degree 3 cannot occur) */
    for (int i = 0; i < 20; ++i) {
        if (a->c[i] == 0.0) continue;
        for (int j = 0; j < 20; ++j) {
            if (b->c[j] == 0.0) continue;
            const int idx = mono_index(MONO[i][0] + MONO[j][0], MONO[i][1] + MONO[j][1], MONO[i][2] + MONO[j][2]);
            if (idx >= 0) out->c[idx] += s * a->c[i] * b->c[j];
        }
    }
}

/* real roots of a polynomial of degree <= 10 in ascending order: Sturm chain for isolation, bisection + Newton to polish */
static int sturm_changes(double chain[12][12], const int *deg, int nchain, double x) {
    int changes = 0, last = 0;
    for (int i = 0; i < nchain; ++i) {
        double v = 0.0;
        for (int k = deg[i]; k >= 0; --k) v = v * x + chain[i][k];
        const int s = (v > 0) - (v < 0);
        if (s != 0) { if (last != 0 && s != last) ++changes; last = s; }
    }
    return changes;
}
static double poly_eval(const double *c, int deg, double x) { double v = 0.0; for (int k = deg; k >= 0; --k) v = v * x + c[k]; return v; }

static void isolate(double chain[12][12], const int *deg, int nchain, double lo, double hi, int clo, int chi, double *roots, int *nr, int depth) {
    const int n = clo - chi;
    if (n <= 0) return;
    if (n == 1 || depth > 200 || hi - lo < 1e-15 * fmax(1.0, fmax(fabs(lo), fabs(hi)))) {
        if (n == 1) {
            /* one root in (lo, hi]: bisection on the sign of p, then Newton */
            const double *p = chain[0];
            const int d = deg[0];
            double a = lo, b = hi, fa = poly_eval(p, d, a);
            for (int it = 0; it < 200; ++it) {
                const double mid = 0.5 * (a + b);
                if (mid == a || mid == b) break;
                const double fm = poly_eval(p, d, mid);
                if (fm == 0.0) { a = b = mid; break; }
                if ((fm > 0) == (fa > 0) && fa != 0.0) { a = mid; fa = fm; } else b = mid;
            }
            double x = 0.5 * (a + b);
            for (int it = 0; it < 3; ++it) {
                double v = 0.0, dv = 0.0;
                for (int k = d; k >= 0; --k) { dv = dv * x + v; v = v * x + p[k]; }
                if (dv != 0.0) { const double xn = x - v / dv; if (xn >= lo && xn <= hi) x = xn; }
            }
            roots[(*nr)++] = x;
        } else {
            for (int i = 0; i < n; ++i) roots[(*nr)++] = 0.5 * (lo + hi); /* multiple / unresolvable cluster */
        }
        return;
    }
    const double mid = 0.5 * (lo + hi);
    const int cm = sturm_changes(chain, deg, nchain, mid);
    isolate(chain, deg, nchain, lo, mid, clo, cm, roots, nr, depth + 1);
    isolate(chain, deg, nchain, mid, hi, cm, chi, roots, nr, depth + 1);
}

int orc_real_roots(const double *coef, int degree, double *roots) {
    static const int MAXD = 10;
    int d = degree;
    while (d > 0 && coef[d] == 0.0) --d;
    if (d <= 0 || d > MAXD) return 0;
    double chain[12][12];
    int deg[12], nchain = 2;
    memset(chain, 0, sizeof chain);
    for (int k = 0; k <= d; ++k) chain[0][k] = coef[k] / coef[d];
    deg[0] = d;
    for (int k = 1; k <= d; ++k) chain[1][k - 1] = k * chain[0][k];
    deg[1] = d - 1;
    while (deg[nchain - 1] > 0) {
        /* remainder of chain[n-2] / chain[n-1], negated */
        double r[12];
        memcpy(r, chain[nchain - 2], sizeof r);
        const double *q = chain[nchain - 1];
        const int dq = deg[nchain - 1];
        for (int k = deg[nchain - 2]; k >= dq; --k) {
            const double f = r[k] / q[dq];
            for (int j = 0; j <= dq; ++j) r[k - dq + j] -= f * q[j];
            r[k] = 0.0;
        }
        int dr = dq - 1;
        while (dr > 0 && fabs(r[dr]) < 1e-300) --dr;
        double mx = 0.0;
        for (int k = 0; k <= dr; ++k) mx = fmax(mx, fabs(r[k]));
        if (mx == 0.0) break;
        for (int k = 0; k <= dr; ++k) chain[nchain][k] = -r[k] / mx;
        deg[nchain] = dr;
        ++nchain;
    }
    double bound = 0.0;
    for (int k = 0; k < d; ++k) bound = fmax(bound, fabs(chain[0][k]));
    bound += 1.0;
    int nr = 0;
    isolate(chain, deg, nchain, -bound, bound, sturm_changes(chain, deg, nchain, -bound), sturm_changes(chain, deg, nchain, bound), roots, &nr, 0);
    return nr;
}

/* ------------------------------------------------------------------------------------------------ 5-point
 * x1h, x2h: 5 bearings each (row-major 5 x 3).  E_out: up to 10 matrices, row-major, Frobenius norm 1. */
int orc_relpose_5pt_E(const double *x1h, const double *x2h, double *E_out) {
    double A[9 * 5], Q[81];
    epipolar_matrix(x1h, x2h, 5, A);
    orc_fullpiv_qr_Q(A, 5, Q);
    const double *N = Q + 9 * 5; /* four basis vectors, each the column-major vec of a 3 x 3 matrix */
    /* E(i, j) = x N0 + y N1 + z N2 + N3 as linear polynomials */
    poly3 E[3][3];
    static const int LIN[4] = {12, 15, 18, 19};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            p_zero(&E[i][j]);
            for (int k = 0; k < 4; ++k) E[i][j].c[LIN[k]] = N[k * 9 + 3 * j + i];
        }
    poly3 EEt[3][3], tr;
    p_zero(&tr);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            p_zero(&EEt[i][j]);
            for (int k = 0; k < 3; ++k) p_mul_add(&E[i][k], &E[j][k], 1.0, &EEt[i][j]);
        }
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 20; ++k) tr.c[k] += EEt[i][i].c[k];
    double C[10][20];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            poly3 t;
            p_zero(&t);
            for (int k = 0; k < 3; ++k) p_mul_add(&EEt[i][k], &E[k][j], 2.0, &t);
            p_mul_add(&tr, &E[i][j], -1.0, &t);
            memcpy(C[3 * i + j], t.c, sizeof t.c);
        }
    {
        poly3 det, m;
        p_zero(&det);
        p_zero(&m); p_mul_add(&E[1][1], &E[2][2], 1.0, &m); p_mul_add(&E[1][2], &E[2][1], -1.0, &m); p_mul_add(&E[0][0], &m, 1.0, &det);
        p_zero(&m); p_mul_add(&E[1][0], &E[2][2], 1.0, &m); p_mul_add(&E[1][2], &E[2][0], -1.0, &m); p_mul_add(&E[0][1], &m, -1.0, &det);
        p_zero(&m); p_mul_add(&E[1][0], &E[2][1], 1.0, &m); p_mul_add(&E[1][1], &E[2][0], -1.0, &m); p_mul_add(&E[0][2], &m, 1.0, &det);
        memcpy(C[9], det.c, sizeof det.c);
    }
    /* Gauss-Jordan on the ten cubic monomials (partial pivoting) */
    for (int col = 0; col < 10; ++col) {
        int piv = col;
        for (int r = col + 1; r < 10; ++r) if (fabs(C[r][col]) > fabs(C[piv][col])) piv = r;
        if (C[piv][col] == 0.0) return 0;
        if (piv != col) for (int k = 0; k < 20; ++k) { const double t = C[col][k]; C[col][k] = C[piv][k]; C[piv][k] = t; }
        const double inv = 1.0 / C[col][col];
        for (int k = 0; k < 20; ++k) C[col][k] *= inv;
        for (int r = 0; r < 10; ++r) {
            if (r == col) continue;
            const double f = C[r][col];
            if (f != 0.0) for (int k = 0; k < 20; ++k) C[r][k] -= f * C[col][k];
        }
    }
    /* rows (4,5), (6,7), (8,9): <x^2 z> - z <x^2>, <xyz> - z <xy>, <y^2 z> - z <y^2>  ->  B(z) [x y 1]' = 0 */
    double bx[3][4], by[3][4], b1[3][5];
    for (int i = 0; i < 3; ++i) {
        const double *u = C[4 + 2 * i] + 10, *v = C[5 + 2 * i] + 10;
        bx[i][3] = -v[0]; bx[i][2] = u[0] - v[1]; bx[i][1] = u[1] - v[2]; bx[i][0] = u[2];
        by[i][3] = -v[3]; by[i][2] = u[3] - v[4]; by[i][1] = u[4] - v[5]; by[i][0] = u[5];
        b1[i][4] = -v[6]; b1[i][3] = u[6] - v[7]; b1[i][2] = u[7] - v[8]; b1[i][1] = u[8] - v[9]; b1[i][0] = u[9];
    }
    double c[11];
    memset(c, 0, sizeof c);
    static const int PERM[6][4] = {{0, 1, 2, 1}, {0, 2, 1, -1}, {1, 0, 2, -1}, {1, 2, 0, 1}, {2, 0, 1, 1}, {2, 1, 0, -1}};
    for (int p = 0; p < 6; ++p) { /* sign * bx[r0] * by[r1] * b1[r2] */
        const double *a = bx[PERM[p][0]], *b = by[PERM[p][1]], *d = b1[PERM[p][2]];
        double ab[7];
        memset(ab, 0, sizeof ab);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) ab[i + j] += a[i] * b[j];
        for (int i = 0; i < 7; ++i) for (int j = 0; j < 5; ++j) c[i + j] += PERM[p][3] * ab[i] * d[j];
    }
    double roots[10];
    const int nr = orc_real_roots(c, 10, roots);
    int n_out = 0;
    for (int s = 0; s < nr; ++s) {
        const double z = roots[s];
        double B[3][3];
        for (int i = 0; i < 3; ++i) {
            B[i][0] = ((bx[i][3] * z + bx[i][2]) * z + bx[i][1]) * z + bx[i][0];
            B[i][1] = ((by[i][3] * z + by[i][2]) * z + by[i][1]) * z + by[i][0];
            B[i][2] = (((b1[i][4] * z + b1[i][3]) * z + b1[i][2]) * z + b1[i][1]) * z + b1[i][0];
        }
        /* [x y] from the best conditioned pair of rows */
        int r0 = 0, r1 = 1;
        double best = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int b = a + 1; b < 3; ++b) {
                const double d = fabs(B[a][0] * B[b][1] - B[a][1] * B[b][0]);
                if (d > best) { best = d; r0 = a; r1 = b; }
            }
        if (best == 0.0) continue;
        const double det = B[r0][0] * B[r1][1] - B[r0][1] * B[r1][0];
        const double x = (-B[r0][2] * B[r1][1] + B[r0][1] * B[r1][2]) / det;
        const double y = (-B[r0][0] * B[r1][2] + B[r0][2] * B[r1][0]) / det;
        double e[9], nrm = 0.0;
        for (int k = 0; k < 9; ++k) { e[k] = x * N[k] + y * N[9 + k] + z * N[18 + k] + N[27 + k]; nrm += e[k] * e[k]; }
        nrm = sqrt(nrm);
        double *Eo = E_out + 9 * n_out++;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Eo[3 * i + j] = e[3 * j + i] / nrm;
    }
    return n_out;
}

/* ------------------------------------------------------------------------------------------------ E -> poses
 * motion_from_essential @0x1dd540: U W and V from cross products of E's columns, four candidates in the order
 * (R1, t), (R1, -t), (R2, -t), (R2, t); a candidate is kept iff every sample point has positive depth in both views. */
static void cross3(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static int cheirality_all(const orc_model *m, const double *x1h, const double *x2h, int npts) {
    for (int k = 0; k < npts; ++k)
        if (!orc_check_cheirality(m, x1h + 3 * k, x2h + 3 * k, 0.0)) return 0;
    return 1;
}
int orc_motion_from_essential(const double *E /*row-major*/, const double *x1h, const double *x2h, int npts, orc_model *out) {
    double c0[3] = {E[0], E[3], E[6]}, c1[3] = {E[1], E[4], E[7]}, c2[3] = {E[2], E[5], E[8]};
    double u12[3], u13[3], u23[3];
    cross3(c0, c1, u12); cross3(c0, c2, u13); cross3(c1, c2, u23);
    const double n12 = u12[0] * u12[0] + u12[1] * u12[1] + u12[2] * u12[2];
    const double n13 = u13[0] * u13[0] + u13[1] * u13[1] + u13[2] * u13[2];
    const double n23 = u23[0] * u23[0] + u23[1] * u23[1] + u23[2] * u23[2];
    double UW[3][3], Vt[3][3]; /* UW columns as UW[col][.] */
    const double *a, *u;
    double nu;
    if (n12 > n13) {
        if (n12 > n23) { a = c0; u = u12; nu = n12; } else { a = c1; u = u23; nu = n23; }
    } else {
        if (n13 > n23) { a = c0; u = u13; nu = n13; } else { a = c1; u = u23; nu = n23; }
    }
    const double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), su = sqrt(nu);
    for (int k = 0; k < 3; ++k) { UW[1][k] = a[k] / na; UW[2][k] = u[k] / su; }
    double t0[3];
    cross3(UW[2], UW[1], t0);
    for (int k = 0; k < 3; ++k) UW[0][k] = -t0[k];
    for (int j = 0; j < 3; ++j) {
        Vt[0][j] = UW[1][0] * E[j] + UW[1][1] * E[3 + j] + UW[1][2] * E[6 + j];
        Vt[1][j] = -(UW[0][0] * E[j] + UW[0][1] * E[3 + j] + UW[0][2] * E[6 + j]);
    }
    double n0 = sqrt(Vt[0][0] * Vt[0][0] + Vt[0][1] * Vt[0][1] + Vt[0][2] * Vt[0][2]);
    for (int j = 0; j < 3; ++j) Vt[0][j] /= n0;
    const double d = Vt[0][0] * Vt[1][0] + Vt[0][1] * Vt[1][1] + Vt[0][2] * Vt[1][2];
    for (int j = 0; j < 3; ++j) Vt[1][j] -= d * Vt[0][j];
    n0 = sqrt(Vt[1][0] * Vt[1][0] + Vt[1][1] * Vt[1][1] + Vt[1][2] * Vt[1][2]);
    for (int j = 0; j < 3; ++j) Vt[1][j] /= n0;
    cross3(Vt[0], Vt[1], Vt[2]);
    int n_out = 0;
    orc_model m;
    memset(&m, 0, sizeof m);
    m.scale = 1.0; m.f1 = m.f2 = 1.0;
    for (int pass = 0; pass < 2; ++pass) {
        double R[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[3 * i + j] = UW[0][i] * Vt[0][j] + UW[1][i] * Vt[1][j] + UW[2][i] * Vt[2][j];
        orc_rotmat_to_quat(R, m.q);
        if (pass == 0) for (int k = 0; k < 3; ++k) m.t[k] = UW[2][k];
        if (cheirality_all(&m, x1h, x2h, npts)) out[n_out++] = m;
        for (int k = 0; k < 3; ++k) m.t[k] = -m.t[k];
        if (cheirality_all(&m, x1h, x2h, npts)) out[n_out++] = m;
        for (int k = 0; k < 3; ++k) { UW[0][k] = -UW[0][k]; UW[1][k] = -UW[1][k]; }
    }
    return n_out;
}

int orc_relpose_5pt(const double *x1h, const double *x2h, orc_model *out /*up to 40*/) {
    double Es[90];
    const int ne = orc_relpose_5pt_E(x1h, x2h, Es);
    int n = 0;
    for (int i = 0; i < ne; ++i) n += orc_motion_from_essential(Es + 9 * i, x1h, x2h, 5, out + n);
    return n;
}

/* ------------------------------------------------------------------------------------------------ 7-point
 * F = r N0 + N1, det F = 0; roots in the order of the binary's solve_cubic_real; each F scaled to unit Frobenius norm */
int orc_relpose_7pt(const double *x1h, const double *x2h, double *F_out /*3 x 9 row-major*/) {
    double A[9 * 7], Q[81];
    epipolar_matrix(x1h, x2h, 7, A);
    orc_fullpiv_qr_Q(A, 7, Q);
    const double *N0 = Q + 9 * 7, *N1 = Q + 9 * 8;
    /* det(r A + B) with A = mat(N0), B = mat(N1): cubic c3 r^3 + c2 r^2 + c1 r + c0 */
    double c[4] = {0, 0, 0, 0};
    static const int P[6][4] = {{0, 1, 2, 1}, {0, 2, 1, -1}, {1, 0, 2, -1}, {1, 2, 0, 1}, {2, 0, 1, 1}, {2, 1, 0, -1}};
    for (int p = 0; p < 6; ++p) {
        /* entries (0, P0), (1, P1), (2, P2); vec index of (i, j) = 3 j + i */
        double lin[3][2];
        for (int i = 0; i < 3; ++i) { lin[i][1] = N0[3 * P[p][i] + i]; lin[i][0] = N1[3 * P[p][i] + i]; }
        double q[3] = {lin[0][0] * lin[1][0], lin[0][0] * lin[1][1] + lin[0][1] * lin[1][0], lin[0][1] * lin[1][1]};
        c[0] += P[p][3] * q[0] * lin[2][0];
        c[1] += P[p][3] * (q[0] * lin[2][1] + q[1] * lin[2][0]);
        c[2] += P[p][3] * (q[1] * lin[2][1] + q[2] * lin[2][0]);
        c[3] += P[p][3] * q[2] * lin[2][1];
    }
    double roots[3];
    const int nr = orc_real_roots(c, 3, roots);
    for (int s = 0; s < nr; ++s) { /* solve_cubic_real of the binary lists the roots in DESCENDING order (cos(phi/3 - 2 pi k/3), k = 0, 1, 2) */
        double f[9], nrm = 0.0;
        for (int k = 0; k < 9; ++k) { f[k] = N0[k] * roots[nr - 1 - s] + N1[k]; nrm += f[k] * f[k]; }
        nrm = sqrt(nrm);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) F_out[9 * s + 3 * i + j] = f[3 * j + i] / nrm;
    }
    return nr;
}

/* ================================================================================================ refinement
 * refine_relpose @0x258f50 (5 parameters: rotation, translation in the tangent plane of t), refine_shared_focal_relpose
 * @0x258ed0 (+ focal), refine_fundamental @0x2590d0 (7 parameters: F = U diag(1, sigma, 0) V', rotations of U and V, sigma).
 * Sampson residual only; same lm_impl<> loop as orc_refine.c.  Models travel as 12-double blobs (orc_model): kind 3 q, t;
 * kind 4 q, t, f1 = f2 = f; kind 5 the nine entries of F, row-major, in the first nine doubles. */
#define CNP 7
static double c_loss_value(int type, double thr, double r2) {
    const double t2 = thr * thr;
    switch (type) {
    case 0: return r2;
    case 1: return r2 < t2 ? r2 : t2;
    case 2: { const double r = sqrt(r2); return r <= thr ? r2 : thr * (2.0 * r - thr); }
    case 3: return t2 * log1p(r2 / t2);
    case 4: return t2 * log1p((r2 < t2 ? r2 : t2) / t2);
    case 5: return r2 < t2 ? r2 : t2;
    }
    return r2;
}
static double c_loss_weight(int type, double thr, double r2, double mu) {
    const double t2 = thr * thr;
    switch (type) {
    case 0: return 1.0;
    case 1: return r2 < t2 ? 1.0 : 0.0;
    case 2: { const double r = sqrt(r2); return r <= thr ? 1.0 : thr / r; }
    case 3: { const double w = 1.0 / (1.0 + r2 / t2); return w > DBL_MIN ? w : DBL_MIN; }
    case 4: { if (!(r2 < t2)) return 0.0; const double w = 1.0 / (1.0 + r2 / t2); return w > DBL_MIN ? w : DBL_MIN; }
    case 5: {
        const double r2h = r2 / t2;
        if (r2h < 1.0) return 0.5;
        const double zstar = 1.0, r2m1 = r2h - 1.0;
        const double rho = (2.0 * r2m1 + sqrt(4.0 * r2m1 * r2m1 * mu * mu + 2.0 * mu * r2m1)) / mu;
        const double a = (r2h + mu * rho * zstar - 0.5 * rho) / (1.0 + mu * rho);
        const double zbar = a < 0.0 ? 0.0 : (a > 1.0 ? 1.0 : a);
        return (zstar - zbar) / rho;
    }
    }
    return 1.0;
}

static void c_quat_mul(const double a[4], const double b[4], double o[4]) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
static void c_quat_exp(const double w[3], double q[4]) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    if (th > 1e-6) {
        const double re = cos(0.5 * th), im = sin(0.5 * th) / th;
        q[0] = re; q[1] = im * w[0]; q[2] = im * w[1]; q[3] = im * w[2];
    } else {
        const double re = 1.0 - th2 / 8.0, im = 0.5 - th2 / 48.0;
        const double nq = sqrt(re * re + im * im * th2);
        q[0] = re / nq; q[1] = im * w[0] / nq; q[2] = im * w[1] / nq; q[3] = im * w[2] / nq;
    }
}

/* Sampson residual r = C / |J_C| of one correspondence under F (row-major) and G = dr / dF */
static double sampson_residual(const double *F, const double *x1, const double *x2, double *G) {
    const double h1[3] = {x1[0], x1[1], 1.0}, h2[3] = {x2[0], x2[1], 1.0};
    double Fh1[3], Fth2[3];
    for (int i = 0; i < 3; ++i) {
        Fh1[i] = F[3 * i] * h1[0] + F[3 * i + 1] * h1[1] + F[3 * i + 2];
        Fth2[i] = F[i] * h2[0] + F[3 + i] * h2[1] + F[6 + i];
    }
    const double C = h2[0] * Fh1[0] + h2[1] * Fh1[1] + Fh1[2];
    const double den = Fh1[0] * Fh1[0] + Fh1[1] * Fh1[1] + Fth2[0] * Fth2[0] + Fth2[1] * Fth2[1];
    const double isd = 1.0 / sqrt(den);
    if (G) {
        const double k = C * isd * isd * isd;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double g = h2[i] * h1[j] * isd;
                if (i < 2) g -= k * Fh1[i] * h1[j];
                if (j < 2) g -= k * Fth2[j] * h2[i];
                G[3 * i + j] = g;
            }
    }
    return C * isd;
}

/* 3 x 3 SVD by one-sided Jacobi on the columns: A = U diag(s) V', s descending, U and V proper or improper as they come */
static void svd3(const double *A /*row-major*/, double *U, double *s, double *V) {
    double B[9], W[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(B, A, sizeof B);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, c = 0;
                for (int i = 0; i < 3; ++i) { a += B[3 * i + p] * B[3 * i + p]; b += B[3 * i + q] * B[3 * i + q]; c += B[3 * i + p] * B[3 * i + q]; }
                off = fmax(off, fabs(c) / sqrt(fmax(a * b, DBL_MIN)));
                if (fabs(c) <= 1e-300) continue;
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < 3; ++i) {
                    const double bp = B[3 * i + p], bq = B[3 * i + q];
                    B[3 * i + p] = cs * bp - sn * bq; B[3 * i + q] = sn * bp + cs * bq;
                    const double wp = W[3 * i + p], wq = W[3 * i + q];
                    W[3 * i + p] = cs * wp - sn * wq; W[3 * i + q] = sn * wp + cs * wq;
                }
            }
        if (off < 1e-16) break;
    }
    double n[3];
    int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; ++j) n[j] = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
    for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (n[ord[b]] > n[ord[a]]) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    for (int j = 0; j < 3; ++j) {
        const int o = ord[j];
        s[j] = n[o];
        for (int i = 0; i < 3; ++i) { V[3 * i + j] = W[3 * i + o]; U[3 * i + j] = n[o] > 0 ? B[3 * i + o] / n[o] : 0.0; }
    }
    /* third left vector from the first two when the matrix is (numerically) singular */
    if (s[2] <= 1e-12 * s[0]) {
        const double u0[3] = {U[0], U[3], U[6]}, u1[3] = {U[1], U[4], U[7]};
        double u2[3];
        cross3(u0, u1, u2);
        const double nn = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
        for (int i = 0; i < 3; ++i) U[3 * i + 2] = u2[i] / nn;
    }
}
static double det3(const double *M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

typedef struct { double q[4], t[3], f; double qU[4], qV[4], sigma; double tb[6]; /* tangent basis of the last accumulate */ } cmodel;

static void cmodel_F(int kind, const cmodel *m, double *F) {
    if (kind == 5) {
        double U[9], V[9];
        orc_quat_to_rotmat(m->qU, U);
        orc_quat_to_rotmat(m->qV, V);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) F[3 * i + j] = U[3 * i] * V[3 * j] + m->sigma * U[3 * i + 1] * V[3 * j + 1];
        return;
    }
    orc_model o;
    memset(&o, 0, sizeof o);
    memcpy(o.q, m->q, sizeof o.q); memcpy(o.t, m->t, sizeof o.t);
    o.f1 = o.f2 = (kind == 4 ? m->f : 1.0);
    orc_fundamental(&o, F);
}

typedef struct { int kind, n, loss; const double *x1, *x2, *w; double thr, mu; } cproblem;

static double c_cost(const cproblem *pb, const cmodel *m) {
    double F[9], cost = 0.0;
    cmodel_F(pb->kind, m, F);
    for (int k = 0; k < pb->n; ++k) {
        const double r = sampson_residual(F, pb->x1 + 2 * k, pb->x2 + 2 * k, NULL);
        cost += (pb->w ? pb->w[k] : 1.0) * c_loss_value(pb->loss, pb->thr, r * r);
    }
    return cost;
}

static void tangent_basis(const double *t, double *tb /*b0[3] b1[3]*/) {
    double e[3] = {0, 0, 0};
    if (fabs(t[0]) < fabs(t[1])) { if (fabs(t[0]) < fabs(t[2])) e[0] = 1; else e[2] = 1; }
    else { if (fabs(t[1]) < fabs(t[2])) e[1] = 1; else e[2] = 1; }
    cross3(t, e, tb);
    double n = sqrt(tb[0] * tb[0] + tb[1] * tb[1] + tb[2] * tb[2]);
    for (int i = 0; i < 3; ++i) tb[i] /= n;
    cross3(tb, t, tb + 3);
    n = sqrt(tb[3] * tb[3] + tb[4] * tb[4] + tb[5] * tb[5]);
    for (int i = 0; i < 3; ++i) tb[3 + i] /= n;
}

static int c_accumulate(const cproblem *pb, cmodel *m, double *JtJ, double *Jtr) {
    const int np = pb->kind == 3 ? 5 : (pb->kind == 4 ? 6 : 7);
    double F[9], dF[CNP][9]; /* dF / d parameter, row-major */
    cmodel_F(pb->kind, m, F);
    if (pb->kind == 5) {
        double U[9], V[9];
        orc_quat_to_rotmat(m->qU, U);
        orc_quat_to_rotmat(m->qV, V);
        for (int a = 0; a < 3; ++a) {
            const int b = (a + 1) % 3, c = (a + 2) % 3;
            for (int j = 0; j < 3; ++j) { dF[a][3 * a + j] = 0.0; dF[a][3 * b + j] = -F[3 * c + j]; dF[a][3 * c + j] = F[3 * b + j]; } /* [e_a]x F */
            for (int i = 0; i < 3; ++i) { dF[3 + a][3 * i + a] = 0.0; dF[3 + a][3 * i + b] = -F[3 * i + c]; dF[3 + a][3 * i + c] = F[3 * i + b]; } /* F [e_a]x' */
        }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dF[6][3 * i + j] = U[3 * i + 1] * V[3 * j + 1];
    } else {
        double R[9], E[9];
        const double f = pb->kind == 4 ? m->f : 1.0;
        orc_quat_to_rotmat(m->q, R);
        { orc_model o; memset(&o, 0, sizeof o); memcpy(o.q, m->q, sizeof o.q); memcpy(o.t, m->t, sizeof o.t); orc_essential(&o, E); }
        tangent_basis(m->t, m->tb);
        for (int a = 0; a < 3; ++a) { /* E [e_a]x */
            const int b = (a + 1) % 3, c = (a + 2) % 3;
            for (int i = 0; i < 3; ++i) { dF[a][3 * i + a] = 0.0; dF[a][3 * i + b] = E[3 * i + c]; dF[a][3 * i + c] = -E[3 * i + b]; }
        }
        for (int k = 0; k < 2; ++k) { /* [b_k]x R */
            const double *d = m->tb + 3 * k;
            for (int j = 0; j < 3; ++j) {
                dF[3 + k][0 + j] = -d[2] * R[3 + j] + d[1] * R[6 + j];
                dF[3 + k][3 + j] = d[2] * R[0 + j] - d[0] * R[6 + j];
                dF[3 + k][6 + j] = -d[1] * R[0 + j] + d[0] * R[3 + j];
            }
        }
        for (int p = 0; p < 5; ++p)
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dF[p][3 * i + j] *= (i == 2 ? f : 1.0) * (j == 2 ? f : 1.0);
        if (pb->kind == 4) {
            memset(dF[5], 0, sizeof dF[5]);
            dF[5][2] = E[2]; dF[5][5] = E[5]; dF[5][6] = E[6]; dF[5][7] = E[7]; dF[5][8] = 2.0 * E[8] * f;
        }
    }
    memset(JtJ, 0, sizeof(double) * np * np);
    memset(Jtr, 0, sizeof(double) * np);
    int cnt = 0;
    for (int k = 0; k < pb->n; ++k) {
        double G[9], J[CNP];
        const double r = sampson_residual(F, pb->x1 + 2 * k, pb->x2 + 2 * k, G);
        const double w = (pb->w ? pb->w[k] : 1.0) * c_loss_weight(pb->loss, pb->thr, r * r, pb->mu);
        if (w == 0.0) continue;
        ++cnt;
        for (int p = 0; p < np; ++p) { double a = 0; for (int i = 0; i < 9; ++i) a += G[i] * dF[p][i]; J[p] = a; }
        for (int a = 0; a < np; ++a) {
            Jtr[a] += w * r * J[a];
            for (int b = 0; b <= a; ++b) JtJ[a * np + b] += w * J[a] * J[b];
        }
    }
    return cnt;
}

static void c_step(const cproblem *pb, const cmodel *m, const double *dp, cmodel *o) {
    *o = *m;
    double dq[4];
    if (pb->kind == 5) {
        c_quat_exp(dp, dq); c_quat_mul(dq, m->qU, o->qU);
        c_quat_exp(dp + 3, dq); c_quat_mul(dq, m->qV, o->qV);
        o->sigma = m->sigma + dp[6];
    } else {
        c_quat_exp(dp, dq); c_quat_mul(m->q, dq, o->q);
        for (int i = 0; i < 3; ++i) o->t[i] = m->t[i] + m->tb[i] * dp[3] + m->tb[3 + i] * dp[4];
        if (pb->kind == 4) o->f = m->f + dp[5];
    }
}

static int c_chol_solve(const double *A, const double *b, double *x, int n) {
    double L[CNP * CNP], y[CNP];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            if (i == j) L[i * n + i] = sqrt(s); else L[i * n + j] = s / L[j * n + j];
        }
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k]; y[i] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
    return 1;
}

static void cmodel_from_blob(int kind, const orc_model *b, cmodel *m) {
    memset(m, 0, sizeof *m);
    if (kind == 5) { /* FactorizedFundamentalMatrix(F): SVD, proper rotations, sigma = s1 / s0 */
        double U[9], V[9], s[3];
        svd3((const double *)b, U, s, V);
        if (det3(U) < 0) for (int i = 0; i < 9; ++i) U[i] = -U[i];
        if (det3(V) < 0) for (int i = 0; i < 9; ++i) V[i] = -V[i];
        orc_rotmat_to_quat(U, m->qU);
        orc_rotmat_to_quat(V, m->qV);
        m->sigma = s[1] / s[0];
    } else {
        memcpy(m->q, b->q, sizeof m->q); memcpy(m->t, b->t, sizeof m->t);
        m->f = b->f1;
    }
}
static void cmodel_to_blob(int kind, const cmodel *m, orc_model *b) {
    if (kind == 5) { cmodel_F(5, m, (double *)b); return; }
    memcpy(b->q, m->q, sizeof m->q); memcpy(b->t, m->t, sizeof m->t);
    if (kind == 4) { b->f1 = m->f; b->f2 = m->f; }
}

orc_bundle_stats orc_refine_classic(int kind, const double *x1, const double *x2, int n, orc_model *blob, const orc_bundle_opt *opt,
                                    const double *weights) {
    cproblem pb = {kind, n, opt->loss_type, x1, x2, weights, opt->loss_scale, 0.5};
    const int np = kind == 3 ? 5 : (kind == 4 ? 6 : 7);
    cmodel m;
    cmodel_from_blob(kind, blob, &m);
    orc_bundle_stats stats;
    memset(&stats, 0, sizeof stats);
    stats.cost = c_cost(&pb, &m);
    stats.initial_cost = stats.cost;
    stats.grad_norm = -1; stats.step_norm = -1;
    stats.lambda = opt->initial_lambda;
    double JtJ[CNP * CNP], Jtr[CNP], sol[CNP];
    int recompute = 1;
    for (stats.iterations = 0; stats.iterations < opt->max_iterations; ++stats.iterations) {
        if (recompute) {
            c_accumulate(&pb, &m, JtJ, Jtr);
            double g = 0;
            for (int p = 0; p < np; ++p) g += Jtr[p] * Jtr[p];
            stats.grad_norm = sqrt(g);
            if (stats.grad_norm < opt->gradient_tol) break;
        }
        for (int p = 0; p < np; ++p) JtJ[p * np + p] += stats.lambda;
        c_chol_solve(JtJ, Jtr, sol, np);
        double sn = 0;
        for (int p = 0; p < np; ++p) { sol[p] = -sol[p]; sn += sol[p] * sol[p]; }
        stats.step_norm = sqrt(sn);
        if (stats.step_norm < opt->step_tol) break;
        cmodel cand;
        c_step(&pb, &m, sol, &cand);
        const double cost_new = c_cost(&pb, &cand);
        if (cost_new < stats.cost) {
            m = cand;
            stats.lambda = fmax(opt->min_lambda, stats.lambda / 10.0);
            stats.cost = cost_new;
            recompute = 1;
        } else {
            stats.invalid_steps++;
            for (int p = 0; p < np; ++p) JtJ[p * np + p] -= stats.lambda;
            stats.lambda = fmin(opt->max_lambda, stats.lambda * 10.0);
            recompute = 0;
        }
        pb.mu *= 1.5;
    }
    cmodel_to_blob(kind, &m, blob);
    return stats;
}

/* ================================================================================================ LO-RANSAC
 * ransac_relpose @0x2288f0, ransac_fundamental @0x22ac20 (+ ransac_shared_focal_relpose @0x229150 once the 6-point solver
 * exists): the same ransac<> / score_models<> loop as orc_ransac.c with the upstream estimators' samplers, scores and LO
 * steps; estimate_relative_pose @0x21f800, estimate_fundamental @0x221a00. */
#include <stdio.h>
#include <stdlib.h>
#define CMAX_MODELS 64

typedef struct {
    int kind, n, sample_sz;
    const double *x1, *x2;
    const orc_ransac_opt *opt;
    double sq_thr;
    uint64_t rng;
} cestimator;

static void draw_sample_k(uint64_t n, int k, uint64_t *state, uint64_t *out) {
    for (int i = 0; i < k; ++i) {
        int dup;
        do {
            out[i] = (uint64_t)(int64_t)orc_random_int(state) % n;
            dup = 0;
            for (int j = 0; j < i; ++j) dup |= (out[j] == out[i]);
        } while (dup);
    }
}

static void blob_F(int kind, const orc_model *m, double *F) {
    if (kind == 5) memcpy(F, m, 9 * sizeof(double));
    else orc_fundamental(m, F);
}

static int c_generate_models(cestimator *e, orc_model *out) {
    uint64_t s[8];
    draw_sample_k((uint64_t)e->n, e->sample_sz, &e->rng, s);
    double x1h[24], x2h[24];
    for (int k = 0; k < e->sample_sz; ++k) {
        const double a = e->x1[2 * s[k]], b = e->x1[2 * s[k] + 1], c = e->x2[2 * s[k]], d = e->x2[2 * s[k] + 1];
        const double n1 = sqrt(a * a + b * b + 1.0), n2 = sqrt(c * c + d * d + 1.0);
        x1h[3 * k] = a / n1; x1h[3 * k + 1] = b / n1; x1h[3 * k + 2] = 1.0 / n1;
        x2h[3 * k] = c / n2; x2h[3 * k + 1] = d / n2; x2h[3 * k + 2] = 1.0 / n2;
    }
    if (e->kind == 3) return orc_relpose_5pt(x1h, x2h, out);
    if (e->kind == 5) {
        double Fs[27];
        const int n = orc_relpose_7pt(x1h, x2h, Fs);
        for (int i = 0; i < n; ++i) { memset(&out[i], 0, sizeof out[i]); memcpy(&out[i], Fs + 9 * i, 9 * sizeof(double)); }
        return n;
    }
    if (e->kind == 4) return orc_relpose_6pt(x1h, x2h, out);
    return 0;
}

static double c_score_model(const cestimator *e, const orc_model *m, uint64_t *cnt) {
    if (e->kind == 3) return orc_msac_pose(m, e->x1, e->x2, e->n, e->sq_thr, cnt);
    double F[9];
    blob_F(e->kind, m, F);
    return orc_msac_F(F, e->x1, e->x2, e->n, e->sq_thr, cnt);
}

/* LO step.  RelativePoseEstimator / SharedFocalRelativePoseEstimator::refine_model: inliers at 5 thr^2 (get_inliers), LM over
 * that subset, TRUNCATED loss at thr, 25 iterations; FundamentalEstimator::refine_model: LM over ALL correspondences. */
int orc_classic_lo_subset = 1; /* probing switch: 1 = kinds 3/4 refine on the 5 thr^2 inlier subset */
static void c_refine_model(const cestimator *e, orc_model *m) {
    orc_bundle_opt b;
    b.max_iterations = 25; b.loss_type = 1; b.loss_scale = e->opt->max_epipolar_error;
    b.gradient_tol = 1e-10; b.step_tol = 1e-8; b.initial_lambda = 1e-3; b.min_lambda = 1e-10; b.max_lambda = 1e10;
    if (e->kind == 5 || !orc_classic_lo_subset) { orc_refine_classic(e->kind, e->x1, e->x2, e->n, m, &b, NULL); return; }
    uint8_t *mask = (uint8_t *)malloc((size_t)e->n);
    int ni;
    if (e->kind == 3) ni = orc_inliers_pose(m, e->x1, e->x2, e->n, 5.0 * e->sq_thr, mask);
    else { double F[9]; blob_F(e->kind, m, F); ni = orc_inliers_F(F, e->x1, e->x2, e->n, 5.0 * e->sq_thr, mask); }
    if (ni > e->sample_sz) {
        double *i1 = (double *)malloc(sizeof(double) * 2 * (size_t)ni), *i2 = (double *)malloc(sizeof(double) * 2 * (size_t)ni);
        int c = 0;
        for (int k = 0; k < e->n; ++k)
            if (mask[k]) { i1[2 * c] = e->x1[2 * k]; i1[2 * c + 1] = e->x1[2 * k + 1]; i2[2 * c] = e->x2[2 * k]; i2[2 * c + 1] = e->x2[2 * k + 1]; ++c; }
        orc_refine_classic(e->kind, i1, i2, ni, m, &b, NULL);
        free(i1); free(i2);
    }
    free(mask);
}

orc_ransac_stats orc_ransac_classic(int kind, const double *x1, const double *x2, int n, const orc_ransac_opt *opt, orc_model *best,
                                    uint8_t *mask) {
    orc_ransac_stats stats;
    memset(&stats, 0, sizeof stats);
    stats.model_score = DBL_MAX;
    cestimator e;
    e.kind = kind; e.n = n; e.x1 = x1; e.x2 = x2; e.opt = opt;
    e.sample_sz = kind == 3 ? 5 : (kind == 4 ? 6 : 7);
    e.sq_thr = opt->max_epipolar_error * opt->max_epipolar_error;
    e.rng = opt->seed;
    if (mask) memset(mask, 0, (size_t)(n > 0 ? n : 0));
    memset(best, 0, sizeof *best);
    if (kind == 5) { double *F = (double *)best; F[0] = F[4] = F[8] = 1.0; }
    else { best->q[0] = 1.0; best->scale = 1.0; best->f1 = best->f2 = 1.0; }
    if (n < e.sample_sz) return stats;

    uint64_t best_min_cnt = 0;
    double best_min_score = DBL_MAX;
    uint64_t dynamic_max_iter = opt->max_iterations;
    const double log_prob_missing = log(1.0 - opt->success_prob);
    orc_model models[CMAX_MODELS];
    int pending_initial = opt->score_initial_model != 0;
    for (;;) {
        int nm;
        if (!pending_initial && stats.iterations >= opt->max_iterations) break; /* max_iterations = 0: no sample is drawn (ransac<> loop head) */
        if (pending_initial) { models[0] = *best; nm = 1; }
        else nm = c_generate_models(&e, models);
        int best_ind = -1;
        for (int i = 0; i < nm; ++i) {
            uint64_t cnt;
            const double score = c_score_model(&e, &models[i], &cnt);
            const int more = cnt > best_min_cnt, better = score < best_min_score;
            if (more || better) {
                if (more) best_min_cnt = cnt;
                if (better) best_min_score = score;
                best_ind = i;
                if (score < stats.model_score) { stats.model_score = score; *best = models[i]; stats.num_inliers = cnt; }
            }
        }
        if (best_ind >= 0) {
            orc_model refined = models[best_ind];
            c_refine_model(&e, &refined);
            stats.refinements++;
            uint64_t cnt;
            const double score = c_score_model(&e, &refined, &cnt);
            if (getenv("ORC_TRACE_LO")) fprintf(stderr, "[orc] it %llu LO of model %d: score %.17g cnt %llu |t| %.12f -> %s\n", (unsigned long long)stats.iterations, best_ind, score, (unsigned long long)cnt, sqrt(refined.t[0]*refined.t[0]+refined.t[1]*refined.t[1]+refined.t[2]*refined.t[2]), score < stats.model_score ? "adopted" : "kept");
            if (score < stats.model_score) { stats.model_score = score; stats.num_inliers = cnt; *best = refined; }
            stats.inlier_ratio = (double)stats.num_inliers / (double)n;
            if (stats.inlier_ratio >= 0.9999) dynamic_max_iter = opt->min_iterations;
            else if (stats.inlier_ratio <= 0.0001) dynamic_max_iter = opt->max_iterations;
            else {
                const double prob_outlier = 1.0 - pow(stats.inlier_ratio, (double)e.sample_sz);
                dynamic_max_iter = orc_f64_to_u64(ceil(log_prob_missing / log(prob_outlier) * opt->dyn_num_trials_mult));
            }
        }
        if (pending_initial) { pending_initial = 0; continue; }
        ++stats.iterations;
        if (stats.iterations >= opt->max_iterations) break;
        if (stats.iterations <= opt->min_iterations) continue;
        if (stats.iterations > dynamic_max_iter) break;
    }
    {
        orc_model refined = *best;
        c_refine_model(&e, &refined);
        stats.refinements++;
        uint64_t cnt;
        const double score = c_score_model(&e, &refined, &cnt);
        if (getenv("ORC_TRACE_LO")) fprintf(stderr, "[orc] final LO: score %.17g (best %.17g) cnt %llu |t| %.12f\n", score, stats.model_score, (unsigned long long)cnt, sqrt(refined.t[0]*refined.t[0]+refined.t[1]*refined.t[1]+refined.t[2]*refined.t[2]));
        if (score < stats.model_score) { *best = refined; stats.num_inliers = cnt; }
    }
    if (mask) {
        if (kind == 3) orc_inliers_pose(best, x1, x2, n, e.sq_thr, mask);
        else { double F[9]; blob_F(kind, best, F); orc_inliers_F(F, x1, x2, n, e.sq_thr, mask); }
    }
    return stats;
}

static double c_cam_focal(const double *cam) { return ((int)cam[0] == 1) ? 0.5 * (cam[2] + cam[3]) : cam[2]; }
static void c_cam_unproject(const double *cam, const double *x, double *o) {
    if ((int)cam[0] == 1) { o[0] = (x[0] - cam[4]) / cam[2]; o[1] = (x[1] - cam[5]) / cam[3]; }
    else { o[0] = (x[0] - cam[3]) / cam[2]; o[1] = (x[1] - cam[4]) / cam[2]; }
}

/* cam: {model_id, nparams, params...} for kind 3; pp[2] for kind 4; nothing for kind 5 */
orc_ransac_stats orc_estimate_classic(int kind, const double *x1, const double *x2, int n, const double *cam1, const double *cam2,
                                      const double *pp, const orc_ransac_opt *ropt, const orc_bundle_opt *bopt, orc_model *best,
                                      uint8_t *mask) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    double *a1 = (double *)malloc(sizeof(double) * 2 * nn), *a2 = (double *)malloc(sizeof(double) * 2 * nn);
    orc_ransac_opt ro = *ropt;
    orc_bundle_opt bo = *bopt;
    double scale = 1.0, c1[2] = {0, 0}, c2[2] = {0, 0};
    const int min_n = kind == 3 ? 5 : (kind == 4 ? 6 : 7);
    if (kind == 3) {
        for (int k = 0; k < n; ++k) { c_cam_unproject(cam1, x1 + 2 * k, a1 + 2 * k); c_cam_unproject(cam2, x2 + 2 * k, a2 + 2 * k); }
        const double k = 0.5 * (1.0 / c_cam_focal(cam1) + 1.0 / c_cam_focal(cam2));
        ro.max_epipolar_error *= k; bo.loss_scale *= k;
    } else {
        /* kind 5: normalize_points(normalize_scale, normalize_centroid, shared_scale) @0x4f6ae0; kind 4: shift by pp, shared scale */
        if (kind == 5) {
            for (int k = 0; k < n; ++k) { c1[0] += x1[2 * k]; c1[1] += x1[2 * k + 1]; c2[0] += x2[2 * k]; c2[1] += x2[2 * k + 1]; }
            if (n > 0) { c1[0] /= n; c1[1] /= n; c2[0] /= n; c2[1] /= n; }
        } else { c1[0] = c2[0] = pp[0]; c1[1] = c2[1] = pp[1]; }
        double acc = 0.0;
        for (int k = 0; k < n; ++k) {
            a1[2 * k] = x1[2 * k] - c1[0]; a1[2 * k + 1] = x1[2 * k + 1] - c1[1];
            a2[2 * k] = x2[2 * k] - c2[0]; a2[2 * k + 1] = x2[2 * k + 1] - c2[1];
            acc += sqrt(a1[2 * k] * a1[2 * k] + a1[2 * k + 1] * a1[2 * k + 1]) + sqrt(a2[2 * k] * a2[2 * k] + a2[2 * k + 1] * a2[2 * k + 1]);
        }
        scale = acc / (sqrt(2.0) * (double)(n > 0 ? n : 1));
        for (int k = 0; k < 2 * n; ++k) { a1[k] /= scale; a2[k] /= scale; }
        ro.max_epipolar_error /= scale; bo.loss_scale /= scale;
    }
    uint8_t *m8 = mask ? mask : (uint8_t *)malloc(nn);
    orc_ransac_stats stats = orc_ransac_classic(kind, a1, a2, n, &ro, best, m8);
    if (stats.num_inliers > (uint64_t)min_n) {
        int ni = 0;
        double *i1 = (double *)malloc(sizeof(double) * 2 * nn), *i2 = (double *)malloc(sizeof(double) * 2 * nn);
        for (int k = 0; k < n; ++k)
            if (m8[k]) { i1[2 * ni] = a1[2 * k]; i1[2 * ni + 1] = a1[2 * k + 1]; i2[2 * ni] = a2[2 * k]; i2[2 * ni + 1] = a2[2 * k + 1]; ++ni; }
        orc_refine_classic(kind, i1, i2, ni, best, &bo, NULL);
        free(i1); free(i2);
    }
    if (kind == 4) { best->f1 *= scale; best->f2 *= scale; }
    if (kind == 5) { /* F <- T2' F T1, T = [1/s 0 -c.x/s; 0 1/s -c.y/s; 0 0 1], then unit Frobenius norm */
        double *F = (double *)best, T1[9] = {1 / scale, 0, -c1[0] / scale, 0, 1 / scale, -c1[1] / scale, 0, 0, 1},
               T2[9] = {1 / scale, 0, -c2[0] / scale, 0, 1 / scale, -c2[1] / scale, 0, 0, 1}, M[9], O[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += F[3 * i + k] * T1[3 * k + j]; M[3 * i + j] = s; }
        double nrm = 0;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += T2[3 * k + i] * M[3 * k + j]; O[3 * i + j] = s; nrm += s * s; }
        nrm = sqrt(nrm);
        for (int i = 0; i < 9; ++i) F[i] = O[i] / nrm;
    }
    if (!mask) free(m8);
    free(a1); free(a2);
    return stats;
}
