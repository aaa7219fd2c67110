/* orc_solvers.c — minimal solvers of the monodepth estimators.  TEST INFRASTRUCTURE (see mdrp_oracle.h).
 *
 * The reference implementations exist only as machine code (p3p @0xecd50, relpose_monodepth_3pt @0x155ca0,
 * ..._shared_focal @0x18fdf0, ..._varying_focal @0x19bcd0; SURVEY.md §8a-4..6').  What is restated here is the
 * published problem each one solves (wheel METADATA:259-310; Ding et al. CVPR23 / ICCV25), derived from the
 * constraint equations; the SOLUTION SETS are pinned against the binary (tests/golden/solvers_*.npz).
 */
#include "mdrp_oracle.h"
#include <math.h>
#include <string.h>

/* ---------- small algebra ---------- */
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static void sub3(const double *a, const double *b, double *c) { c[0] = a[0] - b[0]; c[1] = a[1] - b[1]; c[2] = a[2] - b[2]; }

/* real roots of x^3 + b x^2 + c x + d, Newton-polished; returns count (1 or 3) */
static int solve_cubic_real(double b, double c, double d, double r[3]) {
    const double p = c - b * b / 3.0;
    const double q = 2.0 * b * b * b / 27.0 - b * c / 3.0 + d;
    const double disc = q * q / 4.0 + p * p * p / 27.0;
    int n;
    if (disc > 0) {
        const double sq = sqrt(disc);
        const double u = cbrt(-q / 2.0 + sq), v = cbrt(-q / 2.0 - sq);
        r[0] = u + v - b / 3.0;
        n = 1;
    } else {
        const double rr = sqrt(-p / 3.0);
        double arg = (rr > 0) ? (-q / 2.0) / (rr * rr * rr) : 0.0;
        if (arg > 1) arg = 1;
        if (arg < -1) arg = -1;
        const double phi = acos(arg);
        for (int k = 0; k < 3; ++k) r[k] = 2.0 * rr * cos((phi - 2.0 * M_PI * k) / 3.0) - b / 3.0;
        n = 3;
    }
    for (int k = 0; k < n; ++k) {
        double x = r[k];
        for (int it = 0; it < 3; ++it) {
            const double f = ((x + b) * x + c) * x + d;
            const double fp = (3.0 * x + 2.0 * b) * x + c;
            if (fp == 0.0) break;
            x -= f / fp;
        }
        r[k] = x;
    }
    return n;
}

/* rigid alignment of two congruent 3-point sets: R (X_i - X_j) = Y_i - Y_j, t = Y_0 - R X_0 */
static void align3(const double X[9], const double Y[9], double R[9], double t[3]) {
    double a[3], b[3], c[3], u[3], v[3], w[3];
    sub3(X + 3, X, a); sub3(X + 6, X, b); cross3(a, b, c);
    sub3(Y + 3, Y, u); sub3(Y + 6, Y, v); cross3(u, v, w);
    /* inverse of [a b c] (columns): rows are (b x c, c x a, a x b) / det */
    double bc[3], ca[3], ab[3];
    cross3(b, c, bc); cross3(c, a, ca); cross3(a, b, ab);
    const double det = dot3(a, bc);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = (u[i] * bc[j] + v[i] * ca[j] + w[i] * ab[j]) / det;
    for (int i = 0; i < 3; ++i) t[i] = Y[i] - (R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2]);
}

static void model_init(orc_model *m) {
    memset(m, 0, sizeof *m);
    m->q[0] = 1.0; m->scale = 1.0; m->f1 = 1.0; m->f2 = 1.0;
}

/* ---------- P3P (a-6'): depths l_i>0 with |l_i x_i - l_j x_j|^2 = |X_i - X_j|^2 ----------
 * Two homogeneous conics in (l0:l1:l2) -> degenerate member of their pencil (cubic) -> line pair -> <=4 points. */
static void sym_adj(const double C[6] /*00 01 02 11 12 22*/, double A[6]) {
    A[0] = C[3] * C[5] - C[4] * C[4];
    A[1] = C[2] * C[4] - C[1] * C[5];
    A[2] = C[1] * C[4] - C[2] * C[3];
    A[3] = C[0] * C[5] - C[2] * C[2];
    A[4] = C[1] * C[2] - C[0] * C[4];
    A[5] = C[0] * C[3] - C[1] * C[1];
}
static double sym_det(const double C[6]) {
    return C[0] * (C[3] * C[5] - C[4] * C[4]) - C[1] * (C[1] * C[5] - C[4] * C[2]) + C[2] * (C[1] * C[4] - C[3] * C[2]);
}
static double sym_tr_prod(const double A[6], const double B[6]) { /* trace(A B) for symmetric */
    return A[0] * B[0] + A[3] * B[3] + A[5] * B[5] + 2.0 * (A[1] * B[1] + A[2] * B[2] + A[4] * B[4]);
}
static double sym_quad(const double C[6], const double *u, const double *v) { /* u' C v */
    return u[0] * (C[0] * v[0] + C[1] * v[1] + C[2] * v[2]) + u[1] * (C[1] * v[0] + C[3] * v[1] + C[4] * v[2]) +
           u[2] * (C[2] * v[0] + C[4] * v[1] + C[5] * v[2]);
}

static int p3p_depths(const double m01, const double m02, const double m12, const double a01, const double a02,
                      const double a12, double L[4][3]) {
    /* D1 = a12 Q01 - a01 Q12,  D2 = a12 Q02 - a02 Q12 */
    const double D1[6] = {a12, -a12 * m01, 0.0, a12 - a01, a01 * m12, -a01};
    const double D2[6] = {a12, 0.0, -a12 * m02, -a02, a02 * m12, a12 - a02};
    double A1[6], A2[6];
    sym_adj(D1, A1); sym_adj(D2, A2);
    const double c3 = sym_det(D2), c2 = sym_tr_prod(A2, D1), c1 = sym_tr_prod(A1, D2), c0 = sym_det(D1);
    double roots[3];
    int nr;
    if (fabs(c3) > 1e-300) nr = solve_cubic_real(c2 / c3, c1 / c3, c0 / c3, roots);
    else return 0;
    /* pick the root whose degenerate conic is the most clearly a REAL line pair */
    double best = -1.0, C[6], B[6];
    int found = 0;
    double g = 0;
    for (int k = 0; k < nr; ++k) {
        double Ck[6], Ak[6], nrm = 0;
        for (int i = 0; i < 6; ++i) { Ck[i] = D1[i] + roots[k] * D2[i]; nrm += Ck[i] * Ck[i]; }
        sym_adj(Ck, Ak);
        double mx = -Ak[0];
        if (-Ak[3] > mx) mx = -Ak[3];
        if (-Ak[5] > mx) mx = -Ak[5];
        mx /= nrm;
        if (mx > best) { best = mx; memcpy(C, Ck, sizeof C); for (int i = 0; i < 6; ++i) B[i] = -Ak[i]; g = roots[k]; found = 1; }
    }
    if (!found || best <= 0) return 0;
    /* p = l x m from the largest diagonal of B = -adj(C) = p p' */
    double p[3];
    if (B[0] >= B[3] && B[0] >= B[5]) { const double s = sqrt(B[0]); p[0] = s; p[1] = B[1] / s; p[2] = B[2] / s; }
    else if (B[3] >= B[5]) { const double s = sqrt(B[3]); p[0] = B[1] / s; p[1] = s; p[2] = B[4] / s; }
    else { const double s = sqrt(B[5]); p[0] = B[2] / s; p[1] = B[4] / s; p[2] = s; }
    /* M = C + [p]x = 2 m l' : rows ~ l, columns ~ m */
    const double M[9] = {C[0], C[1] - p[2], C[2] + p[1], C[1] + p[2], C[3], C[4] - p[0], C[2] - p[1], C[4] + p[0], C[5]};
    double lines[2][3];
    {
        int br = 0, bc = 0;
        double nr2 = -1, nc2 = -1;
        for (int i = 0; i < 3; ++i) {
            const double r2 = M[3 * i] * M[3 * i] + M[3 * i + 1] * M[3 * i + 1] + M[3 * i + 2] * M[3 * i + 2];
            const double c2_ = M[i] * M[i] + M[3 + i] * M[3 + i] + M[6 + i] * M[6 + i];
            if (r2 > nr2) { nr2 = r2; br = i; }
            if (c2_ > nc2) { nc2 = c2_; bc = i; }
        }
        for (int j = 0; j < 3; ++j) { lines[0][j] = M[3 * br + j]; lines[1][j] = M[3 * j + bc]; }
    }
    const double *Dq = (fabs(g) < 1.0) ? D2 : D1; /* the pencil member that does NOT vanish on the lines */
    int n = 0;
    for (int li = 0; li < 2; ++li) {
        const double *l = lines[li];
        /* two points spanning the line: u = l x e_a, v = l x e_b with a,b the two smallest |l| */
        int k = 0;
        if (fabs(l[1]) > fabs(l[k])) k = 1;
        if (fabs(l[2]) > fabs(l[k])) k = 2;
        double ea[3] = {0, 0, 0}, eb[3] = {0, 0, 0}, u[3], v[3];
        ea[(k + 1) % 3] = 1.0; eb[(k + 2) % 3] = 1.0;
        cross3(l, ea, u); cross3(l, eb, v);
        const double qa = sym_quad(Dq, v, v), qb = sym_quad(Dq, u, v), qc = sym_quad(Dq, u, u);
        const double disc = qb * qb - qa * qc;
        if (disc < 0) continue;
        const double sq = sqrt(disc);
        const double qq = -(qb + (qb >= 0 ? sq : -sq));
        /* (tau:sigma) = (qq:qa) and (qc:qq);  lambda = sigma u + tau v */
        const double ts[2][2] = {{qq, qa}, {qc, qq}};
        for (int s = 0; s < 2; ++s) {
            double lam[3];
            for (int i = 0; i < 3; ++i) lam[i] = ts[s][1] * u[i] + ts[s][0] * v[i];
            /* fix the scale with the largest of the three distance equations */
            double qv, av;
            if (a12 >= a01 && a12 >= a02) { qv = lam[1] * lam[1] + lam[2] * lam[2] - 2 * m12 * lam[1] * lam[2]; av = a12; }
            else if (a02 >= a01) { qv = lam[0] * lam[0] + lam[2] * lam[2] - 2 * m02 * lam[0] * lam[2]; av = a02; }
            else { qv = lam[0] * lam[0] + lam[1] * lam[1] - 2 * m01 * lam[0] * lam[1]; av = a01; }
            if (!(qv > 0)) continue;
            double sc = sqrt(av / qv);
            if (lam[0] < 0) sc = -sc;
            for (int i = 0; i < 3; ++i) lam[i] *= sc;
            if (!(lam[0] > 0 && lam[1] > 0 && lam[2] > 0)) continue;
            /* Gauss-Newton polish on the three distance equations */
            for (int it = 0; it < 5; ++it) {
                const double r0 = lam[0] * lam[0] + lam[1] * lam[1] - 2 * m01 * lam[0] * lam[1] - a01;
                const double r1 = lam[0] * lam[0] + lam[2] * lam[2] - 2 * m02 * lam[0] * lam[2] - a02;
                const double r2 = lam[1] * lam[1] + lam[2] * lam[2] - 2 * m12 * lam[1] * lam[2] - a12;
                if (fabs(r0) + fabs(r1) + fabs(r2) < 1e-15 * (a01 + a02 + a12)) break;
                const double J[9] = {2 * (lam[0] - m01 * lam[1]), 2 * (lam[1] - m01 * lam[0]), 0,
                                     2 * (lam[0] - m02 * lam[2]), 0, 2 * (lam[2] - m02 * lam[0]),
                                     0, 2 * (lam[1] - m12 * lam[2]), 2 * (lam[2] - m12 * lam[1])};
                const double det = J[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (J[3] * J[8] - J[5] * J[6]) +
                                   J[2] * (J[3] * J[7] - J[4] * J[6]);
                if (fabs(det) < 1e-300) break;
                const double i0 = (r0 * (J[4] * J[8] - J[5] * J[7]) - J[1] * (r1 * J[8] - J[5] * r2) + J[2] * (r1 * J[7] - J[4] * r2)) / det;
                const double i1 = (J[0] * (r1 * J[8] - J[5] * r2) - r0 * (J[3] * J[8] - J[5] * J[6]) + J[2] * (J[3] * r2 - r1 * J[6])) / det;
                const double i2 = (J[0] * (J[4] * r2 - r1 * J[7]) - J[1] * (J[3] * r2 - r1 * J[6]) + r0 * (J[3] * J[7] - J[4] * J[6])) / det;
                lam[0] -= i0; lam[1] -= i1; lam[2] -= i2;
            }
            if (!(lam[0] > 0 && lam[1] > 0 && lam[2] > 0)) continue;
            memcpy(L[n], lam, sizeof lam);
            if (++n == 4) return n;
        }
    }
    return n;
}

int orc_p3p(const double x[9], const double X[9], orc_model out[4]) {
    double d01[3], d02[3], d12[3];
    sub3(X, X + 3, d01); sub3(X, X + 6, d02); sub3(X + 3, X + 6, d12);
    const double a01 = dot3(d01, d01), a02 = dot3(d02, d02), a12 = dot3(d12, d12);
    const double m01 = dot3(x, x + 3), m02 = dot3(x, x + 6), m12 = dot3(x + 3, x + 6);
    double L[4][3];
    const int n = p3p_depths(m01, m02, m12, a01, a02, a12, L);
    for (int s = 0; s < n; ++s) {
        double Y[9], R[9];
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) Y[3 * i + k] = L[s][i] * x[3 * i + k];
        model_init(&out[s]);
        align3(X, Y, R, out[s].t);
        orc_rotmat_to_quat(R, out[s].q);
    }
    return n;
}

/* a-6': default calibrated hypothesis generator (monodepth_estimate_shift = false) */
int orc_solver_calib_p3p(const double x1h[9], const double x2h[9], const double d1[3], const double d2[3], orc_model out[4]) {
    double X[9], xb[9];
    for (int i = 0; i < 3; ++i) {
        const double nrm = sqrt(dot3(x2h + 3 * i, x2h + 3 * i));
        for (int k = 0; k < 3; ++k) { X[3 * i + k] = d1[i] * x1h[3 * i + k]; xb[3 * i + k] = x2h[3 * i + k] / nrm; }
    }
    const int n = orc_p3p(xb, X, out);
    for (int s = 0; s < n; ++s) {
        double R[9];
        orc_quat_to_rotmat(out[s].q, R);
        const double px = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + out[s].t[0];
        out[s].scale = px / (d2[0] * x2h[0]);
    }
    return n;
}

/* When does the reference's P3P hand back NaN poses?  Its p3p() (Ding et al., "Revisiting the P3P problem": one real root s of a cubic, the degenerate
 * conic C(s) of the pencil split into two lines p, q) takes sqrt of the largest diagonal entry of -adj(C) without testing its sign.  Where C(s) is a
 * POINT conic (complex line pair) all three are negative: the root is NaN, both "lines" are NaN, every test that would discard a solution is a
 * comparison with NaN (false), and 2 x 2 NaN poses come out.  Black-box on 12 000 samples of noisy pairs: the binary returns 4 NaN poses on exactly the
 * 398 samples where this predicate holds (on all of them our solver, rightly, finds no real pose) and never otherwise.  A NaN model scores N * thr
 * with 0 inliers — a record while no model has been scored yet — and costs the reference one LO (DESIGN.md §5 class i): generate_models() below
 * reproduces that.  x: unit bearings in image 2, X: points of image 1 (d1 * x1h). */
int orc_p3p_reference_nan(const double x[9], const double X[9]) {
    double xs[3][3], a01 = 0, a02 = 0, a12 = 0;
    for (int k = 0; k < 3; ++k) {
        for (int i = 0; i < 3; ++i) xs[i][k] = x[3 * i + k];
        a01 += (X[k] - X[3 + k]) * (X[k] - X[3 + k]); a02 += (X[k] - X[6 + k]) * (X[k] - X[6 + k]); a12 += (X[3 + k] - X[6 + k]) * (X[3 + k] - X[6 + k]);
    }
    /* the largest of the three distances becomes "12" */
    if (a01 > a02) {
        if (a01 > a12) { for (int k = 0; k < 3; ++k) { const double t = xs[0][k]; xs[0][k] = xs[2][k]; xs[2][k] = t; } const double t = a01; a01 = a12; a12 = t; }
    } else if (a02 > a12) { for (int k = 0; k < 3; ++k) { const double t = xs[0][k]; xs[0][k] = xs[1][k]; xs[1][k] = t; } const double t = a02; a02 = a12; a12 = t; }
    const double a12d = 1.0 / a12, a = a01 * a12d, b = a02 * a12d;
    const double m01 = dot3(xs[0], xs[1]), m02 = dot3(xs[0], xs[2]), m12 = dot3(xs[1], xs[2]);
    const double m12sq = -m12 * m12 + 1.0, m02sq = -1.0 + m02 * m02, m01sq = -1.0 + m01 * m01;
    const double ab = a * b, bsq = b * b, asq = a * a, m013 = -2.0 + 2.0 * m01 * m02 * m12;
    const double bsqm12sq = bsq * m12sq, asqm12sq = asq * m12sq, abm12sq = 2.0 * ab * m12sq;
    const double k3i = 1.0 / (bsqm12sq + b * m02sq);
    const double k2 = k3i * ((-1.0 + a) * m02sq + abm12sq + bsqm12sq + b * m013);
    const double k1 = k3i * (asqm12sq + abm12sq + a * m013 + (-1.0 + b) * m01sq);
    const double k0 = k3i * (asqm12sq + a * m01sq);
    /* solve_cubic_single_real: the real root if there is one, the largest of three otherwise */
    double s;
    {
        const double ca = k1 - k2 * k2 / 3.0;
        double cb = (2.0 * k2 * k2 * k2 - 9.0 * k2 * k1) / 27.0 + k0;
        double cc = cb * cb / 4.0 + ca * ca * ca / 27.0;
        if (cc != 0) {
            if (cc > 0) { cc = sqrt(cc); cb *= -0.5; s = cbrt(cb + cc) + cbrt(cb - cc) - k2 / 3.0; }
            else { cc = 3.0 * cb / (2.0 * ca) * sqrt(-3.0 / ca); s = 2.0 * sqrt(-ca / 3.0) * cos(acos(cc) / 3.0) - k2 / 3.0; }
        } else s = -k2 / 3.0 + (ca != 0 ? (3.0 * cb / ca) : 0);
    }
    const double C00 = -a + s * (1 - b), C01 = -m02 * s, C02 = a * m12 + b * m12 * s, C11 = s + 1, C12 = -m01, C22 = -a - b * s + 1;
    const double A0 = C12 * C12 - C11 * C22, A1 = C02 * C02 - C00 * C22, A2 = C01 * C01 - C00 * C11;
    const double mx = A0 > A1 ? (A0 > A2 ? A0 : A2) : (A1 > A2 ? A1 : A2);
    return mx < 0;
}

/* ---------- univariate helpers ---------- */
/* real roots of x^4 + b x^3 + c x^2 + d x + e (Ferrari via the resolvent cubic), Newton-polished */
static int solve_quartic_real(double b, double c, double d, double e, double roots[4]) {
    const double b2 = b * b;
    const double p = c - 3.0 * b2 / 8.0;
    const double q = d - b * c / 2.0 + b2 * b / 8.0;
    const double r = e - b * d / 4.0 + b2 * c / 16.0 - 3.0 * b2 * b2 / 256.0;
    double z3[3];
    const int nz = solve_cubic_real(2.0 * p, p * p - 4.0 * r, -q * q, z3);
    double z = z3[0];
    for (int k = 1; k < nz; ++k)
        if (z3[k] > z) z = z3[k];
    int n = 0;
    double y[4];
    const double scale = fabs(p) + sqrt(fabs(r)) + 1e-300;
    if (z <= 1e-14 * scale) { /* biquadratic: y^4 + p y^2 + r */
        const double disc = p * p - 4.0 * r;
        if (disc < 0) return 0;
        const double sq = sqrt(disc);
        const double y2a = (-p + sq) / 2.0, y2b = (-p - sq) / 2.0;
        if (y2a >= 0) { y[n++] = sqrt(y2a); y[n++] = -sqrt(y2a); }
        if (y2b >= 0) { y[n++] = sqrt(y2b); y[n++] = -sqrt(y2b); }
    } else {
        const double s = sqrt(z);
        const double t1 = (p + z - q / s) / 2.0, t2 = (p + z + q / s) / 2.0;
        const double disc1 = z - 4.0 * t1, disc2 = z - 4.0 * t2;
        if (disc1 >= 0) { const double sq = sqrt(disc1); y[n++] = (-s + sq) / 2.0; y[n++] = (-s - sq) / 2.0; }
        if (disc2 >= 0) { const double sq = sqrt(disc2); y[n++] = (s + sq) / 2.0; y[n++] = (s - sq) / 2.0; }
    }
    for (int k = 0; k < n; ++k) {
        double x = y[k] - b / 4.0;
        for (int it = 0; it < 3; ++it) {
            const double f = (((x + b) * x + c) * x + d) * x + e;
            const double fp = ((4.0 * x + 3.0 * b) * x + 2.0 * c) * x + d;
            if (fp == 0.0) break;
            x -= f / fp;
        }
        roots[k] = x;
    }
    return n;
}

static int solve3x3(const double A[9], const double b[3], double x[3]) {
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    if (!(fabs(det) > 0)) return 0;
    const double id = 1.0 / det;
    x[0] = (c00 * b[0] + (A[2] * A[7] - A[1] * A[8]) * b[1] + (A[1] * A[5] - A[2] * A[4]) * b[2]) * id;
    x[1] = (c01 * b[0] + (A[0] * A[8] - A[2] * A[6]) * b[1] + (A[2] * A[3] - A[0] * A[5]) * b[2]) * id;
    x[2] = (c02 * b[0] + (A[1] * A[6] - A[0] * A[7]) * b[1] + (A[0] * A[4] - A[1] * A[3]) * b[2]) * id;
    return 1;
}

static const int PAIR_I[3] = {0, 0, 1}, PAIR_J[3] = {1, 2, 2};

/* ---------- a-4: calibrated 3pt with scale + two shifts ----------
 * |(d1_i+u) x1_i - (d1_j+u) x1_j|^2 = s^2 |(d2_i+v) x2_i - (d2_j+v) x2_j|^2  for the 3 point pairs.
 * With (a,b,c) = (s^2, s^2 v, s^2 v^2) the equations are linear in (a,b,c) and quadratic in u;
 * a c = b^2 leaves a quartic in u. */
int orc_solver_calib_shift(const double x1h[9], const double x2h[9], const double d1[3], const double d2[3], orc_model out[4]) {
    double A1[3], B1[3], C1[3], M[9];
    for (int k = 0; k < 3; ++k) {
        const int i = PAIR_I[k], j = PAIR_J[k];
        double p[3], q[3];
        for (int c = 0; c < 3; ++c) { p[c] = d1[i] * x1h[3 * i + c] - d1[j] * x1h[3 * j + c]; q[c] = x1h[3 * i + c] - x1h[3 * j + c]; }
        A1[k] = dot3(p, p); B1[k] = dot3(p, q); C1[k] = dot3(q, q);
        for (int c = 0; c < 3; ++c) { p[c] = d2[i] * x2h[3 * i + c] - d2[j] * x2h[3 * j + c]; q[c] = x2h[3 * i + c] - x2h[3 * j + c]; }
        M[3 * k] = dot3(p, p); M[3 * k + 1] = 2.0 * dot3(p, q); M[3 * k + 2] = dot3(q, q);
    }
    double g0[3], g1[3], g2[3], B2[3] = {2 * B1[0], 2 * B1[1], 2 * B1[2]};
    if (!solve3x3(M, A1, g0) || !solve3x3(M, B2, g1) || !solve3x3(M, C1, g2)) return 0;
    /* a(u) = g0[0] + g1[0] u + g2[0] u^2, b(u) = g*[1], c(u) = g*[2] */
    const double k4 = g2[0] * g2[2] - g2[1] * g2[1];
    const double k3 = g1[0] * g2[2] + g2[0] * g1[2] - 2.0 * g1[1] * g2[1];
    const double k2 = g0[0] * g2[2] + g1[0] * g1[2] + g2[0] * g0[2] - g1[1] * g1[1] - 2.0 * g0[1] * g2[1];
    const double k1 = g0[0] * g1[2] + g1[0] * g0[2] - 2.0 * g0[1] * g1[1];
    const double k0 = g0[0] * g0[2] - g0[1] * g0[1];
    if (!(fabs(k4) > 0)) return 0;
    double us[4];
    const int nu = solve_quartic_real(k3 / k4, k2 / k4, k1 / k4, k0 / k4, us);
    int n = 0;
    for (int r = 0; r < nu; ++r) {
        double u = us[r];
        const double a = g0[0] + u * (g1[0] + u * g2[0]);
        const double b = g0[1] + u * (g1[1] + u * g2[1]);
        if (!(a > 0)) continue;
        double s = sqrt(a), v = b / a;
        /* Newton polish of (s,u,v) on the three distance equations (refine_suv @0x15de40) */
        for (int it = 0; it < 5; ++it) {
            double J[9], res[3], dx[3];
            for (int k = 0; k < 3; ++k) {
                const double lhs = A1[k] + u * (2.0 * B1[k] + u * C1[k]);
                const double rhs = M[3 * k] + v * (M[3 * k + 1] + v * M[3 * k + 2]);
                res[k] = lhs - s * s * rhs;
                J[3 * k] = -2.0 * s * rhs;
                J[3 * k + 1] = 2.0 * B1[k] + 2.0 * u * C1[k];
                J[3 * k + 2] = -s * s * (M[3 * k + 1] + 2.0 * v * M[3 * k + 2]);
            }
            if (!solve3x3(J, res, dx)) break;
            s -= dx[0]; u -= dx[1]; v -= dx[2];
            if (fabs(dx[0]) + fabs(dx[1]) + fabs(dx[2]) < 1e-15 * (fabs(s) + fabs(u) + fabs(v))) break;
        }
        if (!(s > 0)) continue;
        /* the reference keeps only solutions whose shifted depths are all positive (black-box: 0 exceptions / 2839) */
        int pos = 1;
        for (int i = 0; i < 3; ++i) pos &= (d1[i] + u > 0) && (d2[i] + v > 0);
        if (!pos) continue;
        double X[9], Y[9], R[9];
        for (int i = 0; i < 3; ++i)
            for (int c = 0; c < 3; ++c) { X[3 * i + c] = (d1[i] + u) * x1h[3 * i + c]; Y[3 * i + c] = s * (d2[i] + v) * x2h[3 * i + c]; }
        model_init(&out[n]);
        align3(X, Y, R, out[n].t);
        orc_rotmat_to_quat(R, out[n].q);
        out[n].scale = s; out[n].shift1 = u; out[n].shift2 = v;
        if (++n == 4) break;
    }
    return n;
}

/* ---------- a-6: varying focal, linear in (1/f1^2, s^2/f2^2, s^2) ---------- */
int orc_solver_varying(const double x1h[9], const double x2h[9], const double d1[3], const double d2[3], orc_model out[4]) {
    double A[9], rhs[3], sol[3];
    for (int k = 0; k < 3; ++k) {
        const int i = PAIR_I[k], j = PAIR_J[k];
        const double ax = d1[i] * x1h[3 * i] - d1[j] * x1h[3 * j], ay = d1[i] * x1h[3 * i + 1] - d1[j] * x1h[3 * j + 1];
        const double bx = d2[i] * x2h[3 * i] - d2[j] * x2h[3 * j], by = d2[i] * x2h[3 * i + 1] - d2[j] * x2h[3 * j + 1];
        const double dz1 = d1[i] - d1[j], dz2 = d2[i] - d2[j];
        A[3 * k] = ax * ax + ay * ay;
        A[3 * k + 1] = -(bx * bx + by * by);
        A[3 * k + 2] = -dz2 * dz2;
        rhs[k] = -dz1 * dz1;
    }
    if (!solve3x3(A, rhs, sol)) return 0;
    if (!(sol[0] > 0 && sol[1] > 0 && sol[2] > 0)) return 0;
    const double f1 = 1.0 / sqrt(sol[0]), s = sqrt(sol[2]), f2 = sqrt(sol[2] / sol[1]);
    double X[9], Y[9], R[9];
    for (int i = 0; i < 3; ++i) {
        X[3 * i] = d1[i] * x1h[3 * i] / f1; X[3 * i + 1] = d1[i] * x1h[3 * i + 1] / f1; X[3 * i + 2] = d1[i];
        Y[3 * i] = s * d2[i] * x2h[3 * i] / f2; Y[3 * i + 1] = s * d2[i] * x2h[3 * i + 1] / f2; Y[3 * i + 2] = s * d2[i];
    }
    model_init(&out[0]);
    align3(X, Y, R, out[0].t);
    orc_rotmat_to_quat(R, out[0].q);
    out[0].scale = s; out[0].f1 = f1; out[0].f2 = f2;
    return 1;
}

/* ---------- a-5: shared focal ----------
 * Unknowns w = 1/f^2, sigma = s^2 and rho = (depth of point 2 in image 2)/s.  Points 0 and 1 use both depths,
 * point 2 uses d1[2] and only the bearing in image 2 (d2[2] is NOT read — black-box property of the reference,
 * SURVEY.md §8a-5).  The three distance equations
 *     N(w)           = sigma D(w)                                   (pair 0-1)
 *     L02(w)         = sigma (a0(w) - 2 rho c0(w) + rho^2 e(w))     (pair 0-2)
 *     L12(w)         = sigma (a1(w) - 2 rho c1(w) + rho^2 e(w))     (pair 1-2)
 * reduce (sigma from the first, rho from the difference of the other two) to a quintic in w whose constant term
 * vanishes identically (w = 0 is f = infinity) -> quartic, <= 4 solutions.  Kept: w > 0, sigma > 0, rho > 0. */
#define PMAX 8
typedef struct { double c[PMAX]; int n; } poly_t; /* c[0] + c[1] w + ... ; n = number of coefficients */
static poly_t pl(double c0, double c1) { poly_t p; memset(&p, 0, sizeof p); p.c[0] = c0; p.c[1] = c1; p.n = 2; return p; }
static poly_t pmul(poly_t a, poly_t b) {
    poly_t r; memset(&r, 0, sizeof r); r.n = a.n + b.n - 1;
    for (int i = 0; i < a.n; ++i) for (int j = 0; j < b.n; ++j) r.c[i + j] += a.c[i] * b.c[j];
    return r;
}
static poly_t padd(poly_t a, poly_t b, double sb) {
    poly_t r; memset(&r, 0, sizeof r); r.n = a.n > b.n ? a.n : b.n;
    for (int i = 0; i < r.n; ++i) r.c[i] = (i < a.n ? a.c[i] : 0.0) + sb * (i < b.n ? b.c[i] : 0.0);
    return r;
}
static poly_t pscale(poly_t a, double s) { for (int i = 0; i < a.n; ++i) a.c[i] *= s; return a; }
static double peval(poly_t a, double w) { double v = 0; for (int i = a.n - 1; i >= 0; --i) v = v * w + a.c[i]; return v; }

int orc_solver_shared(const double x1h[9], const double x2h[9], const double d1[3], const double d2[3], orc_model out[4]) {
    double P1[3], Q1[3];
    for (int k = 0; k < 3; ++k) {
        const int i = PAIR_I[k], j = PAIR_J[k];
        const double ax = d1[i] * x1h[3 * i] - d1[j] * x1h[3 * j], ay = d1[i] * x1h[3 * i + 1] - d1[j] * x1h[3 * j + 1];
        P1[k] = ax * ax + ay * ay;
        Q1[k] = (d1[i] - d1[j]) * (d1[i] - d1[j]);
    }
    const double bx = d2[0] * x2h[0] - d2[1] * x2h[3], by = d2[0] * x2h[1] - d2[1] * x2h[4];
    const double Pp = bx * bx + by * by, Qp = (d2[0] - d2[1]) * (d2[0] - d2[1]);
    const double r0 = x2h[0] * x2h[0] + x2h[1] * x2h[1], r1 = x2h[3] * x2h[3] + x2h[4] * x2h[4], r2 = x2h[6] * x2h[6] + x2h[7] * x2h[7];
    const double m02 = x2h[0] * x2h[6] + x2h[1] * x2h[7], m12 = x2h[3] * x2h[6] + x2h[4] * x2h[7];
    const poly_t N = pl(Q1[0], P1[0]), D = pl(Qp, Pp);
    const poly_t a0 = pl(d2[0] * d2[0], d2[0] * d2[0] * r0), a1 = pl(d2[1] * d2[1], d2[1] * d2[1] * r1);
    const poly_t c0 = pl(d2[0], d2[0] * m02), c1 = pl(d2[1], d2[1] * m12), e = pl(1.0, r2);
    const poly_t L02 = pl(Q1[1], P1[1]), L12 = pl(Q1[2], P1[2]);
    const poly_t dc = padd(c0, c1, -1.0);
    const poly_t U = padd(pmul(N, padd(a0, a1, -1.0)), pmul(padd(L02, L12, -1.0), D), -1.0);
    const poly_t t1 = pscale(pmul(pmul(N, pmul(dc, dc)), padd(pmul(L02, D), pmul(N, a0), -1.0)), 4.0);
    const poly_t t2 = pscale(pmul(pmul(c0, N), pmul(U, dc)), 4.0);
    const poly_t t3 = pmul(e, pmul(U, U));
    const poly_t q5 = padd(padd(t1, t2, 1.0), t3, -1.0); /* c[0] == 0 structurally */
    if (!(fabs(q5.c[5]) > 0)) return 0;
    double ws[4];
    const int nw = solve_quartic_real(q5.c[4] / q5.c[5], q5.c[3] / q5.c[5], q5.c[2] / q5.c[5], q5.c[1] / q5.c[5], ws);
    int n = 0;
    for (int r = 0; r < nw; ++r) {
        double w = ws[r];
        if (!(w > 0)) continue;
        double sig = peval(N, w) / peval(D, w);
        if (!(sig > 0)) continue;
        double rho = peval(U, w) / (2.0 * peval(N, w) * peval(dc, w));
        /* Newton polish of (w, sigma, rho) on the three distance equations */
        for (int it = 0; it < 4; ++it) {
            const double ga0 = peval(a0, w), ga1 = peval(a1, w), gc0 = peval(c0, w), gc1 = peval(c1, w), ge = peval(e, w);
            const double h0 = ga0 - 2 * rho * gc0 + rho * rho * ge, h1 = ga1 - 2 * rho * gc1 + rho * rho * ge;
            const double dh0 = a0.c[1] - 2 * rho * c0.c[1] + rho * rho * e.c[1], dh1 = a1.c[1] - 2 * rho * c1.c[1] + rho * rho * e.c[1];
            const double res[3] = {peval(N, w) - sig * peval(D, w), peval(L02, w) - sig * h0, peval(L12, w) - sig * h1};
            const double J[9] = {N.c[1] - sig * D.c[1], -peval(D, w), 0.0,
                                 L02.c[1] - sig * dh0, -h0, -sig * (-2 * gc0 + 2 * rho * ge),
                                 L12.c[1] - sig * dh1, -h1, -sig * (-2 * gc1 + 2 * rho * ge)};
            double dx[3];
            if (!solve3x3(J, res, dx)) break;
            w -= dx[0]; sig -= dx[1]; rho -= dx[2];
            if (fabs(dx[0]) + fabs(dx[1]) + fabs(dx[2]) < 1e-15 * (fabs(w) + fabs(sig) + fabs(rho))) break;
        }
        if (!(w > 0 && sig > 0 && rho > 0)) continue;
        const double f = 1.0 / sqrt(w), s = sqrt(sig), lam2 = rho * s;
        double X[9], Y[9], R[9];
        for (int i = 0; i < 3; ++i) {
            const double dy = (i < 2) ? s * d2[i] : lam2;
            X[3 * i] = d1[i] * x1h[3 * i] / f; X[3 * i + 1] = d1[i] * x1h[3 * i + 1] / f; X[3 * i + 2] = d1[i];
            Y[3 * i] = dy * x2h[3 * i] / f; Y[3 * i + 1] = dy * x2h[3 * i + 1] / f; Y[3 * i + 2] = dy;
        }
        model_init(&out[n]);
        align3(X, Y, R, out[n].t);
        orc_rotmat_to_quat(R, out[n].q);
        out[n].scale = s; out[n].f1 = f; out[n].f2 = f;
        if (++n == 4) break;
    }
    return n;
}
