// mdrp_math.h — per-lane fp64 arithmetic of the RePoseD RANSAC hot path: pose algebra, the four minimal solvers,
// Sampson/cheirality scoring and the per-correspondence residual/Jacobian of the hybrid refinement.
//
// Product code (gfx950).  Every function is straight-line register code for ONE lane: one lane = one minimal
// sample (solvers) or one hypothesis (scoring) — nothing here is a dense contraction, so no MFMA.  Local arrays are
// only indexed by compile-time constants after unrolling (runtime-indexed arrays would go to scratch).
// The same header also compiles with a host C++ compiler (MDRP_HD expands to `inline`) for the CPU unit tests of
// the arithmetic (tests/hostmath); that build is test scaffolding, not a fallback: the library has no CPU path.
//
// Reference functions replaced (binary only): p3p @0xecd50, relpose_monodepth_3pt @0x155ca0,
// relpose_monodepth_3pt_shared_focal @0x18fdf0, relpose_monodepth_3pt_varying_focal @0x19bcd0,
// compute_sampson_msac_score @0x4f61d0/@0x4f65d0, check_cheirality @0x1dce00, and the Jacobian accumulators
// behind refine_monodepth_*relpose @0x261030/@0x2592e0/@0x260fa0.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MDRP_HD __host__ __device__ __forceinline__
#else
#define MDRP_HD inline
#endif

namespace mdrp {

struct Model { // == mdrp_model
    double q[4];
    double t[3];
    double scale, shift1, shift2;
    double f1, f2;
};

MDRP_HD void model_identity(Model &m) {
    m.q[0] = 1.0; m.q[1] = 0.0; m.q[2] = 0.0; m.q[3] = 0.0;
    m.t[0] = 0.0; m.t[1] = 0.0; m.t[2] = 0.0;
    m.scale = 1.0; m.shift1 = 0.0; m.shift2 = 0.0; m.f1 = 1.0; m.f2 = 1.0;
}

// ---------------------------------------------------------------- division and square root of the minimal solvers
// On the device an IEEE fp64 division is an 11-instruction dependent chain (v_div_scale x 2, v_rcp, four FMAs, v_div_fmas, v_div_fixup) and a
// square root ~15 with its range scaling; the calibrated P3P path has 71 + 35 of them among 5 800 instructions (round 4 count).  The device
// build takes the hardware seed and two Newton steps (<= 1.5 ulp instead of correctly rounded: the same class of difference as the FMA
// contraction it already has against the oracle; every solution is Newton-polished against its own equations afterwards).  The host build of
// this header — the CPU tests against the oracle — keeps the IEEE operations.  MDRP_SOLVER_IEEE_DIV=1 restores them on the device.
MDRP_HD double sv_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
#else
    return 1.0 / x;
#endif
}
MDRP_HD double sv_div(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return a * sv_rcp(b);
#else
    return a / b;
#endif
}
MDRP_HD double sv_rsqrt(double x) { // 1 / sqrt(x), x > 0
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * fma(-h * y, y, 1.5);
    return y * fma(-h * y, y, 1.5);
#else
    return 1.0 / sqrt(x);
#endif
}
MDRP_HD double sv_sqrt(double x) { // sqrt(x) without the compiler's range scaling (the solvers' arguments are O(1) quantities)
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    const double g = x * r;
    const double s = fma(0.5 * r, fma(-g, g, x), g); // Heron correction: <= 1 ulp; NaN for x < 0 as sqrt, and for x = 0 (inf * 0)
    return x == 0.0 ? x : s;
#else
    return sqrt(x);
#endif
}

// ---------------------------------------------------------------- pose algebra
MDRP_HD void quat_to_R(const double q[4], double R[9]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

MDRP_HD void R_to_quat(const double R[9], double q[4]) {
    const double tr = R[0] + R[4] + R[8];
    double w, x, y, z;
    if (tr > 0) {
        double t = sv_sqrt(tr + 1.0);
        w = 0.5 * t; t = sv_div(0.5, t);
        x = (R[7] - R[5]) * t; y = (R[2] - R[6]) * t; z = (R[3] - R[1]) * t;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {
        double t = sv_sqrt(R[0] - R[4] - R[8] + 1.0);
        x = 0.5 * t; t = sv_div(0.5, t);
        w = (R[7] - R[5]) * t; y = (R[3] + R[1]) * t; z = (R[6] + R[2]) * t;
    } else if (R[4] >= R[8]) {
        double t = sv_sqrt(R[4] - R[8] - R[0] + 1.0);
        y = 0.5 * t; t = sv_div(0.5, t);
        w = (R[2] - R[6]) * t; z = (R[7] + R[5]) * t; x = (R[1] + R[3]) * t;
    } else {
        double t = sv_sqrt(R[8] - R[0] - R[4] + 1.0);
        z = 0.5 * t; t = sv_div(0.5, t);
        w = (R[3] - R[1]) * t; x = (R[2] + R[6]) * t; y = (R[5] + R[7]) * t;
    }
    const double inv = sv_rsqrt(w * w + x * x + y * y + z * z);
    q[0] = w * inv; q[1] = x * inv; q[2] = y * inv; q[3] = z * inv;
}

// E = [t]x R (row-major)
MDRP_HD void essential_from_Rt(const double R[9], const double t[3], double E[9]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        E[0 + c] = t[1] * R[6 + c] - t[2] * R[3 + c];
        E[3 + c] = t[2] * R[0 + c] - t[0] * R[6 + c];
        E[6 + c] = t[0] * R[3 + c] - t[1] * R[0 + c];
    }
}

// F = diag(1,1,f2) E diag(1,1,f1)   (score_model @0x4fac60 / @0x4faf90)
MDRP_HD void fundamental_from_E(const double E[9], double f1, double f2, double F[9]) {
    F[0] = E[0]; F[1] = E[1]; F[2] = E[2] * f1;
    F[3] = E[3]; F[4] = E[4]; F[5] = E[5] * f1;
    F[6] = E[6] * f2; F[7] = E[7] * f2; F[8] = E[8] * f1 * f2;
}

MDRP_HD double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
MDRP_HD void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// rigid alignment of two congruent triangles: R (X_i - X_0) = Y_i - Y_0, t = Y_0 - R X_0
MDRP_HD void align3(const double X[9], const double Y[9], double R[9], double t[3]) {
    double a[3], b[3], c[3], u[3], v[3], w[3], bc[3], ca[3], ab[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { a[i] = X[3 + i] - X[i]; b[i] = X[6 + i] - X[i]; u[i] = Y[3 + i] - Y[i]; v[i] = Y[6 + i] - Y[i]; }
    cross3(a, b, c); cross3(u, v, w);
    cross3(b, c, bc); cross3(c, a, ca); cross3(a, b, ab);
    const double idet = sv_rcp(dot3(a, bc));
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[3 * i + j] = (u[i] * bc[j] + v[i] * ca[j] + w[i] * ab[j]) * idet;
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = Y[i] - (R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2]);
}

// ---------------------------------------------------------------- univariate polynomials
// real roots of x^3 + b x^2 + c x + d; Newton-polished; r0 always valid, (r1,r2) valid iff return value is 3
// FAST (device builds only; the minimal solvers of the monodepth estimators): the three-real-root branch takes acos from the 8-term
// form sqrt(1 - x) P(x) (Abramowitz & Stegun 4.4.46, 2e-8) and cos / sin of phi in [0, pi / 3] from their Taylor polynomials (1e-13)
// instead of the library's acos + three cos (606 of the P3P kernel's 4 980 instructions); the three Newton steps below bring the roots
// to the rounding level of the closed form either way (quadratic convergence from 1e-8), and what the solvers do with a root
// is polished against its own equations afterwards (p3p_finish, the Newton loops of the shift / focal solvers).
template <bool FAST = false>
MDRP_HD int solve_cubic_real(double b, double c, double d, double &r0, double &r1, double &r2) {
    const double third = 1.0 / 3.0;
    const double p = c - b * b * third;
    const double q = sv_div(2.0 * b * b * b, 27.0) - b * c * third + d;
    const double disc = q * q * 0.25 + sv_div(p * p * p, 27.0);
    int n;
    if (disc > 0) {
        const double sq = sv_sqrt(disc);
        r0 = cbrt(-0.5 * q + sq) + cbrt(-0.5 * q - sq) - b * third;
        r1 = r0; r2 = r0;
        n = 1;
    } else {
        const double rr = sv_sqrt(-p * third);
        double arg = (rr > 0) ? sv_div(-0.5 * q, rr * rr * rr) : 0.0;
        arg = arg > 1 ? 1 : (arg < -1 ? -1 : arg);
#if defined(__HIP_DEVICE_COMPILE__)
        if (FAST) {
            const double ax = fabs(arg);
            double pa = -0.0012624911;
            pa = fma(pa, ax, 0.0066700901); pa = fma(pa, ax, -0.0170881256); pa = fma(pa, ax, 0.0308918810); pa = fma(pa, ax, -0.0501743046);
            pa = fma(pa, ax, 0.0889789874); pa = fma(pa, ax, -0.2145988016); pa = fma(pa, ax, 1.5707963050);
            const double ac = sv_sqrt(1.0 - ax) * pa;                                  // acos(|arg|)
            const double phi = (arg < 0.0 ? 3.14159265358979323846 - ac : ac) * third; // in [0, pi / 3]
            const double x2 = phi * phi;
            double cs = -1.0 / 87178291200.0, sn = -1.0 / 1307674368000.0;             // -x^14 / 14!, -x^15 / 15!
            cs = fma(cs, x2, 1.0 / 479001600.0); sn = fma(sn, x2, 1.0 / 6227020800.0);
            cs = fma(cs, x2, -1.0 / 3628800.0); sn = fma(sn, x2, -1.0 / 39916800.0);
            cs = fma(cs, x2, 1.0 / 40320.0); sn = fma(sn, x2, 1.0 / 362880.0);
            cs = fma(cs, x2, -1.0 / 720.0); sn = fma(sn, x2, -1.0 / 5040.0);
            cs = fma(cs, x2, 1.0 / 24.0); sn = fma(sn, x2, 1.0 / 120.0);
            cs = fma(cs, x2, -0.5); sn = fma(sn, x2, -1.0 / 6.0);
            cs = fma(cs, x2, 1.0); sn = fma(sn, x2, 1.0) * phi;
            const double h = 0.86602540378443864676 * sn;                              // sin(2 pi / 3) sin(phi)
            r0 = 2.0 * rr * cs - b * third;
            r1 = 2.0 * rr * (h - 0.5 * cs) - b * third;                                // cos(phi - 2 pi / 3)
            r2 = 2.0 * rr * (-h - 0.5 * cs) - b * third;                               // cos(phi - 4 pi / 3)
            n = 3;
        } else
#endif
        {
            const double phi = acos(arg) * third;
            const double tp3 = 2.0943951023931954923; // 2 pi / 3
            r0 = 2.0 * rr * cos(phi) - b * third;
            r1 = 2.0 * rr * cos(phi - tp3) - b * third;
            r2 = 2.0 * rr * cos(phi - 2.0 * tp3) - b * third;
            n = 3;
        }
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const double f0 = ((r0 + b) * r0 + c) * r0 + d, g0 = (3.0 * r0 + 2.0 * b) * r0 + c;
        const double f1 = ((r1 + b) * r1 + c) * r1 + d, g1 = (3.0 * r1 + 2.0 * b) * r1 + c;
        const double f2 = ((r2 + b) * r2 + c) * r2 + d, g2 = (3.0 * r2 + 2.0 * b) * r2 + c;
        if (g0 != 0.0) r0 -= sv_div(f0, g0);
        if (g1 != 0.0) r1 -= sv_div(f1, g1);
        if (g2 != 0.0) r2 -= sv_div(f2, g2);
    }
    return n;
}

// real roots of x^4 + b x^3 + c x^2 + d x + e (Ferrari, resolvent cubic), Newton-polished.
// Returns a 4-bit validity mask for roots[0..3] (pairs {0,1} and {2,3} come from the two quadratic factors).
MDRP_HD int solve_quartic_real(double b, double c, double d, double e, double roots[4]) {
    const double b2 = b * b;
    const double p = c - 0.375 * b2;
    const double q = d - 0.5 * b * c + 0.125 * b2 * b;
    const double r = e - 0.25 * b * d + b2 * c * 0.0625 - 3.0 * b2 * b2 / 256.0;
    double z0, z1, z2;
    const int nz = solve_cubic_real<true>(2.0 * p, p * p - 4.0 * r, -q * q, z0, z1, z2);
    double z = z0;
    if (nz == 3) { z = z1 > z ? z1 : z; z = z2 > z ? z2 : z; }
    int mask = 0;
    double y0 = 0, y1 = 0, y2 = 0, y3 = 0;
    const double scale = fabs(p) + sv_sqrt(fabs(r)) + 1e-300;
    if (z <= 1e-14 * scale) { // biquadratic
        const double disc = p * p - 4.0 * r;
        if (disc >= 0) {
            const double sq = sv_sqrt(disc);
            const double ya = 0.5 * (-p + sq), yb = 0.5 * (-p - sq);
            if (ya >= 0) { y0 = sv_sqrt(ya); y1 = -y0; mask |= 3; }
            if (yb >= 0) { y2 = sv_sqrt(yb); y3 = -y2; mask |= 12; }
        }
    } else {
        const double s = sv_sqrt(z);
        const double qs = sv_div(q, s);
        const double t1 = 0.5 * (p + z - qs), t2 = 0.5 * (p + z + qs);
        const double disc1 = z - 4.0 * t1, disc2 = z - 4.0 * t2;
        if (disc1 >= 0) { const double sq = sv_sqrt(disc1); y0 = 0.5 * (-s + sq); y1 = 0.5 * (-s - sq); mask |= 3; }
        if (disc2 >= 0) { const double sq = sv_sqrt(disc2); y2 = 0.5 * (s + sq); y3 = 0.5 * (s - sq); mask |= 12; }
    }
    double x[4] = {y0 - 0.25 * b, y1 - 0.25 * b, y2 - 0.25 * b, y3 - 0.25 * b};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const double f = (((x[k] + b) * x[k] + c) * x[k] + d) * x[k] + e;
            const double fp = ((4.0 * x[k] + 3.0 * b) * x[k] + 2.0 * c) * x[k] + d;
            if (fp != 0.0) x[k] -= sv_div(f, fp);
        }
        roots[k] = x[k];
    }
    return mask;
}

MDRP_HD bool solve3x3(const double A[9], const double b[3], double x[3]) {
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    if (!(fabs(det) > 0)) return false;
    const double id = sv_rcp(det);
    x[0] = (c00 * b[0] + (A[2] * A[7] - A[1] * A[8]) * b[1] + (A[1] * A[5] - A[2] * A[4]) * b[2]) * id;
    x[1] = (c01 * b[0] + (A[0] * A[8] - A[2] * A[6]) * b[1] + (A[2] * A[3] - A[0] * A[5]) * b[2]) * id;
    x[2] = (c02 * b[0] + (A[1] * A[6] - A[0] * A[7]) * b[1] + (A[0] * A[4] - A[1] * A[3]) * b[2]) * id;
    return true;
}

// ---------------------------------------------------------------- P3P
// depths l_i > 0 with |l_i x_i - l_j x_j|^2 = a_ij.  Two homogeneous conics in (l0:l1:l2) -> degenerate member of
// their pencil (cubic) -> real line pair -> intersect each line with a conic of the pencil -> <= 4 points.
// symmetric 3x3 stored as {00,01,02,11,12,22}
MDRP_HD void sym_adj(const double C[6], double A[6]) {
    A[0] = C[3] * C[5] - C[4] * C[4];
    A[1] = C[2] * C[4] - C[1] * C[5];
    A[2] = C[1] * C[4] - C[2] * C[3];
    A[3] = C[0] * C[5] - C[2] * C[2];
    A[4] = C[1] * C[2] - C[0] * C[4];
    A[5] = C[0] * C[3] - C[1] * C[1];
}
MDRP_HD double sym_det(const double C[6]) {
    return C[0] * (C[3] * C[5] - C[4] * C[4]) - C[1] * (C[1] * C[5] - C[4] * C[2]) + C[2] * (C[1] * C[4] - C[3] * C[2]);
}
MDRP_HD double sym_trprod(const double A[6], const double B[6]) {
    return A[0] * B[0] + A[3] * B[3] + A[5] * B[5] + 2.0 * (A[1] * B[1] + A[2] * B[2] + A[4] * B[4]);
}
MDRP_HD double sym_quad(const double C[6], const double *u, const double *v) {
    return u[0] * (C[0] * v[0] + C[1] * v[1] + C[2] * v[2]) + u[1] * (C[1] * v[0] + C[3] * v[1] + C[4] * v[2]) +
           u[2] * (C[2] * v[0] + C[4] * v[1] + C[5] * v[2]);
}

// quality of a pencil member as a REAL line pair: max diagonal of -adj(C), normalised
MDRP_HD double linepair_quality(const double D1[6], const double D2[6], double g) {
    double C[6], A[6], nrm = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { C[i] = D1[i] + g * D2[i]; nrm += C[i] * C[i]; }
    sym_adj(C, A);
    double mx = -A[0];
    mx = (-A[3] > mx) ? -A[3] : mx;
    mx = (-A[5] > mx) ? -A[5] : mx;
    return sv_div(mx, nrm);
}

// one candidate (tau:sigma) on a line spanned by u,v -> scaled, sign-fixed, polished depths; returns validity
MDRP_HD bool p3p_finish(double tau, double sig, const double u[3], const double v[3], double m01, double m02, double m12,
                        double a01, double a02, double a12, double lam[3]) {
    double l0 = sig * u[0] + tau * v[0], l1 = sig * u[1] + tau * v[1], l2 = sig * u[2] + tau * v[2];
    double qv, av;
    if (a12 >= a01 && a12 >= a02) { qv = l1 * l1 + l2 * l2 - 2 * m12 * l1 * l2; av = a12; }
    else if (a02 >= a01) { qv = l0 * l0 + l2 * l2 - 2 * m02 * l0 * l2; av = a02; }
    else { qv = l0 * l0 + l1 * l1 - 2 * m01 * l0 * l1; av = a01; }
    if (!(qv > 0)) return false;
    double sc = sv_sqrt(sv_div(av, qv));
    if (l0 < 0) sc = -sc;
    l0 *= sc; l1 *= sc; l2 *= sc;
    if (!(l0 > 0 && l1 > 0 && l2 > 0)) return false;
    const double tol = 1e-15 * (a01 + a02 + a12);
    for (int it = 0; it < 5; ++it) {
        const double r0 = l0 * l0 + l1 * l1 - 2 * m01 * l0 * l1 - a01;
        const double r1 = l0 * l0 + l2 * l2 - 2 * m02 * l0 * l2 - a02;
        const double r2 = l1 * l1 + l2 * l2 - 2 * m12 * l1 * l2 - a12;
        if (fabs(r0) + fabs(r1) + fabs(r2) < tol) break;
        const double J[9] = {2 * (l0 - m01 * l1), 2 * (l1 - m01 * l0), 0.0,
                             2 * (l0 - m02 * l2), 0.0, 2 * (l2 - m02 * l0),
                             0.0, 2 * (l1 - m12 * l2), 2 * (l2 - m12 * l1)};
        const double res[3] = {r0, r1, r2};
        double dx[3];
        if (!solve3x3(J, res, dx)) break;
        l0 -= dx[0]; l1 -= dx[1]; l2 -= dx[2];
    }
    if (!(l0 > 0 && l1 > 0 && l2 > 0)) return false;
    lam[0] = l0; lam[1] = l1; lam[2] = l2;
    return true;
}

// returns number of depth triples written to L (<= 4)
MDRP_HD int p3p_depths(double m01, double m02, double m12, double a01, double a02, double a12, double L[4][3]) {
    const double D1[6] = {a12, -a12 * m01, 0.0, a12 - a01, a01 * m12, -a01};
    const double D2[6] = {a12, 0.0, -a12 * m02, -a02, a02 * m12, a12 - a02};
    double A1[6], A2[6];
    sym_adj(D1, A1); sym_adj(D2, A2);
    const double c3 = sym_det(D2), c2 = sym_trprod(A2, D1), c1 = sym_trprod(A1, D2), c0 = sym_det(D1);
    if (!(fabs(c3) > 1e-300)) return 0;
    const double ic3 = sv_rcp(c3);
    double g0, g1, g2;
    const int nr = solve_cubic_real<true>(c2 * ic3, c1 * ic3, c0 * ic3, g0, g1, g2);
    double g = g0, best = linepair_quality(D1, D2, g0);
    if (nr == 3) {
        const double q1 = linepair_quality(D1, D2, g1), q2 = linepair_quality(D1, D2, g2);
        if (q1 > best) { best = q1; g = g1; }
        if (q2 > best) { best = q2; g = g2; }
    }
    if (!(best > 0)) return 0;
    double C[6], B[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) C[i] = D1[i] + g * D2[i];
    sym_adj(C, B);
#pragma unroll
    for (int i = 0; i < 6; ++i) B[i] = -B[i];
    double p0, p1, p2; // p = l x m, B = p p'
    if (B[0] >= B[3] && B[0] >= B[5]) { const double s = sv_sqrt(B[0]), is = sv_rcp(s); p0 = s; p1 = B[1] * is; p2 = B[2] * is; }
    else if (B[3] >= B[5]) { const double s = sv_sqrt(B[3]), is = sv_rcp(s); p0 = B[1] * is; p1 = s; p2 = B[4] * is; }
    else { const double s = sv_sqrt(B[5]), is = sv_rcp(s); p0 = B[2] * is; p1 = B[4] * is; p2 = s; }
    // M = C + [p]x = 2 m l' : rows ~ l, columns ~ m
    const double M[9] = {C[0], C[1] - p2, C[2] + p1, C[1] + p2, C[3], C[4] - p0, C[2] - p1, C[4] + p0, C[5]};
    double la[3], lb[3];
    {
        const double r0 = M[0] * M[0] + M[1] * M[1] + M[2] * M[2], r1 = M[3] * M[3] + M[4] * M[4] + M[5] * M[5],
                     r2 = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
        if (r0 >= r1 && r0 >= r2) { la[0] = M[0]; la[1] = M[1]; la[2] = M[2]; }
        else if (r1 >= r2) { la[0] = M[3]; la[1] = M[4]; la[2] = M[5]; }
        else { la[0] = M[6]; la[1] = M[7]; la[2] = M[8]; }
        const double k0 = M[0] * M[0] + M[3] * M[3] + M[6] * M[6], k1 = M[1] * M[1] + M[4] * M[4] + M[7] * M[7],
                     k2 = M[2] * M[2] + M[5] * M[5] + M[8] * M[8];
        if (k0 >= k1 && k0 >= k2) { lb[0] = M[0]; lb[1] = M[3]; lb[2] = M[6]; }
        else if (k1 >= k2) { lb[0] = M[1]; lb[1] = M[4]; lb[2] = M[7]; }
        else { lb[0] = M[2]; lb[1] = M[5]; lb[2] = M[8]; }
    }
    const bool useD2 = fabs(g) < 1.0; // the pencil member that does NOT vanish on the lines
    double Dq[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) Dq[i] = useD2 ? D2[i] : D1[i];
    int n = 0;
#pragma unroll
    for (int li = 0; li < 2; ++li) {
        const double l0 = li ? lb[0] : la[0], l1 = li ? lb[1] : la[1], l2 = li ? lb[2] : la[2];
        // two points spanning the line: u = l x e_a, v = l x e_b, (a,b) = the two axes other than argmax|l|
        double u[3], v[3];
        const double f0 = fabs(l0), f1 = fabs(l1), f2 = fabs(l2);
        if (f0 >= f1 && f0 >= f2) { // k=0: a=1,b=2 : l x e1 = (-l2,0,l0), l x e2 = (l1,-l0,0)
            u[0] = -l2; u[1] = 0; u[2] = l0; v[0] = l1; v[1] = -l0; v[2] = 0;
        } else if (f1 >= f2) {       // k=1: a=2,b=0 : l x e2 = (l1,-l0,0), l x e0 = (0,l2,-l1)
            u[0] = l1; u[1] = -l0; u[2] = 0; v[0] = 0; v[1] = l2; v[2] = -l1;
        } else {                     // k=2: a=0,b=1 : l x e0 = (0,l2,-l1), l x e1 = (-l2,0,l0)
            u[0] = 0; u[1] = l2; u[2] = -l1; v[0] = -l2; v[1] = 0; v[2] = l0;
        }
        const double qa = sym_quad(Dq, v, v), qb = sym_quad(Dq, u, v), qc = sym_quad(Dq, u, u);
        const double disc = qb * qb - qa * qc;
        if (disc >= 0) {
            const double sq = sv_sqrt(disc);
            const double qq = -(qb + (qb >= 0 ? sq : -sq));
            double lam[3];
            if (n < 4 && p3p_finish(qq, qa, u, v, m01, m02, m12, a01, a02, a12, lam)) { L[n][0] = lam[0]; L[n][1] = lam[1]; L[n][2] = lam[2]; ++n; }
            if (n < 4 && p3p_finish(qc, qq, u, v, m01, m02, m12, a01, a02, a12, lam)) { L[n][0] = lam[0]; L[n][1] = lam[1]; L[n][2] = lam[2]; ++n; }
        }
    }
    return n;
}

// ---------------------------------------------------------------- minimal solvers (inputs: 3 correspondences)
// x1,x2: normalised image points (x,y) per correspondence, homogeneous z = 1 implied.
struct Sample3 {
    double x1[3][2], x2[3][2], d1[3], d2[3];
};

// a-6': P3P on X_k = d1_k (x1_k,1) and unit bearings of image 2; scale from the first correspondence (x component)
MDRP_HD int solver_calib_p3p(const Sample3 &s, Model out[4]) {
    double X[9], xb[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double inv = sv_rsqrt(s.x2[i][0] * s.x2[i][0] + s.x2[i][1] * s.x2[i][1] + 1.0);
        X[3 * i] = s.d1[i] * s.x1[i][0]; X[3 * i + 1] = s.d1[i] * s.x1[i][1]; X[3 * i + 2] = s.d1[i];
        xb[3 * i] = s.x2[i][0] * inv; xb[3 * i + 1] = s.x2[i][1] * inv; xb[3 * i + 2] = inv;
    }
    double d01[3], d02[3], d12[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { d01[c] = X[c] - X[3 + c]; d02[c] = X[c] - X[6 + c]; d12[c] = X[3 + c] - X[6 + c]; }
    double L[4][3];
    const int n = p3p_depths(dot3(xb, xb + 3), dot3(xb, xb + 6), dot3(xb + 3, xb + 6), dot3(d01, d01), dot3(d02, d02),
                             dot3(d12, d12), L);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < n) {
            double Y[9], R[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) Y[3 * i + c] = L[k][i] * xb[3 * i + c];
            model_identity(out[k]);
            align3(X, Y, R, out[k].t);
            R_to_quat(R, out[k].q);
            double Rq[9];
            quat_to_R(out[k].q, Rq);
            const double px = Rq[0] * X[0] + Rq[1] * X[1] + Rq[2] * X[2] + out[k].t[0];
            out[k].scale = sv_div(px, s.d2[0] * s.x2[0][0]);
        }
    }
    return n;
}

// Does the reference's P3P hand back NaN poses for this sample?  Its p3p() (Ding et al.: one real root s of a cubic — the largest when there are
// three —, the degenerate conic C(s) of the pencil split into two lines) takes the square root of the largest diagonal entry of -adj(C) without a
// sign test; where C(s) is a POINT conic all three are negative, both lines are NaN, every test that would discard a solution is a comparison with NaN,
// and four NaN poses come out (black-box: exactly the 398 of 12 000 noisy samples where this predicate holds; our solver, rightly, finds no real pose
// on any of them).  A NaN model scores N * thr with no inlier: a record while nothing has been scored yet, one LO that cannot change anything, and
// the run's answer if no sample ever gives a real pose.  Only reachable with three real roots (a single real root's conic is a real line pair), so
// the trigonometric branch below runs for ~5 % of the samples.  Same arithmetic as oracle/orc_solvers.c orc_p3p_reference_nan.
MDRP_HD bool p3p_reference_nan(const Sample3 &sm) {
    double xs[3][3], X[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double inv = sv_rsqrt(sm.x2[i][0] * sm.x2[i][0] + sm.x2[i][1] * sm.x2[i][1] + 1.0);
        X[i][0] = sm.d1[i] * sm.x1[i][0]; X[i][1] = sm.d1[i] * sm.x1[i][1]; X[i][2] = sm.d1[i];
        xs[i][0] = sm.x2[i][0] * inv; xs[i][1] = sm.x2[i][1] * inv; xs[i][2] = inv;
    }
    double a01 = 0, a02 = 0, a12 = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a01 += (X[0][k] - X[1][k]) * (X[0][k] - X[1][k]); a02 += (X[0][k] - X[2][k]) * (X[0][k] - X[2][k]); a12 += (X[1][k] - X[2][k]) * (X[1][k] - X[2][k]);
    }
    // the largest of the three distances becomes "12": swap bearing 0 with 2 (or 1) — as selects, no divergent copies
    const bool sw02 = a01 > a02 && a01 > a12, sw01 = !(a01 > a02) && a02 > a12;
    double x0[3], x1v[3], x2v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        x0[k] = sw02 ? xs[2][k] : (sw01 ? xs[1][k] : xs[0][k]);
        x1v[k] = sw01 ? xs[0][k] : xs[1][k];
        x2v[k] = sw02 ? xs[0][k] : xs[2][k];
    }
    if (sw02) { const double t = a01; a01 = a12; a12 = t; }
    if (sw01) { const double t = a02; a02 = a12; a12 = t; }
    const double a12d = 1.0 / a12, a = a01 * a12d, b = a02 * a12d;
    const double m01 = dot3(x0, x1v), m02 = dot3(x0, x2v), m12 = dot3(x1v, x2v);
    const double m12sq = -m12 * m12 + 1.0, m02sq = -1.0 + m02 * m02, m01sq = -1.0 + m01 * m01;
    const double ab = a * b, bsq = b * b, asq = a * a, m013 = -2.0 + 2.0 * m01 * m02 * m12;
    const double bsqm12sq = bsq * m12sq, asqm12sq = asq * m12sq, abm12sq = 2.0 * ab * m12sq;
    const double k3i = 1.0 / (bsqm12sq + b * m02sq);
    const double k2 = k3i * ((-1.0 + a) * m02sq + abm12sq + bsqm12sq + b * m013);
    const double k1 = k3i * (asqm12sq + abm12sq + a * m013 + (-1.0 + b) * m01sq);
    const double k0 = k3i * (asqm12sq + a * m01sq);
    const double ca = k1 - k2 * k2 / 3.0;
    const double cb = (2.0 * k2 * k2 * k2 - 9.0 * k2 * k1) / 27.0 + k0;
    const double cc = cb * cb / 4.0 + ca * ca * ca / 27.0;
    if (!(cc < 0)) return false; // one real root (or the degenerate cc == 0 / NaN cases): never a point conic
    const double arg = 3.0 * cb / (2.0 * ca) * sqrt(-3.0 / ca);
    const double s = 2.0 * sqrt(-ca / 3.0) * cos(acos(arg) / 3.0) - k2 / 3.0;
    const double C00 = -a + s * (1 - b), C01 = -m02 * s, C02 = a * m12 + b * m12 * s, C11 = s + 1, C12 = -m01, C22 = -a - b * s + 1;
    const double A0 = C12 * C12 - C11 * C22, A1 = C02 * C02 - C00 * C22, A2 = C01 * C01 - C00 * C11;
    const double mx = A0 > A1 ? (A0 > A2 ? A0 : A2) : (A1 > A2 ? A1 : A2);
    return mx < 0;
}

// a-4: scale + two shifts.  |(d1_i+u) x1_i - (d1_j+u) x1_j|^2 = s^2 |(d2_i+v) x2_i - (d2_j+v) x2_j|^2 for the 3
// pairs: linear in (a,b,c) = (s^2, s^2 v, s^2 v^2), quadratic in u; a c = b^2 -> quartic in u.
MDRP_HD int solver_calib_shift(const Sample3 &s, Model out[4]) {
    double A1[3], B1[3], C1[3], M[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = (k == 2) ? 1 : 0, j = (k == 0) ? 1 : 2;
        double p[3], q[3];
        p[0] = s.d1[i] * s.x1[i][0] - s.d1[j] * s.x1[j][0]; p[1] = s.d1[i] * s.x1[i][1] - s.d1[j] * s.x1[j][1]; p[2] = s.d1[i] - s.d1[j];
        q[0] = s.x1[i][0] - s.x1[j][0]; q[1] = s.x1[i][1] - s.x1[j][1]; q[2] = 0.0;
        A1[k] = dot3(p, p); B1[k] = dot3(p, q); C1[k] = dot3(q, q);
        p[0] = s.d2[i] * s.x2[i][0] - s.d2[j] * s.x2[j][0]; p[1] = s.d2[i] * s.x2[i][1] - s.d2[j] * s.x2[j][1]; p[2] = s.d2[i] - s.d2[j];
        q[0] = s.x2[i][0] - s.x2[j][0]; q[1] = s.x2[i][1] - s.x2[j][1]; q[2] = 0.0;
        M[3 * k] = dot3(p, p); M[3 * k + 1] = 2.0 * dot3(p, q); M[3 * k + 2] = dot3(q, q);
    }
    double g0[3], g1[3], g2[3];
    const double B2[3] = {2 * B1[0], 2 * B1[1], 2 * B1[2]};
    if (!solve3x3(M, A1, g0) || !solve3x3(M, B2, g1) || !solve3x3(M, C1, g2)) return 0;
    const double k4 = g2[0] * g2[2] - g2[1] * g2[1];
    const double k3 = g1[0] * g2[2] + g2[0] * g1[2] - 2.0 * g1[1] * g2[1];
    const double k2 = g0[0] * g2[2] + g1[0] * g1[2] + g2[0] * g0[2] - g1[1] * g1[1] - 2.0 * g0[1] * g2[1];
    const double k1 = g0[0] * g1[2] + g1[0] * g0[2] - 2.0 * g0[1] * g1[1];
    const double k0 = g0[0] * g0[2] - g0[1] * g0[1];
    if (!(fabs(k4) > 0)) return 0;
    const double ik4 = sv_rcp(k4);
    double us[4];
    const int mask = solve_quartic_real(k3 * ik4, k2 * ik4, k1 * ik4, k0 * ik4, us);
    int n = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (!((mask >> r) & 1)) continue;
        double u = us[r];
        const double a = g0[0] + u * (g1[0] + u * g2[0]);
        const double b = g0[1] + u * (g1[1] + u * g2[1]);
        if (!(a > 0)) continue;
        double sc = sv_sqrt(a), v = sv_div(b, a);
        for (int it = 0; it < 5; ++it) { // Newton polish of (s,u,v) on the three distance equations
            double J[9], res[3], dx[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double lhs = A1[k] + u * (2.0 * B1[k] + u * C1[k]);
                const double rhs = M[3 * k] + v * (M[3 * k + 1] + v * M[3 * k + 2]);
                res[k] = lhs - sc * sc * rhs;
                J[3 * k] = -2.0 * sc * rhs;
                J[3 * k + 1] = 2.0 * B1[k] + 2.0 * u * C1[k];
                J[3 * k + 2] = -sc * sc * (M[3 * k + 1] + 2.0 * v * M[3 * k + 2]);
            }
            if (!solve3x3(J, res, dx)) break;
            sc -= dx[0]; u -= dx[1]; v -= dx[2];
            if (fabs(dx[0]) + fabs(dx[1]) + fabs(dx[2]) < 1e-15 * (fabs(sc) + fabs(u) + fabs(v))) break;
        }
        if (!(sc > 0)) continue;
        bool pos = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) pos = pos && (s.d1[i] + u > 0) && (s.d2[i] + v > 0);
        if (!pos) continue;
        double X[9], Y[9], R[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double e1 = s.d1[i] + u, e2 = sc * (s.d2[i] + v);
            X[3 * i] = e1 * s.x1[i][0]; X[3 * i + 1] = e1 * s.x1[i][1]; X[3 * i + 2] = e1;
            Y[3 * i] = e2 * s.x2[i][0]; Y[3 * i + 1] = e2 * s.x2[i][1]; Y[3 * i + 2] = e2;
        }
        Model m;
        model_identity(m);
        align3(X, Y, R, m.t);
        R_to_quat(R, m.q);
        m.scale = sc; m.shift1 = u; m.shift2 = v;
        // n is a compile-time-unknown index: write through a small switch to keep `out` in registers
        if (n == 0) out[0] = m; else if (n == 1) out[1] = m; else if (n == 2) out[2] = m; else out[3] = m;
        ++n;
    }
    return n;
}

// a-6: varying focal, linear in (1/f1^2, s^2/f2^2, s^2)
MDRP_HD int solver_varying(const Sample3 &s, Model out[4]) {
    double A[9], rhs[3], sol[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = (k == 2) ? 1 : 0, j = (k == 0) ? 1 : 2;
        const double ax = s.d1[i] * s.x1[i][0] - s.d1[j] * s.x1[j][0], ay = s.d1[i] * s.x1[i][1] - s.d1[j] * s.x1[j][1];
        const double bx = s.d2[i] * s.x2[i][0] - s.d2[j] * s.x2[j][0], by = s.d2[i] * s.x2[i][1] - s.d2[j] * s.x2[j][1];
        const double dz1 = s.d1[i] - s.d1[j], dz2 = s.d2[i] - s.d2[j];
        A[3 * k] = ax * ax + ay * ay;
        A[3 * k + 1] = -(bx * bx + by * by);
        A[3 * k + 2] = -dz2 * dz2;
        rhs[k] = -dz1 * dz1;
    }
    if (!solve3x3(A, rhs, sol)) return 0;
    if (!(sol[0] > 0 && sol[1] > 0 && sol[2] > 0)) return 0;
    const double f1 = sv_rsqrt(sol[0]), sc = sv_sqrt(sol[2]), f2 = sv_sqrt(sv_div(sol[2], sol[1]));
    double X[9], Y[9], R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        X[3 * i] = sv_div(s.d1[i] * s.x1[i][0], f1); X[3 * i + 1] = sv_div(s.d1[i] * s.x1[i][1], f1); X[3 * i + 2] = s.d1[i];
        Y[3 * i] = sv_div(sc * s.d2[i] * s.x2[i][0], f2); Y[3 * i + 1] = sv_div(sc * s.d2[i] * s.x2[i][1], f2); Y[3 * i + 2] = sc * s.d2[i];
    }
    model_identity(out[0]);
    align3(X, Y, R, out[0].t);
    R_to_quat(R, out[0].q);
    out[0].scale = sc; out[0].f1 = f1; out[0].f2 = f2;
    return 1;
}

// a-5: shared focal.  Unknowns w = 1/f^2, sigma = s^2, rho = depth of point 2 in image 2 over s (d2[2] unused).
//   N(w) = sigma D(w) ; L02(w) = sigma (a0 - 2 rho c0 + rho^2 e) ; L12(w) = sigma (a1 - 2 rho c1 + rho^2 e)
// -> quintic in w with zero constant term (w = 0 is f = inf) -> quartic.  Kept: w, sigma, rho > 0.
// Linear polynomials are stored as {c0, c1}; products are expanded by hand (degrees <= 5).
MDRP_HD int solver_shared(const Sample3 &s, Model out[4]) {
    double P1[3], Q1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = (k == 2) ? 1 : 0, j = (k == 0) ? 1 : 2;
        const double ax = s.d1[i] * s.x1[i][0] - s.d1[j] * s.x1[j][0], ay = s.d1[i] * s.x1[i][1] - s.d1[j] * s.x1[j][1];
        P1[k] = ax * ax + ay * ay;
        Q1[k] = (s.d1[i] - s.d1[j]) * (s.d1[i] - s.d1[j]);
    }
    const double bx = s.d2[0] * s.x2[0][0] - s.d2[1] * s.x2[1][0], by = s.d2[0] * s.x2[0][1] - s.d2[1] * s.x2[1][1];
    const double Pp = bx * bx + by * by, Qp = (s.d2[0] - s.d2[1]) * (s.d2[0] - s.d2[1]);
    const double r0 = s.x2[0][0] * s.x2[0][0] + s.x2[0][1] * s.x2[0][1], r1 = s.x2[1][0] * s.x2[1][0] + s.x2[1][1] * s.x2[1][1],
                 r2 = s.x2[2][0] * s.x2[2][0] + s.x2[2][1] * s.x2[2][1];
    const double m02 = s.x2[0][0] * s.x2[2][0] + s.x2[0][1] * s.x2[2][1], m12 = s.x2[1][0] * s.x2[2][0] + s.x2[1][1] * s.x2[2][1];
    // linear polys (c0 + c1 w)
    const double N0 = Q1[0], N1 = P1[0], D0 = Qp, D1 = Pp;
    const double a00 = s.d2[0] * s.d2[0], a01 = a00 * r0, a10 = s.d2[1] * s.d2[1], a11 = a10 * r1;
    const double c00 = s.d2[0], c01 = s.d2[0] * m02, c10 = s.d2[1], c11 = s.d2[1] * m12;
    const double e0 = 1.0, e1 = r2;
    const double L020 = Q1[1], L021 = P1[1], L120 = Q1[2], L121 = P1[2];
    const double dc0 = c00 - c10, dc1 = c01 - c11;
    const double da0 = a00 - a10, da1 = a01 - a11;
    const double dL0 = L020 - L120, dL1 = L021 - L121;
    // U = N*(a0-a1) - (L02-L12)*D   (degree 2)
    const double U0 = N0 * da0 - dL0 * D0;
    const double U1 = N0 * da1 + N1 * da0 - dL0 * D1 - dL1 * D0;
    const double U2 = N1 * da1 - dL1 * D1;
    // G = L02*D - N*a0 (degree 2)
    const double G0 = L020 * D0 - N0 * a00;
    const double G1 = L020 * D1 + L021 * D0 - N0 * a01 - N1 * a00;
    const double G2 = L021 * D1 - N1 * a01;
    // H = N*dc (degree 2)
    const double H0 = N0 * dc0, H1 = N0 * dc1 + N1 * dc0, H2 = N1 * dc1;
    // T1 = 4 * (H*dc) * G : (H*dc) degree 3
    const double K0 = H0 * dc0, K1 = H0 * dc1 + H1 * dc0, K2 = H1 * dc1 + H2 * dc0, K3 = H2 * dc1;
    // T2 = 4 * c0 * H * U : (c0*H) degree 3
    const double W0 = c00 * H0, W1 = c00 * H1 + c01 * H0, W2 = c00 * H2 + c01 * H1, W3 = c01 * H2;
    // U^2 degree 4
    const double V0 = U0 * U0, V1 = 2 * U0 * U1, V2 = 2 * U0 * U2 + U1 * U1, V3 = 2 * U1 * U2, V4 = U2 * U2;
    // q5 = 4 K G + 4 W U - e V   (coefficients 1..5; coefficient 0 vanishes identically)
    const double q1 = 4 * (K0 * G1 + K1 * G0) + 4 * (W0 * U1 + W1 * U0) - (e0 * V1 + e1 * V0);
    const double q2 = 4 * (K0 * G2 + K1 * G1 + K2 * G0) + 4 * (W0 * U2 + W1 * U1 + W2 * U0) - (e0 * V2 + e1 * V1);
    const double q3 = 4 * (K1 * G2 + K2 * G1 + K3 * G0) + 4 * (W1 * U2 + W2 * U1 + W3 * U0) - (e0 * V3 + e1 * V2);
    const double q4 = 4 * (K2 * G2 + K3 * G1) + 4 * (W2 * U2 + W3 * U1) - (e0 * V4 + e1 * V3);
    const double q5 = 4 * (K3 * G2) + 4 * (W3 * U2) - (e1 * V4);
    if (!(fabs(q5) > 0)) return 0;
    const double iq = sv_rcp(q5);
    double ws[4];
    const int mask = solve_quartic_real(q4 * iq, q3 * iq, q2 * iq, q1 * iq, ws);
    int n = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (!((mask >> r) & 1)) continue;
        double w = ws[r];
        if (!(w > 0)) continue;
        double Nw = N0 + N1 * w, Dw = D0 + D1 * w;
        double sig = sv_div(Nw, Dw);
        if (!(sig > 0)) continue;
        double rho = sv_div(U0 + w * (U1 + w * U2), 2.0 * Nw * (dc0 + dc1 * w));
        for (int it = 0; it < 4; ++it) { // Newton polish of (w, sigma, rho)
            const double ga0 = a00 + a01 * w, ga1 = a10 + a11 * w, gc0 = c00 + c01 * w, gc1 = c10 + c11 * w, ge = e0 + e1 * w;
            const double h0 = ga0 - 2 * rho * gc0 + rho * rho * ge, h1 = ga1 - 2 * rho * gc1 + rho * rho * ge;
            const double dh0 = a01 - 2 * rho * c01 + rho * rho * e1, dh1 = a11 - 2 * rho * c11 + rho * rho * e1;
            Nw = N0 + N1 * w; Dw = D0 + D1 * w;
            const double res[3] = {Nw - sig * Dw, (L020 + L021 * w) - sig * h0, (L120 + L121 * w) - sig * h1};
            const double J[9] = {N1 - sig * D1, -Dw, 0.0,
                                 L021 - sig * dh0, -h0, -sig * (-2 * gc0 + 2 * rho * ge),
                                 L121 - sig * dh1, -h1, -sig * (-2 * gc1 + 2 * rho * ge)};
            double dx[3];
            if (!solve3x3(J, res, dx)) break;
            w -= dx[0]; sig -= dx[1]; rho -= dx[2];
            if (fabs(dx[0]) + fabs(dx[1]) + fabs(dx[2]) < 1e-15 * (fabs(w) + fabs(sig) + fabs(rho))) break;
        }
        if (!(w > 0 && sig > 0 && rho > 0)) continue;
        const double f = sv_rsqrt(w), sc = sv_sqrt(sig), lam2 = rho * sc;
        const double invf = sv_rcp(f);
        double X[9], Y[9], R[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double dy = (i < 2) ? sc * s.d2[i] : lam2;
            X[3 * i] = s.d1[i] * s.x1[i][0] * invf; X[3 * i + 1] = s.d1[i] * s.x1[i][1] * invf; X[3 * i + 2] = s.d1[i];
            Y[3 * i] = dy * s.x2[i][0] * invf; Y[3 * i + 1] = dy * s.x2[i][1] * invf; Y[3 * i + 2] = dy;
        }
        Model m;
        model_identity(m);
        align3(X, Y, R, m.t);
        R_to_quat(R, m.q);
        m.scale = sc; m.f1 = f; m.f2 = f;
        if (n == 0) out[0] = m; else if (n == 1) out[1] = m; else if (n == 2) out[2] = m; else out[3] = m;
        ++n;
    }
    return n;
}

enum { SOLVER_P3P = 0, SOLVER_SHIFT = 1, SOLVER_SHARED = 2, SOLVER_VARYING = 3 };

MDRP_HD int run_solver(int solver, const Sample3 &s, Model out[4]) {
    switch (solver) {
    case SOLVER_P3P: return solver_calib_p3p(s, out);
    case SOLVER_SHIFT: return solver_calib_shift(s, out);
    case SOLVER_SHARED: return solver_shared(s, out);
    default: return solver_varying(s, out);
    }
}

// Reciprocal and reciprocal square root of the LM sweeps.  The oracle (and the host build of this header) divide; on the device an IEEE
// fp64 division is an 11-instruction dependent chain (v_div_scale x 2, v_rcp, four FMAs, v_div_fmas, v_div_fixup) and `1 / sqrt(x)` a
// square root followed by one — three of them per correspondence and sweep, ~18 % of the cost sweep's instructions.  The device takes the
// hardware seed and two Newton steps: <= 1 ulp instead of correctly rounded, the same class of difference as the FMA contraction the
// device build already has against the oracle (gated by the 3 x 1024-pair fixtures of tests/test_gpu_headline.py: masks and inlier
// counts identical, models to 1e-8).  x = 0 gives NaN where the division gives inf; both end as "term truncated / row of weight zero".
MDRP_HD double lm_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
#else
    return 1.0 / x;
#endif
}
MDRP_HD double lm_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * fma(-h * y, y, 1.5);
    return y * fma(-h * y, y, 1.5);
#else
    return 1.0 / sqrt(x);
#endif
}

// ---------------------------------------------------------------- robust losses (a-8)
MDRP_HD double loss_value(int type, double thr, double r2) {
    const double t2 = thr * thr;
    switch (type) {
    // the truncated losses are std::min(r2, t2) = (t2 < r2) ? t2 : r2 in the reference: a NaN residual makes the COST NaN (no LM step is ever accepted)
    case 1: case 5: return t2 < r2 ? t2 : r2;
    case 2: { const double r = sqrt(r2); return r <= thr ? r2 : thr * (2.0 * r - thr); }
#if defined(__HIP_DEVICE_COMPILE__)
    case 3: return t2 * log1p(r2 * lm_rcp(t2));
    case 4: return t2 * log1p((t2 < r2 ? t2 : r2) * lm_rcp(t2));
#else
    case 3: return t2 * log1p(r2 / t2);
    case 4: return t2 * log1p((t2 < r2 ? t2 : r2) / t2);
#endif
    default: return r2;
    }
}
#if defined(__HIPCC__)
// log(1 + x), x >= 0, from a 128-entry table in LDS (mdrp_logtab.h: per mantissa interval of width 1 / 128 the double inv = fl(1 / c), c its
// midpoint, and -ln(inv)).  1 + x = 2^e m, m in [1, 2);  r = m inv - 1 (one FMA: |r| <= 1 / 256, exact to half an ulp of r);
// log(1 + x) = e ln 2 - ln(inv) + log1p(r), log1p(r) by its series to r^6 (truncation r^7 / 7 < 2e-18).  The rounding of 1 + x is put back
// to first order (err / (1 + x), err = x - ((1 + x) - 1)).  ~22 instructions against ~150 of the library's log1p: the Cauchy losses take
// three per record in every cost sweep of the final refinements (round 5).  Non-finite arguments come back as they are (inf, NaN).
// (Declared in the host pass too — templates that call it are parsed there — with the device body only in the device pass.)
__device__ __forceinline__ double lm_log1p(double x, const double *tab_lds) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    const double y = 1.0 + x;
    const double err = x - (y - 1.0);
    const unsigned hi = (unsigned)__double2hiint(y);
    const int e = (int)(hi >> 20) - 1023;
    const unsigned idx = (hi >> 13) & 127u;
    const double m = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), __double2loint(y));
    lds_cdouble *t = (lds_cdouble *)tab_lds + 2 * idx;
    const double inv_c = t[0], log_c = t[1];
    const double r = fma(m, inv_c, -1.0);
    double q = fma(r, -1.0 / 6.0, 0.2);
    q = fma(r, q, -0.25);
    q = fma(r, q, 1.0 / 3.0);
    q = fma(r, q, -0.5);
    const double lp = fma(r * r, q, r) + ldexp(err * inv_c, -e);
    const double v = fma((double)e, 0.69314718055994530942, log_c + lp);
    return y < __builtin_huge_val() ? v : y;
#else
    (void)tab_lds;
    return log1p(x);
#endif
}
// loss_value with the table for the two Cauchy losses (tab == nullptr: the plain function)
__device__ __forceinline__ double loss_value_tab(int type, double thr, double r2, const double *tab_lds) {
    if (!tab_lds || (type != 3 && type != 4)) return loss_value(type, thr, r2);
    const double t2 = thr * thr;
    const double x = (type == 4 ? (t2 < r2 ? t2 : r2) : r2) * lm_rcp(t2);
    return t2 * lm_log1p(x, tab_lds);
}
#endif
// mu: penalty strength of TRUNCATED_LE_ZACH (Le & Zach, 3DV 2021): 0.5, multiplied by 1.5 after every LM iteration
// (the reference's per-iteration callback; pinned against the binary to 1e-15 over 5 iterations)
MDRP_HD double loss_weight(int type, double thr, double r2, double mu = 0.5) {
    const double t2 = thr * thr;
    const double dmin = 2.2250738585072014e-308;
    switch (type) {
    case 1: return r2 < t2 ? 1.0 : 0.0;
    case 5: {
        const double r2h = r2 / t2;
        if (r2h < 1.0) return 0.5;
        const double r2m1 = r2h - 1.0;
        const double rho = (2.0 * r2m1 + sqrt(4.0 * r2m1 * r2m1 * mu * mu + 2.0 * mu * r2m1)) / mu;
        const double a = (r2h + mu * rho - 0.5 * rho) / (1.0 + mu * rho);
        const double zbar = a < 0.0 ? 0.0 : (a > 1.0 ? 1.0 : a);
        return (1.0 - zbar) / rho;
    }
    case 2: { const double r = sqrt(r2); return r <= thr ? 1.0 : thr / r; }
#if defined(__HIP_DEVICE_COMPILE__)
    case 3: { const double w = lm_rcp(1.0 + r2 * lm_rcp(t2)); return w > dmin ? w : dmin; }
    case 4: { if (!(r2 < t2)) return 0.0; const double w = lm_rcp(1.0 + r2 * lm_rcp(t2)); return w > dmin ? w : dmin; }
#else
    case 3: { const double w = 1.0 / (1.0 + r2 / t2); return w > dmin ? w : dmin; }
    case 4: { if (!(r2 < t2)) return 0.0; const double w = 1.0 / (1.0 + r2 / t2); return w > dmin ? w : dmin; }
#endif
    default: return 1.0;
    }
}

// ---------------------------------------------------------------- conservative fp32 pre-filter of the scoring sweep
// Exact inlier test of compute_sampson_msac_score: C^2 < thr * den, C = x2' E x1, den = |(E x1)_xy|^2 + |(E' x2)_xy|^2.
// With Dmax >= den for every record inside the pair's coordinate box (|x1| <= (ax, ay), |x2| <= (cx, cy)):
//     inlier  =>  |C| < T := sqrt(thr * Dmax).
// In fp32 (E and the coordinates rounded once, 8 FMAs, depth 7):  |C32 - C| <= 7u * M,  u = 2^-24,
// M = sum |E_ij| |x2_i|max |x1_j|max.  filter_keeps() drops a record only if |C32| > tb with
// tb = (T + 2e-6 M)(1 + 1e-6) + 1e-30 (5x margin on 7u = 4.2e-7; the absolute term covers flushed denormals).
// NaN keeps; M >= 1e30 or non-finite (fp32 overflow possible) -> tb = inf, keeps everything.
// thr_dmax = Dmax * (1 + 1e-9) is returned for the fp64 variant of the same bound.
MDRP_HD void filter_setup(const double E[9], const double box[4], double thr, float Ef[9], float &tb, double &thr_dmax) {
    const double ax = box[0], ay = box[1], cx = box[2], cy = box[3];
    const double e0 = fabs(E[0]) * ax + fabs(E[1]) * ay + fabs(E[2]), e1 = fabs(E[3]) * ax + fabs(E[4]) * ay + fabs(E[5]);
    const double g0 = fabs(E[0]) * cx + fabs(E[3]) * cy + fabs(E[6]), g1 = fabs(E[1]) * cx + fabs(E[4]) * cy + fabs(E[7]);
    thr_dmax = (1.0 + 1e-9) * (e0 * e0 + e1 * e1 + g0 * g0 + g1 * g1);
    const double e2 = fabs(E[6]) * ax + fabs(E[7]) * ay + fabs(E[8]);
    const double M = e0 * cx + e1 * cy + e2;
    const double T = sqrt(thr * (1.0 + 1e-12) * thr_dmax);
    tb = (M < 1e30) ? (float)((T + 2e-6 * M) * (1.0 + 1e-6)) + 1e-30f : __builtin_inff(); // !(M < 1e30) also catches NaN
    for (int i = 0; i < 9; ++i) Ef[i] = (float)E[i];
}

// phase 1 for one record (a, b) = x1, (c, d) = x2, all already rounded to fp32; the sweep runs the same arithmetic two
// records at a time (v_pk_fma_f32)
MDRP_HD bool filter_keeps(const float Ef[9], float tb, float a, float b, float c, float d) {
    const float e0 = fmaf(Ef[0], a, fmaf(Ef[1], b, Ef[2]));
    const float e1 = fmaf(Ef[3], a, fmaf(Ef[4], b, Ef[5]));
    const float e2 = fmaf(Ef[6], a, fmaf(Ef[7], b, Ef[8]));
    const float C = fmaf(c, e0, fmaf(d, e1, e2));
    return !(fabsf(C) > tb);
}

// ---------------------------------------------------------------- candidate COUNT on the matrix cores (k_count)
// The numerator of the Sampson test is bilinear in the correspondence: with x1 = (a, b, 1), x2 = (c, d, 1)
//     C = x2' E x1 = E0 ac + E1 bc + E2 c + E3 ad + E4 bd + E5 d + E6 a + E7 b + E8,
// a contraction over 8 monomials + 1: [models x 9] . [9 x correspondences] — this one piece of the path IS a dense
// matrix product, so it runs on MFMA (v_mfma_f32_16x16x32_bf16).  bf16 keeps 8 significand bits, so every coefficient and
// every monomial is split into bf16 parts x = xh + xl (+ xl2) and the K = 32 slots of one instruction hold
//     k  0.. 7: Eh_j * mh_j      k  8..15: Eh_j * ml_j      k 16..23: El_j * mh_j      k 24..26: (E8h, E8l, E8l2) * 1.
// Error against the exact C (u = 2^-9 per bf16 rounding, two-stage split exact to u^2 = 2^-18):
//     dropped El_j ml_j and the two split remainders: <= 3.02 * 2^-18 * sum_j |E_j m_j|    = 1.16e-5 * A
//     fp32 accumulation of 27 exact products inside the MFMA (<= 1 ulp per add assumed):   <= 27 * 2^-23 * A = 3.2e-6 * A
// with A <= M = sum |E_ij| |x2_i|max |x1_j|max over the pair's coordinate box.  KAPPA carries a 4x margin on the sum
// (tests/test_hostmath.py emulates the arithmetic; the GPU test measures the real instruction against fp64).
// A correspondence is a CANDIDATE iff |C_mfma| <= tb, tb = (T + KAPPA M)(1 + 1e-6), T = sqrt(thr Dmax) as in filter_setup.
// The kernel tests the sign of tb^2 - C^2 with ONE packed FMA whose clamp modifier turns it into 0 / 1 directly: the model's
// coefficients are pre-scaled by S = 2^k (exact in bf16) so that tb S lies in [2^20, 2^21); then (tb S)^2 - (C S)^2 is either
// <= 0 (definite outlier -> 0) or >= one fp32 ulp at 2^40, far above 1 (candidate -> 1), and the sum of the clamped values is
// the candidate count.  M S <= 2^21 / KAPPA < 2^36, so nothing overflows; models the filter cannot judge (non-finite, or
// beyond the fp32 range) get zero coefficients and tb^2 = 1: every correspondence stays a candidate.
constexpr double COUNT_KAPPA = 6e-5;

MDRP_HD uint16_t bf16_bits(float x) { // round to nearest even; NaN stays NaN
    uint32_t u;
    __builtin_memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
MDRP_HD float bf16_value(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
// x = h + l (+ l2), each a bf16
MDRP_HD void bf16_split(double x, uint16_t &h, uint16_t &l) {
    h = bf16_bits((float)x);
    l = bf16_bits((float)(x - (double)bf16_value(h)));
}
MDRP_HD void bf16_split3(double x, uint16_t &h, uint16_t &l, uint16_t &l2) {
    bf16_split(x, h, l);
    l2 = bf16_bits((float)(x - (double)bf16_value(h) - (double)bf16_value(l)));
}

// the 8 monomials of one correspondence in coefficient order E0..E7
MDRP_HD void count_monomials(double a, double b, double c, double d, double m[8]) {
    m[0] = a * c; m[1] = b * c; m[2] = c; m[3] = a * d; m[4] = b * d; m[5] = d; m[6] = a; m[7] = b;
}

// model side of the contraction: eh[8], el[8], e8[3] (bf16 bit patterns) and the outlier threshold tb
MDRP_HD void count_setup(const double E[9], const double box[4], double thr, uint16_t eh[8], uint16_t el[8], uint16_t e8[3], float &tb) {
    const double ax = box[0], ay = box[1], cx = box[2], cy = box[3];
    const double e0 = fabs(E[0]) * ax + fabs(E[1]) * ay + fabs(E[2]), e1 = fabs(E[3]) * ax + fabs(E[4]) * ay + fabs(E[5]);
    const double g0 = fabs(E[0]) * cx + fabs(E[3]) * cy + fabs(E[6]), g1 = fabs(E[1]) * cx + fabs(E[4]) * cy + fabs(E[7]);
    const double dmax = (1.0 + 1e-9) * (e0 * e0 + e1 * e1 + g0 * g0 + g1 * g1);
    const double e2 = fabs(E[6]) * ax + fabs(E[7]) * ay + fabs(E[8]);
    const double M = e0 * cx + e1 * cy + e2;
    const double T = sqrt(thr * (1.0 + 1e-12) * dmax);
    tb = (M < 1e30) ? (float)((T + COUNT_KAPPA * M) * (1.0 + 1e-6)) + 1e-30f : __builtin_inff(); // !(M < 1e30) also catches NaN
    for (int j = 0; j < 8; ++j) bf16_split(E[j], eh[j], el[j]);
    bf16_split3(E[8], e8[0], e8[1], e8[2]);
}

// the same with the power-of-two scale of the clamp test folded in: coefficients of S E, tb2 = (tb S)^2 rounded up.
// This is k_count's per-hypothesis prologue, so it is kept cheap: the bounds M and Dmax in fp64 (they decide correctness),
// the threshold and the splits in fp32 — S E is rounded to fp32 once (2^-24 relative, a 250th of the 2^-18 split remainder
// that KAPPA already carries with a 4x margin) and split exactly from there; the fp32 roundings of tb (sqrt, add, two
// products: <= 5 * 2^-24) are covered by its (1 + 1e-6) factor.
MDRP_HD void bf16_split_f32(float x, uint16_t &h, uint16_t &l) {
    h = bf16_bits(x);
    l = bf16_bits(x - bf16_value(h)); // exact difference
}
MDRP_HD void count_setup_scaled(const double E[9], const double box[4], double thr, uint16_t eh[8], uint16_t el[8], uint16_t e8[3], float &tb2) {
    const double ax = box[0], ay = box[1], cx = box[2], cy = box[3];
    const double e0 = fabs(E[0]) * ax + fabs(E[1]) * ay + fabs(E[2]), e1 = fabs(E[3]) * ax + fabs(E[4]) * ay + fabs(E[5]);
    const double g0 = fabs(E[0]) * cx + fabs(E[3]) * cy + fabs(E[6]), g1 = fabs(E[1]) * cx + fabs(E[4]) * cy + fabs(E[7]);
    const double dmax = (1.0 + 1e-9) * (e0 * e0 + e1 * e1 + g0 * g0 + g1 * g1);
    const double e2 = fabs(E[6]) * ax + fabs(E[7]) * ay + fabs(E[8]);
    const double M = e0 * cx + e1 * cy + e2;
    const float td = (float)(thr * (1.0 + 1e-12) * dmax), km = (float)(COUNT_KAPPA * M);
    const float tb = (sqrtf(td) + km) * (1.0f + 1e-6f) + 1e-30f;
    for (int j = 0; j < 8; ++j) { eh[j] = 0; el[j] = 0; }
    e8[0] = e8[1] = e8[2] = 0;
    tb2 = 1.0f;
    // Not judgeable -> C = 0 against tb^2 = 1 keeps everything.  That includes models so small that thr * Dmax leaves the fp32
    // NORMAL range: td would flush to zero and tb = kappa M alone would be far below sqrt(thr Dmax) — an UNDERcount (found by
    // tests/test_gpu_adversarial.py with |F| = 1e-24; pose and unit-norm F models are 20 decades away from it).
    if (!(M < 1e30) || !(tb < 1e30f) || !(tb > 0.0f) || !(td > 1e-30f) || !(km > 1e-30f)) return;
    int ex;
    (void)frexpf(tb, &ex);            // tb = m 2^ex, m in [0.5, 1)
    const int k = 21 - ex;            // tb 2^k in [2^20, 2^21)
    const double S = ldexp(1.0, k);
    for (int j = 0; j < 8; ++j) bf16_split_f32((float)(E[j] * S), eh[j], el[j]);
    const float c8 = (float)(E[8] * S);
    bf16_split_f32(c8, e8[0], e8[1]);
    e8[2] = bf16_bits((c8 - bf16_value(e8[0])) - bf16_value(e8[1]));
    const float tbs = ldexpf(tb, k);
    tb2 = (tbs * tbs) * (1.0f + 2e-7f);
}

// ---------------------------------------------------------------- fp32 LOWER bound of the MSAC score (k_bound)
// compute_sampson_msac_score adds min(r^2, thr) per correspondence (cheirality failures add thr >= r^2), r^2 = C^2 / den.
// In fp32, with E and the coordinates rounded once (u = 2^-24) and the bounds of the pair's coordinate box:
//     |C32 - C|     <= 7u M            (8 FMAs, depth 7; M as in filter_setup)                 -> eC = 2e-6 M      (5x margin)
//     |den32 - den| <= 14u Dmax        (|e32 - e| <= 5u e_bar per linear form, squares, 4 adds) -> eD = 4e-6 Dmax   (5x margin)
// so  r^2 >= max(|C32| - eC, 0)^2 / (den32 + eD)  =: q, and the sum over the correspondences of min(q, thr_down) is a lower
// bound of the exact score up to the fp32 roundings of q itself (<= 8u relative per term) and of the summation (partial
// sums of 64 terms in fp32, flushed to fp64: <= 64u relative); BOUND_SLACK covers both with a 4x margin.
// A hypothesis whose lower bound is no better than a record score — and whose candidate count (k_count) is no better than
// the record count — cannot break a record, whatever its exact score is.  Models outside the fp32 range (setup returns
// false) are never judged by the bound; a NaN / inf correspondence yields a NaN quotient, which adds thr like the reference's
// `r2 < thr` test does.
constexpr double BOUND_SLACK = 2e-5;

MDRP_HD bool bound_setup32(const double E[9], const double box[4], double thr, float Ef[9], float &eC, float &eD, float &thr_dn) {
    const double ax = box[0], ay = box[1], cx = box[2], cy = box[3];
    const double e0 = fabs(E[0]) * ax + fabs(E[1]) * ay + fabs(E[2]), e1 = fabs(E[3]) * ax + fabs(E[4]) * ay + fabs(E[5]);
    const double g0 = fabs(E[0]) * cx + fabs(E[3]) * cy + fabs(E[6]), g1 = fabs(E[1]) * cx + fabs(E[4]) * cy + fabs(E[7]);
    const double dmax = e0 * e0 + e1 * e1 + g0 * g0 + g1 * g1;
    const double e2 = fabs(E[6]) * ax + fabs(E[7]) * ay + fabs(E[8]);
    const double M = e0 * cx + e1 * cy + e2;
    const bool sane = (M < 1e15) && (dmax < 1e30) && (dmax > 1e-30); // fp32 range: squares must neither overflow nor vanish
    eC = sane ? (float)(2e-6 * M * (1.0 + 1e-6)) + 1e-30f : 0.0f;
    eD = sane ? (float)(4e-6 * dmax * (1.0 + 1e-6)) + 1e-37f : 1.0f;
    float t = (float)thr;
    if ((double)t > thr && t > 0.0f) { // next float towards zero
        uint32_t bits;
        __builtin_memcpy(&bits, &t, 4);
        --bits;
        __builtin_memcpy(&t, &bits, 4);
    }
    thr_dn = t;
    for (int i = 0; i < 9; ++i) Ef[i] = (float)E[i];
    return sane;
}

// lower bound q of r^2 for one correspondence, everything fp32 (the MSAC term is min(q, thr_dn); NaN for a NaN / inf
// correspondence, which the callers turn into thr as the reference's `r2 < thr` test does)
MDRP_HD float bound_r2_32(const float Ef[9], float eC, float eD, float a, float b, float c, float d) {
    const float e0 = fmaf(Ef[0], a, fmaf(Ef[1], b, Ef[2]));
    const float e1 = fmaf(Ef[3], a, fmaf(Ef[4], b, Ef[5]));
    const float e2 = fmaf(Ef[6], a, fmaf(Ef[7], b, Ef[8]));
    const float g0 = fmaf(Ef[0], c, fmaf(Ef[3], d, Ef[6]));
    const float g1 = fmaf(Ef[1], c, fmaf(Ef[4], d, Ef[7]));
    const float C = fmaf(c, e0, fmaf(d, e1, e2));
    const float den = fmaf(e0, e0, fmaf(e1, e1, fmaf(g0, g0, fmaf(g1, g1, eD))));
    const float t = fmaxf(fabsf(C) - eC, 0.0f);
#if defined(__HIP_DEVICE_COMPILE__)
    return (t * t) * __builtin_amdgcn_rcpf(den); // v_rcp_f32: 1 ulp, inside the 8u budget of BOUND_SLACK (an IEEE division is ~10 instructions)
#else
    return (t * t) / den;
#endif
}

// ---------------------------------------------------------------- refinement: per-correspondence residuals
// State of one hypothesis during LM, expanded once per cost/accumulate pass.
struct LmState {
    double R[9], t[3], s, u, v, f1, f2;
    double E[9], F[9];
    double if1, if2; // 1 / f1, 1 / f2: the same IEEE quotients the reprojection terms used to form per correspondence (round 4: once per state)
};

MDRP_HD void lm_state_from_model(const Model &m, bool focal, LmState &st) {
    quat_to_R(m.q, st.R);
    st.t[0] = m.t[0]; st.t[1] = m.t[1]; st.t[2] = m.t[2];
    st.s = m.scale; st.u = m.shift1; st.v = m.shift2;
    st.f1 = focal ? m.f1 : 1.0; st.f2 = focal ? m.f2 : 1.0;
    st.if1 = 1.0 / st.f1; st.if2 = 1.0 / st.f2;
    essential_from_Rt(st.R, st.t, st.E);
    fundamental_from_E(st.E, st.f1, st.f2, st.F);
}

constexpr int LM_NPAR = 11; // rot(3) t(3) s u v f1 f2

// residuals r[0..4] = {sampson, fwd.x, fwd.y, bwd.x, bwd.y} (reprojection ones times sqrt(sr)); zf / zb = depth of
// the forward / backward transferred point (terms with negative depth are skipped by the callers).
// WITH_J: Jacobian rows J[LM_NPAR] with R <- R exp([w]x), t <- t + dt, s <- s + ds.
// The three terms are separate functions so that the GPU accumulate sweep can consume each term's rows before it
// computes the next term (register pressure); point_residuals() below strings them together.
// Calibrated estimator (FOCAL = false): f1 = f2 = 1 folds away at compile time (F == E, no focal columns).
template <bool WITH_J, bool FOCAL>
MDRP_HD void lm_sampson_term(const LmState &st, double x1x, double x1y, double x2x, double x2y, double &r0, double *J0) {
    const double *R = st.R, *F = FOCAL ? st.F : st.E, *E = st.E;
    const double f1 = FOCAL ? st.f1 : 1.0, f2 = FOCAL ? st.f2 : 1.0;
    const double Fh1_0 = F[0] * x1x + F[1] * x1y + F[2], Fh1_1 = F[3] * x1x + F[4] * x1y + F[5], Fh1_2 = F[6] * x1x + F[7] * x1y + F[8];
    const double Ft2_0 = F[0] * x2x + F[3] * x2y + F[6], Ft2_1 = F[1] * x2x + F[4] * x2y + F[7];
    const double C = x2x * Fh1_0 + x2y * Fh1_1 + Fh1_2;
    const double den = Fh1_0 * Fh1_0 + Fh1_1 * Fh1_1 + Ft2_0 * Ft2_0 + Ft2_1 * Ft2_1;
    const double isd = lm_rsqrt(den);
    r0 = C * isd;
    if (!WITH_J) return;
    // G = d r0 / d F, then chain to E, rotation (post), translation, focals
    const double h1[3] = {x1x, x1y, 1.0}, h2[3] = {x2x, x2y, 1.0};
    const double Fh1[3] = {Fh1_0, Fh1_1, Fh1_2}, Ft2[3] = {Ft2_0, Ft2_1, 0.0};
    const double k = C * isd * isd * isd;
    double GE[9]; // d r0 / d E_ij
    double G[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double g = h2[i] * h1[j] * isd;
            if (i < 2) g -= k * Fh1[i] * h1[j];
            if (j < 2) g -= k * Ft2[j] * h2[i];
            G[3 * i + j] = g;
            GE[3 * i + j] = g * (i == 2 ? f2 : 1.0) * (j == 2 ? f1 : 1.0);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int b = (a + 1) % 3, c = (a + 2) % 3;
        // dE/dw_a = E [e_a]x : column b = +E[:,c], column c = -E[:,b], column a = 0
        double acc = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) acc += GE[3 * i + b] * E[3 * i + c] - GE[3 * i + c] * E[3 * i + b];
        J0[a] = acc;
        // dE/dt_a = [e_a]x R : row b = -R row c, row c = +R row b
        double acc2 = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc2 += -GE[3 * b + j] * R[3 * c + j] + GE[3 * c + j] * R[3 * b + j];
        J0[3 + a] = acc2;
    }
    J0[6] = 0; J0[7] = 0; J0[8] = 0;
    J0[9] = G[2] * E[2] + G[5] * E[5] + G[8] * E[8] * f2;
    J0[10] = G[6] * E[6] + G[7] * E[7] + G[8] * E[8] * f1;
}

// forward reprojection: Z = R X1 + t, X1 = (d1 + u) (x1 / f1, 1);  r = sqrt_sr (f2 Z.xy / Z.z - x2)
template <bool WITH_J, bool FOCAL>
MDRP_HD void lm_forward_term(const LmState &st, double sqrt_sr, double x1x, double x1y, double x2x, double x2y, double d1,
                             double &r1, double &r2, double &zf, double *J1, double *J2) {
    const double *R = st.R, *t = st.t;
    const double f2 = FOCAL ? st.f2 : 1.0;
    const double if1 = FOCAL ? st.if1 : 1.0;
    const double b1x = x1x * if1, b1y = x1y * if1;
    const double dd1 = d1 + st.u;
    const double X1[3] = {dd1 * b1x, dd1 * b1y, dd1};
    const double Z0 = R[0] * X1[0] + R[1] * X1[1] + R[2] * X1[2] + t[0];
    const double Z1 = R[3] * X1[0] + R[4] * X1[1] + R[5] * X1[2] + t[1];
    const double Z2 = R[6] * X1[0] + R[7] * X1[1] + R[8] * X1[2] + t[2];
    const double iz = lm_rcp(Z2);
    r1 = sqrt_sr * (f2 * Z0 * iz - x2x);
    r2 = sqrt_sr * (f2 * Z1 * iz - x2y);
    zf = Z2;
    if (!WITH_J) return;
    const double px0 = sqrt_sr * f2 * iz, pz0 = -sqrt_sr * f2 * Z0 * iz * iz, pz1 = -sqrt_sr * f2 * Z1 * iz * iz;
    // dZ/dw_a = R (e_a x X1)
    const double cr[3][3] = {{0.0, -X1[2], X1[1]}, {X1[2], 0.0, -X1[0]}, {-X1[1], X1[0], 0.0}}; // e_a x X1
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double v0 = R[0] * cr[a][0] + R[1] * cr[a][1] + R[2] * cr[a][2];
        const double v1 = R[3] * cr[a][0] + R[4] * cr[a][1] + R[5] * cr[a][2];
        const double v2 = R[6] * cr[a][0] + R[7] * cr[a][1] + R[8] * cr[a][2];
        J1[a] = px0 * v0 + pz0 * v2;
        J2[a] = px0 * v1 + pz1 * v2;
    }
    J1[3] = px0; J1[4] = 0.0; J1[5] = pz0;
    J2[3] = 0.0; J2[4] = px0; J2[5] = pz1;
    J1[6] = 0.0; J2[6] = 0.0;
    { // shift1: dZ = R b1
        const double v0 = R[0] * b1x + R[1] * b1y + R[2], v1 = R[3] * b1x + R[4] * b1y + R[5], v2 = R[6] * b1x + R[7] * b1y + R[8];
        J1[7] = px0 * v0 + pz0 * v2;
        J2[7] = px0 * v1 + pz1 * v2;
    }
    J1[8] = 0.0; J2[8] = 0.0;
    { // f1: dX1 = -dd1 (b1x, b1y, 0) / f1
        const double ex = -dd1 * b1x * if1, ey = -dd1 * b1y * if1;
        const double v0 = R[0] * ex + R[1] * ey, v1 = R[3] * ex + R[4] * ey, v2 = R[6] * ex + R[7] * ey;
        J1[9] = px0 * v0 + pz0 * v2;
        J2[9] = px0 * v1 + pz1 * v2;
    }
    J1[10] = sqrt_sr * Z0 * iz;
    J2[10] = sqrt_sr * Z1 * iz;
}

// backward reprojection: W = R'(X2 - t), X2 = s (d2 + v) (x2 / f2, 1);  r = sqrt_sr (f1 W.xy / W.z - x1)
template <bool WITH_J, bool FOCAL>
MDRP_HD void lm_backward_term(const LmState &st, double sqrt_sr, double x1x, double x1y, double x2x, double x2y, double d2,
                              double &r3, double &r4, double &zb, double *J3, double *J4) {
    const double *R = st.R, *t = st.t;
    const double f1 = FOCAL ? st.f1 : 1.0;
    const double if2 = FOCAL ? st.if2 : 1.0;
    const double b2x = x2x * if2, b2y = x2y * if2;
    const double dd2 = d2 + st.v;
    const double sd = st.s * dd2;
    const double Y[3] = {sd * b2x - t[0], sd * b2y - t[1], sd - t[2]};
    const double W0 = R[0] * Y[0] + R[3] * Y[1] + R[6] * Y[2];
    const double W1 = R[1] * Y[0] + R[4] * Y[1] + R[7] * Y[2];
    const double W2 = R[2] * Y[0] + R[5] * Y[1] + R[8] * Y[2];
    const double iw = lm_rcp(W2);
    r3 = sqrt_sr * (f1 * W0 * iw - x1x);
    r4 = sqrt_sr * (f1 * W1 * iw - x1y);
    zb = W2;
    if (!WITH_J) return;
    const double px0 = sqrt_sr * f1 * iw, pz0 = -sqrt_sr * f1 * W0 * iw * iw, pz1 = -sqrt_sr * f1 * W1 * iw * iw;
    // dW/dw_a = W x e_a
    const double wc[3][3] = {{0.0, W2, -W1}, {-W2, 0.0, W0}, {W1, -W0, 0.0}};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        J3[a] = px0 * wc[a][0] + pz0 * wc[a][2];
        J4[a] = px0 * wc[a][1] + pz1 * wc[a][2];
        // dW/dt_a = -R' e_a = -(row a of R)
        J3[3 + a] = -(px0 * R[3 * a + 0] + pz0 * R[3 * a + 2]);
        J4[3 + a] = -(px0 * R[3 * a + 1] + pz1 * R[3 * a + 2]);
    }
    { // scale: dW = R' (dd2 b2)
        const double y0 = dd2 * b2x, y1 = dd2 * b2y, y2 = dd2;
        const double v0 = R[0] * y0 + R[3] * y1 + R[6] * y2, v1 = R[1] * y0 + R[4] * y1 + R[7] * y2, v2 = R[2] * y0 + R[5] * y1 + R[8] * y2;
        J3[6] = px0 * v0 + pz0 * v2;
        J4[6] = px0 * v1 + pz1 * v2;
    }
    J3[7] = 0.0; J4[7] = 0.0;
    { // shift2: dW = R' (s b2)
        const double y0 = st.s * b2x, y1 = st.s * b2y, y2 = st.s;
        const double v0 = R[0] * y0 + R[3] * y1 + R[6] * y2, v1 = R[1] * y0 + R[4] * y1 + R[7] * y2, v2 = R[2] * y0 + R[5] * y1 + R[8] * y2;
        J3[8] = px0 * v0 + pz0 * v2;
        J4[8] = px0 * v1 + pz1 * v2;
    }
    J3[9] = sqrt_sr * W0 * iw;
    J4[9] = sqrt_sr * W1 * iw;
    { // f2: dX2 = -s dd2 (b2x, b2y, 0)/f2
        const double ex = -sd * b2x * if2, ey = -sd * b2y * if2;
        const double v0 = R[0] * ex + R[3] * ey, v1 = R[1] * ex + R[4] * ey, v2 = R[2] * ex + R[5] * ey;
        J3[10] = px0 * v0 + pz0 * v2;
        J4[10] = px0 * v1 + pz1 * v2;
    }
}

template <bool WITH_J, bool FOCAL = true>
MDRP_HD void point_residuals(const LmState &st, double sqrt_sr, double x1x, double x1y, double x2x, double x2y, double d1,
                             double d2, double r[5], double &zf, double &zb, double J[5][LM_NPAR]) {
    lm_sampson_term<WITH_J, FOCAL>(st, x1x, x1y, x2x, x2y, r[0], WITH_J ? J[0] : nullptr);
    lm_forward_term<WITH_J, FOCAL>(st, sqrt_sr, x1x, x1y, x2x, x2y, d1, r[1], r[2], zf, WITH_J ? J[1] : nullptr, WITH_J ? J[2] : nullptr);
    lm_backward_term<WITH_J, FOCAL>(st, sqrt_sr, x1x, x1y, x2x, x2y, d2, r[3], r[4], zb, WITH_J ? J[3] : nullptr, WITH_J ? J[4] : nullptr);
}

// parameter update of the refinement (lm step): R <- R exp([w]x), additive elsewhere; shifts are zeroed when they are
// not estimated (black-box behaviour of the reference's step()).  delta is in FULL 11-vector layout.
MDRP_HD void lm_apply_step(const Model &m, const double d[LM_NPAR], bool focal, bool est_shift, Model &o) {
    const double th2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], th = sqrt(th2);
    double re, im;
    if (th > 1e-6) { re = cos(0.5 * th); im = sin(0.5 * th) / th; }
    else { re = 1.0 - th2 / 8.0; im = 0.5 - th2 / 48.0; const double nq = 1.0 / sqrt(re * re + im * im * th2); re *= nq; im *= nq; }
    const double b0 = re, b1 = im * d[0], b2 = im * d[1], b3 = im * d[2];
    const double a0 = m.q[0], a1 = m.q[1], a2 = m.q[2], a3 = m.q[3];
    o.q[0] = a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3;
    o.q[1] = a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2;
    o.q[2] = a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1;
    o.q[3] = a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0;
    o.t[0] = m.t[0] + d[3]; o.t[1] = m.t[1] + d[4]; o.t[2] = m.t[2] + d[5];
    o.scale = m.scale + d[6];
    o.shift1 = est_shift ? m.shift1 + d[7] : 0.0;
    o.shift2 = est_shift ? m.shift2 + d[8] : 0.0;
    o.f1 = focal ? m.f1 + d[9] : m.f1;
    o.f2 = focal ? m.f2 + d[10] : m.f2;
}

// Cholesky solve of the (lower-stored, row-major n x n) damped normal equations; n <= 9.
// Device (round 4): one reciprocal square root per column (lm_rsqrt: hardware seed + two Newton steps) instead of a square root and
// N (N + 3) / 2 divisions by the diagonal — 35 division chains of 11 dependent instructions each for N = 7, on the serial path of
// every LM iteration.  L(j, j) is kept as s * rsqrt(s); every quotient by it becomes a product with rsqrt(s): <= 1 ulp each.
template <int N>
MDRP_HD void chol_solve(const double *A, const double *b, double *x) {
    double L[N * N];
#if defined(__HIP_DEVICE_COMPILE__)
    double inv[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = A[i * N + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i * N + k] * L[j * N + k];
            if (i == j) { inv[i] = lm_rsqrt(s); L[i * N + i] = s * inv[i]; }
            else L[i * N + j] = s * inv[j];
        }
    double y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i * N + k] * y[k];
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s -= L[k * N + i] * x[k];
        x[i] = s * inv[i];
    }
#else
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = A[i * N + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i * N + k] * L[j * N + k];
            if (i == j) L[i * N + i] = sqrt(s);
            else L[i * N + j] = s / L[j * N + j];
        }
    double y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i * N + k] * y[k];
        y[i] = s / L[i * N + i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s -= L[k * N + i] * x[k];
        x[i] = s / L[i * N + i];
    }
#endif
}

// ---------------------------------------------------------------- sampler (a-3)
MDRP_HD int32_t splitmix_int(uint64_t &state) {
    state += 0x9e3779b97f4a7c15ULL;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return (int32_t)(z ^ (z >> 31));
}
MDRP_HD void draw_sample3(uint64_t n, uint64_t &state, uint32_t &i0, uint32_t &i1, uint32_t &i2) {
    i0 = (uint32_t)((uint64_t)(int64_t)splitmix_int(state) % n);
    do { i1 = (uint32_t)((uint64_t)(int64_t)splitmix_int(state) % n); } while (i1 == i0);
    do { i2 = (uint32_t)((uint64_t)(int64_t)splitmix_int(state) % n); } while (i2 == i0 || i2 == i1);
}

// The first chunk of a run is scored exactly in full (nothing has set a bar yet); everything behind it meets the bar it leaves, and a pair whose
// first chunk holds no outlier-free sample has none.  With r the pair's inlier ratio and k its sample size: 6 / r^k iterations — six such samples
// expected — between 256 and 1024; 128 where (nearly) every sample is one.  Measured at 0 / 50 / 75 / 85 % outliers (mdrp_capi.hip run_pass).
MDRP_HD int32_t first_chunk_wish(double r, int k) {
    if (!(r > 0.05)) r = 0.05;
    double p = r;
    for (int i = 1; i < k; ++i) p *= r;
    const double want = 6.0 / p;
    return want <= 16.0 ? 128 : (want >= 1024.0 ? 1024 : (want <= 256.0 ? 256 : (int32_t)want));
}

} // namespace mdrp
