// mdrp_classic_math.h — minimal solvers of the non-monodepth baselines: 5-point relative pose and
// 7-point fundamental matrix, as the reference binary computes them (upstream PoseLib 2.0.5):
//   relpose_5pt @0x145900 / @0x14ae80   Nistér 2004: null space of the 5 epipolar constraints, 10 cubic constraints in
//                                       (x, y, z), Gauss-Jordan, det B(z) = 0 of degree 10, real roots in ascending order,
//                                       motion_from_essential @0x1dd540 with the cheirality of all five sample points
//   relpose_7pt @0x4ff2e0               null space, det(r N0 + N1) = 0, roots in descending order
// The null-space basis is the one Eigen's FullPivHouseholderQR gives (last columns of matrixQ()): the parametrisation, and
// with it the ORDER of the solutions — which the LO-RANSAC trajectory depends on — follows from it.
// Compiles for host and device (MDRP_HD) so that tests/hostmath can pin it against the oracle on the CPU.
#pragma once
#include "mdrp_math.h"

namespace mdrp {

constexpr int CLASSIC_RELPOSE = 3, CLASSIC_SHARED = 4, CLASSIC_FUND = 5; // == MDRP_RELPOSE_5PT / MDRP_SHARED_6PT / MDRP_FUNDAMENTAL_7PT

// ---------------------------------------------------------------- null space (Eigen::FullPivHouseholderQR + matrixQ)
// A: 9 x M column-major in caller-provided strided storage (element (r, c) at A[(9 c + r) as]; destroyed): the full pivoting
// addresses it with data-dependent row and column indices, so on the device it lives in LDS — everything else is statically
// indexed and stays in registers.  N: the last 9 - M columns of Q, each the column-major vec of a 3 x 3 matrix.
template <int M>
MDRP_HD void fullpiv_nullspace(double *A, int as, double *N /*[9 - M][9]*/) {
    constexpr int rows = 9, NN = 9 - M;
#define AQ(r, c) A[(9 * (c) + (r)) * as]
    double tau[M];
    int rt[M];
    double ess[M][9]; // Householder vectors (entries below the diagonal), kept for Q
    const double prec = 2.220446049250313e-16 * (double)M;
    double biggest = 0.0;
    bool degenerate = false;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        tau[k] = 0.0; rt[k] = k;
#pragma unroll
        for (int r = 0; r < 9; ++r) ess[k][r] = 0.0;
        if (degenerate) continue;
        int br = k, bc = k;
        double big = -1.0;
        for (int c = k; c < M; ++c)
            for (int r = k; r < rows; ++r) {
                const double v = fabs(AQ(r, c));
                if (v > big) { big = v; br = r; bc = c; }
            }
        if (k == 0) biggest = big;
        if (big <= biggest * prec) { degenerate = true; continue; }
        rt[k] = br;
        if (br != k)
            for (int c = k; c < M; ++c) { const double t = AQ(k, c); AQ(k, c) = AQ(br, c); AQ(br, c) = t; }
        if (bc != k)
            for (int r = 0; r < rows; ++r) { const double t = AQ(r, k); AQ(r, k) = AQ(r, bc); AQ(r, bc) = t; }
        double col[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) col[r] = (r >= k) ? AQ(r, k) : 0.0;
        double tail = 0.0;
#pragma unroll
        for (int r = 0; r < 9; ++r) if (r > k) tail += col[r] * col[r];
        const double c0 = col[k];
        double beta;
        if (tail <= 2.2250738585072014e-308) {
            beta = c0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            const double inv = 1.0 / (c0 - beta); // Eigen divides; the reciprocal differs by an ulp at most
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k) ess[k][r] = col[r] * inv;
            tau[k] = (beta - c0) / beta;
        }
        for (int c = k + 1; c < M; ++c) {
            double tmp = 0.0;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k) tmp += ess[k][r] * AQ(r, c);
            tmp += AQ(k, c);
            AQ(k, c) -= tau[k] * tmp;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k) AQ(r, c) -= tau[k] * ess[k][r] * tmp;
        }
    }
    // only the last NN columns of Q = H_0 P_0 ... H_{M-1} P_{M-1} I are needed: start from unit vectors e_M .. e_8
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        double q[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) q[r] = (r == M + j) ? 1.0 : 0.0;
#pragma unroll
        for (int k = M - 1; k >= 0; --k) {
            double tmp = q[k];
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k) tmp += ess[k][r] * q[r];
            q[k] -= tau[k] * tmp;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k) q[r] -= tau[k] * ess[k][r] * tmp;
            // swap q[k] <-> q[rt[k]] (rt[k] >= k is data dependent: select chain instead of a dynamic index)
            const double qk = q[k];
            double qr = qk;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k && r == rt[k]) qr = q[r];
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k && r == rt[k]) q[r] = qk;
            q[k] = qr;
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) N[9 * j + r] = q[r];
    }
#undef AQ
}

// The same factorisation of the 9 x M epipolar constraint matrix (column p = kron(x1[p], x2[p]): entry 3 j + i multiplies E(i, j)) with its
// last RC columns in registers and only the first M - RC in the caller's strided storage (element (r, c) at A[(9 c + r) as]).  The device keeps the
// matrix in LDS, one 64-lane column per element, and that allocation is what bounds the occupancy of the 7-point solver and of the 5-point null
// space kernel (63 columns = 32 KB: five wavefronts per CU; 36 = 18 KB: eight).  Only the two swaps of a step address the matrix with data-dependent
// indices: on the register columns they are select chains, between a stored and a register column an exchange.  Same operations on the same values in
// the same order as fullpiv_nullspace<M> behind epipolar_columns<M> (tests/hostmath pins both against the oracle).
template <int K> struct StepC { static constexpr int value = K; };
template <int M, int RC>
MDRP_HD void epipolar_nullspace(const double (*x1h)[3], const double (*x2h)[3], double *A, int as, double *N /*[9 - M][9]*/) {
    constexpr int NN = 9 - M, ML = M - RC;
    double rc[RC][9];
#define AQ(r, c) A[(9 * (c) + (r)) * as]
    auto get = [&](int r, int c) __attribute__((always_inline)) -> double { return c < ML ? AQ(r, c) : rc[c - ML][r]; }; // r, c: constants after unrolling
    auto set = [&](int r, int c, double v) __attribute__((always_inline)) { if (c < ML) AQ(r, c) = v; else rc[c - ML][r] = v; };
#pragma unroll
    for (int p = 0; p < M; ++p)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) set(3 * j + i, p, x1h[p][j] * x2h[p][i]);
    double tau[M];
    int rt[M];
    double ess[M][9];
    const double prec = 2.220446049250313e-16 * (double)M;
    double biggest = 0.0;
    bool degenerate = false;
    // one elimination step with the step number as a compile-time constant (a `#pragma unroll` over the steps is only a request: at M = 7 the body
    // is past the compiler's threshold, the loop stays rolled and every array it indexes with k moves to scratch memory)
    auto step = [&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        tau[k] = 0.0; rt[k] = k;
#pragma unroll
        for (int r = 0; r < 9; ++r) ess[k][r] = 0.0;
        if (degenerate) return;
        int br = k, bc = k;
        double big = -1.0;
#pragma unroll
        for (int c = k; c < M; ++c)
#pragma unroll
            for (int r = k; r < 9; ++r) {
                const double v = fabs(get(r, c));
                if (v > big) { big = v; br = r; bc = c; }
            }
        if (k == 0) biggest = big;
        if (big <= biggest * prec) { degenerate = true; return; }
        rt[k] = br;
        if (br != k) {
#pragma unroll
            for (int c = k; c < M; ++c) {
                if (c < ML) { const double t = AQ(k, c); AQ(k, c) = AQ(br, c); AQ(br, c) = t; }
                else {
                    const double top = rc[c - ML][k];
                    double low = top;
#pragma unroll
                    for (int r = 0; r < 9; ++r) if (r > k && r == br) { low = rc[c - ML][r]; rc[c - ML][r] = top; }
                    rc[c - ML][k] = low;
                }
            }
        }
        if (bc != k) {
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const double tk = get(r, k);
                double tb = tk;
                if (k < ML && bc < ML) { tb = AQ(r, bc); AQ(r, bc) = tk; } // both stored
                else {
#pragma unroll
                    for (int j = 0; j < RC; ++j) if (ML + j > k && ML + j == bc) { tb = rc[j][r]; rc[j][r] = tk; }
                }
                set(r, k, tb);
            }
        }
        double col[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) col[r] = (r >= k) ? get(r, k) : 0.0;
        double tail = 0.0;
#pragma unroll
        for (int r = 0; r < 9; ++r) if (r > k) tail += col[r] * col[r];
        const double c0 = col[k];
        double beta;
        if (tail <= 2.2250738585072014e-308) {
            beta = c0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            const double inv = 1.0 / (c0 - beta);
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k) ess[k][r] = col[r] * inv;
            tau[k] = (beta - c0) / beta;
        }
#pragma unroll
        for (int c = k + 1; c < M; ++c) {
            double tmp = 0.0;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k) tmp += ess[k][r] * get(r, c);
            tmp += get(k, c);
            set(k, c, get(k, c) - tau[k] * tmp);
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k) set(r, c, get(r, c) - tau[k] * ess[k][r] * tmp);
        }
    };
    step(StepC<0>{}); step(StepC<1>{}); step(StepC<2>{}); step(StepC<3>{}); step(StepC<4>{});
    if constexpr (M > 5) step(StepC<5>{});
    if constexpr (M > 6) step(StepC<6>{});
    static_assert(M >= 5 && M <= 7, "five to seven constraints");
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        double q[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) q[r] = (r == M + j) ? 1.0 : 0.0;
#pragma unroll
        for (int k = M - 1; k >= 0; --k) {
            double tmp = q[k];
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k) tmp += ess[k][r] * q[r];
            q[k] -= tau[k] * tmp;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k) q[r] -= tau[k] * ess[k][r] * tmp;
            const double qk = q[k];
            double qr = qk;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k && r == rt[k]) qr = q[r];
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r > k && r == rt[k]) q[r] = qk;
            q[k] = qr;
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) N[9 * j + r] = q[r];
    }
#undef AQ
}

// epipolar constraint columns kron(x1, x2): entry 3 j + i multiplies E(i, j)  (x2' E x1 = 0)
template <int M>
MDRP_HD void epipolar_columns(const double (*x1h)[3], const double (*x2h)[3], double *A /*9 x M col-major, strided*/, int as) {
#pragma unroll
    for (int p = 0; p < M; ++p)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) A[(p * 9 + 3 * j + i) * as] = x1h[p][j] * x2h[p][i];
}

// ---------------------------------------------------------------- real roots, ascending (Sturm isolation, bisection, Newton)
// chain rows are stored triangularly: row i has degree <= D - i
template <int D>
struct SturmChain {
    double c[D + 1][D + 1];
    int deg[D + 1];
    int n;
};
template <int D>
MDRP_HD int sturm_changes(const SturmChain<D> &s, double x) {
    int changes = 0, last = 0;
    for (int i = 0; i < s.n; ++i) {
        double v = 0.0;
        for (int k = s.deg[i]; k >= 0; --k) v = v * x + s.c[i][k];
        const int sg = (v > 0) - (v < 0);
        if (sg != 0) { if (last != 0 && sg != last) ++changes; last = sg; }
    }
    return changes;
}
template <int D>
MDRP_HD int real_roots(const double *coef /*ascending powers, degree D*/, double *roots) {
    int d = D;
    while (d > 0 && coef[d] == 0.0) --d;
    if (d <= 0) return 0;
    SturmChain<D> s;
    for (int k = 0; k <= d; ++k) s.c[0][k] = coef[k] / coef[d];
    s.deg[0] = d;
    for (int k = 1; k <= d; ++k) s.c[1][k - 1] = (double)k * s.c[0][k];
    s.deg[1] = d - 1;
    s.n = 2;
    while (s.deg[s.n - 1] > 0) {
        double r[D + 1];
        const int dp = s.deg[s.n - 2], dq = s.deg[s.n - 1];
        for (int k = 0; k <= dp; ++k) r[k] = s.c[s.n - 2][k];
        const double *q = s.c[s.n - 1];
        for (int k = dp; k >= dq; --k) {
            const double f = r[k] / q[dq];
            for (int j = 0; j <= dq; ++j) r[k - dq + j] -= f * q[j];
            r[k] = 0.0;
        }
        int dr = dq - 1;
        while (dr > 0 && fabs(r[dr]) < 1e-300) --dr;
        double mx = 0.0;
        for (int k = 0; k <= dr; ++k) mx = fmax(mx, fabs(r[k]));
        if (mx == 0.0) break;
        const double inv = -1.0 / mx;
        for (int k = 0; k <= dr; ++k) s.c[s.n][k] = r[k] * inv;
        s.deg[s.n] = dr;
        ++s.n;
    }
    double bound = 0.0;
    for (int k = 0; k < d; ++k) bound = fmax(bound, fabs(s.c[0][k]));
    bound += 1.0;
    // explicit stack of intervals (lo, hi] with their sign-change counts; the left half is pushed last: ascending order
    double lo_s[D + 2], hi_s[D + 2];
    int clo_s[D + 2], chi_s[D + 2], depth_s[D + 2];
    int sp = 0, nr = 0;
    lo_s[0] = -bound; hi_s[0] = bound; clo_s[0] = sturm_changes<D>(s, -bound); chi_s[0] = sturm_changes<D>(s, bound); depth_s[0] = 0;
    sp = 1;
    while (sp > 0 && nr < D) {
        --sp;
        const double lo = lo_s[sp], hi = hi_s[sp];
        const int clo = clo_s[sp], chi = chi_s[sp], depth = depth_s[sp];
        const int n = clo - chi;
        if (n <= 0) continue;
        if (n == 1 || depth > 200 || hi - lo < 1e-15 * fmax(1.0, fmax(fabs(lo), fabs(hi)))) {
            if (n == 1) {
                double a = lo, b = hi, fa = 0.0;
                for (int k = d; k >= 0; --k) fa = fa * a + s.c[0][k];
                for (int it = 0; it < 200; ++it) {
                    const double mid = 0.5 * (a + b);
                    if (mid == a || mid == b) break;
                    double fm = 0.0;
                    for (int k = d; k >= 0; --k) fm = fm * mid + s.c[0][k];
                    if (fm == 0.0) { a = b = mid; break; }
                    if ((fm > 0) == (fa > 0) && fa != 0.0) { a = mid; fa = fm; } else b = mid;
                    if (it >= 12 && b - a < 1e-3 * fmax(1e-300, fabs(a) + fabs(b))) break; // Newton takes over from a tight bracket
                }
                double x = 0.5 * (a + b);
                for (int it = 0; it < 8; ++it) {
                    double v = 0.0, dv = 0.0;
                    for (int k = d; k >= 0; --k) { dv = dv * x + v; v = v * x + s.c[0][k]; }
                    if (dv == 0.0) break;
                    double xn = x - v / dv;
                    if (!(xn >= a && xn <= b)) { // outside the bracket: bisect instead
                        double fm = 0.0;
                        const double mid = 0.5 * (a + b);
                        for (int k = d; k >= 0; --k) fm = fm * mid + s.c[0][k];
                        if ((fm > 0) == (fa > 0) && fa != 0.0) { a = mid; fa = fm; } else b = mid;
                        xn = 0.5 * (a + b);
                    }
                    const bool done = fabs(xn - x) <= 4e-16 * fabs(xn);
                    x = xn;
                    if (done) break;
                }
                roots[nr++] = x;
            } else {
                for (int i = 0; i < n && nr < D; ++i) roots[nr++] = 0.5 * (lo + hi); // multiple / unresolvable cluster
            }
            continue;
        }
        const double mid = 0.5 * (lo + hi);
        const int cm = sturm_changes<D>(s, mid);
        if (cm - chi > 0) { lo_s[sp] = mid; hi_s[sp] = hi; clo_s[sp] = cm; chi_s[sp] = chi; depth_s[sp] = depth + 1; ++sp; }
        if (clo - cm > 0) { lo_s[sp] = lo; hi_s[sp] = mid; clo_s[sp] = clo; chi_s[sp] = cm; depth_s[sp] = depth + 1; ++sp; }
    }
    return nr;
}

// The same algorithm with the chain's generic shape (row i has degree exactly D - i) compiled in: every index is a compile-time
// constant, so the 66 coefficients of a degree-10 chain live in registers instead of scratch memory (the dynamic version spends
// half of the 5-point solver's time in scratch round trips).  The interval stack is caller-provided strided storage (LDS on the
// device).  A chain that drops a degree (or a vanishing leading coefficient) goes to the generic routine above — same results.
// stack: D + 2 entries each (lo, hi, cc: entry i at [i stride]); isolating intervals: entry j at ilo / ihi [j istride].  The two may share their
// storage — the stack growing from entry 0, the isolated intervals down from entry D + 1 (ilo = lo + (D + 1) stride, istride = -stride): every
// interval on the stack holds at least one root that is not isolated yet, so the two together never need more than D entries (the device does this:
// the root kernel's LDS is what bounds its occupancy).
struct RootStack { double *lo, *hi; int *cc; double *ilo, *ihi; int stride, istride; };
template <int D>
MDRP_HD constexpr int chain_off(int i) { return i * (D + 1) - i * (i - 1) / 2; }
template <int D>
MDRP_HD int sturm_changes_static(const double *ch, double x) {
    int changes = 0, last = 0;
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        double v = ch[chain_off<D>(i) + D - i];
#pragma unroll
        for (int k = D - i - 1; k >= 0; --k) v = v * x + ch[chain_off<D>(i) + k];
        const int sg = (v > 0) - (v < 0);
        if (sg != 0) { if (last != 0 && sg != last) ++changes; last = sg; }
    }
    return changes;
}
template <int D>
MDRP_HD int real_roots_fast(const double *coef, double *roots, const RootStack &st) {
    if (coef[D] == 0.0) return real_roots<D>(coef, roots);
    double ch[(D + 1) * (D + 2) / 2];
    bool generic = true;
    {
        const double lead = coef[D];
#pragma unroll
        for (int k = 0; k <= D; ++k) ch[k] = coef[k] / lead;
#pragma unroll
        for (int k = 1; k <= D; ++k) ch[chain_off<D>(1) + k - 1] = (double)k * ch[k];
    }
#pragma unroll
    for (int i = 2; i <= D; ++i) {
        const int m = D - i + 1; // degree of the divisor, row i - 1
        double r[D + 1];
#pragma unroll
        for (int k = 0; k <= D; ++k) r[k] = (k <= m + 1) ? ch[chain_off<D>(i - 2) + k] : 0.0;
        const double *q = ch + chain_off<D>(i - 1);
        const double f1 = r[m + 1] / q[m];
#pragma unroll
        for (int j = 0; j <= D; ++j) if (j <= m) r[j + 1] -= f1 * q[j];
        const double f0 = r[m] / q[m];
#pragma unroll
        for (int j = 0; j <= D; ++j) if (j <= m) r[j] -= f0 * q[j];
        double mx = 0.0;
#pragma unroll
        for (int k = 0; k <= D; ++k) if (k < m) mx = fmax(mx, fabs(r[k]));
        if (!(fabs(r[m - 1]) >= 1e-300) || !(mx > 0.0) || !(mx < 1e300)) generic = false;
        const double inv = -1.0 / mx;
#pragma unroll
        for (int k = 0; k <= D; ++k) if (k < m) ch[chain_off<D>(i) + k] = r[k] * inv;
    }
    double bound = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) bound = fmax(bound, fabs(ch[k]));
    bound += 1.0;
    const int ss = st.stride, is = st.istride;
    // phase 1 — isolation only: every trip of the loop is one Sturm evaluation for every lane (the root polishing used to sit
    // inside this loop: a wavefront then paid for both branches on every trip).  Isolating intervals come out in ascending order.
    int sp = 1, ni = 0;
    st.lo[0] = -bound; st.hi[0] = bound;
    st.cc[0] = sturm_changes_static<D>(ch, -bound) | (sturm_changes_static<D>(ch, bound) << 8);
    if (!generic) sp = 0;
    while (sp > 0 && ni < D) {
        --sp;
        const double lo = st.lo[sp * ss], hi = st.hi[sp * ss];
        const int packed = st.cc[sp * ss];
        const int clo = packed & 0xFF, chi = (packed >> 8) & 0xFF, depth = packed >> 16;
        const int n = clo - chi;
        if (n <= 0) continue;
        // (sign counts that do not add up — an interval claiming more roots than the polynomial has left — could outgrow the D + 2 entries: the
        // generic path then; never seen, and before round 6 it would have written past the arrays)
        if (sp + ni + (n == 1 ? 1 : 2) > D + 2) { generic = false; break; }
        if (n == 1) { st.ilo[ni * is] = lo; st.ihi[ni * is] = hi; ++ni; continue; }
        if (depth > 200 || hi - lo < 1e-15 * fmax(1.0, fmax(fabs(lo), fabs(hi)))) { // multiple / unresolvable cluster
            for (int i = 0; i < n && ni < D && sp + ni < D + 2; ++i) { st.ilo[ni * is] = 0.5 * (lo + hi); st.ihi[ni * is] = 0.5 * (lo + hi); ++ni; }
            continue;
        }
        const double mid = 0.5 * (lo + hi);
        const int cm = sturm_changes_static<D>(ch, mid);
        if (cm - chi > 0) { st.lo[sp * ss] = mid; st.hi[sp * ss] = hi; st.cc[sp * ss] = cm | (chi << 8) | ((depth + 1) << 16); ++sp; }
        if (clo - cm > 0) { st.lo[sp * ss] = lo; st.hi[sp * ss] = mid; st.cc[sp * ss] = clo | (cm << 8) | ((depth + 1) << 16); ++sp; }
    }
    if (!generic) return real_roots<D>(coef, roots); // (a chain with a vanishing leading coefficient; the overflow above)
    // phase 2 — one root per trip: bisection on the sign of p down to a tight bracket, then safeguarded Newton
    auto peval = [&](double x) {
        double v = ch[D];
#pragma unroll
        for (int k = D - 1; k >= 0; --k) v = v * x + ch[k];
        return v;
    };
    for (int j = 0; j < ni; ++j) {
        double a = st.ilo[j * is], b = st.ihi[j * is];
        if (a == b) { roots[j] = a; continue; }
        double fa = peval(a);
        for (int it = 0; it < 200; ++it) {
            const double mid = 0.5 * (a + b);
            if (mid == a || mid == b) break;
            const double fm = peval(mid);
            if (fm == 0.0) { a = b = mid; break; }
            if ((fm > 0) == (fa > 0) && fa != 0.0) { a = mid; fa = fm; } else b = mid;
            if (it >= 12 && b - a < 1e-3 * fmax(1e-300, fabs(a) + fabs(b))) break;
        }
        double x = 0.5 * (a + b);
        for (int it = 0; it < 8; ++it) {
            double v = ch[D], dv = 0.0;
#pragma unroll
            for (int k = D - 1; k >= 0; --k) { dv = dv * x + v; v = v * x + ch[k]; }
            if (dv == 0.0) break;
            double xn = x - v / dv;
            if (!(xn >= a && xn <= b)) {
                const double mid = 0.5 * (a + b);
                const double fm = peval(mid);
                if ((fm > 0) == (fa > 0) && fa != 0.0) { a = mid; fa = fm; } else b = mid;
                xn = 0.5 * (a + b);
            }
            const bool done = fabs(xn - x) <= 4e-16 * fabs(xn);
            x = xn;
            if (done) break;
        }
        roots[j] = x;
    }
    return ni;
}

// ---------------------------------------------------------------- polynomials in (x, y, z)
// linear: [x, y, z, 1]; quadratic: [xx, yy, zz, xy, xz, yz, x, y, z, 1]; cubic: Nistér's order — the first ten monomials are
// eliminated, the last ten are [x, y, 1] (x) powers of z:
//   x3 y3 x2y xy2 x2z x2 xyz xy y2z y2 | xz2 xz x | yz2 yz y | z3 z2 z 1
MDRP_HD constexpr int quad_index(int a, int b, int c) { // exponents of x, y, z (sum <= 2)
    return a == 2 ? 0 : b == 2 ? 1 : c == 2 ? 2 : (a == 1 && b == 1) ? 3 : (a == 1 && c == 1) ? 4 : (b == 1 && c == 1) ? 5 : a == 1 ? 6 : b == 1 ? 7 : c == 1 ? 8 : 9;
}
MDRP_HD constexpr int cubic_index(int a, int b, int c) {
    return a == 3 ? 0 : b == 3 ? 1 : (a == 2 && b == 1) ? 2 : (a == 1 && b == 2) ? 3 : (a == 2 && c == 1) ? 4 : (a == 2) ? 5
         : (a == 1 && b == 1 && c == 1) ? 6 : (a == 1 && b == 1) ? 7 : (b == 2 && c == 1) ? 8 : (b == 2) ? 9
         : (a == 1 && c == 2) ? 10 : (a == 1 && c == 1) ? 11 : (a == 1) ? 12 : (b == 1 && c == 2) ? 13 : (b == 1 && c == 1) ? 14 : (b == 1) ? 15
         : c == 3 ? 16 : c == 2 ? 17 : c == 1 ? 18 : 19;
}
MDRP_HD constexpr int lin_exp(int i, int v) { return i == v ? 1 : 0; } // exponent of variable v in linear monomial i (i = 3: constant)
MDRP_HD constexpr int quad_exp(int i, int v) {
    constexpr int T[10][3] = {{2, 0, 0}, {0, 2, 0}, {0, 0, 2}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
    return T[i][v];
}
// q += s * a * b   (linear x linear)
MDRP_HD void lin_mul_add(const double *a, const double *b, double s, double *q) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            q[quad_index(lin_exp(i, 0) + lin_exp(j, 0), lin_exp(i, 1) + lin_exp(j, 1), lin_exp(i, 2) + lin_exp(j, 2))] += s * a[i] * b[j];
}
// c += s * q * l   (quadratic x linear)
MDRP_HD void quad_lin_mul_add(const double *q, const double *l, double s, double *c) {
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c[cubic_index(quad_exp(i, 0) + lin_exp(j, 0), quad_exp(i, 1) + lin_exp(j, 1), quad_exp(i, 2) + lin_exp(j, 2))] += s * q[i] * l[j];
}

// ---------------------------------------------------------------- 5-point: essential matrices
// x1h, x2h: the five sample bearings (unit vectors).  Es: up to 10 matrices, row-major, unit Frobenius norm.
// Caller-provided strided storage for the two dynamically indexed arrays of the solver: the 10 x 10 matrix that is LU-factorised
// with partial pivoting (element (r, k) at C[(10 r + k) cs]) and, once that is dead, the root finder's interval stack (the two
// may alias).  On the device both live in LDS, one column of 64 lanes per element (in scratch memory the solver spent 90 % of
// its time waiting for them); on the host they are plain local arrays.
struct Solve5Store { double *C; int cs; RootStack rs; };
// Round 6: the solver in two halves, so that the device can run them as two kernels (mdrp_classic.h kc_solve5_reduce / kc_solve5_roots) — the
// elimination needs the 10 x 10 LU in LDS and R in registers (512 registers, 51 KB per wavefront: three wavefronts per CU), the root finder and the
// pose extraction need neither.  What passes from one half to the other is Reduce5 (75 doubles): the null space as linear polynomials and the
// three rows of B(z).  Every value is computed by the same expressions as in rounds 1-5; relpose_5pt_emit below is the two halves back to back.
struct Reduce5 { double El[3][3][4], bx[3][4], by[3][4], b1[3][5]; };
constexpr int REDUCE5_DOUBLES = 36 + 12 + 12 + 15;

// E(i, j) = x N0 + y N1 + z N2 + N3 as linear polynomials [x, y, z, 1]: the only copy of the null space that is kept
MDRP_HD void relpose_5pt_nullspace(const double (*x1h)[3], const double (*x2h)[3], double *C /*columns 0..2 of the 9 x 5 constraint matrix, strided*/, int cs, double El[3][3][4]) {
    double N[36];
    epipolar_nullspace<5, 2>(x1h, x2h, C, cs, N); // 27 elements in C
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) El[i][j][k] = N[k * 9 + 3 * j + i];
}

// red.El in; red.bx / by / b1 out
MDRP_HD bool relpose_5pt_eliminate(const Solve5Store &store, Reduce5 &red) {
    double (&El)[3][3][4] = red.El;
    // The ten cubic constraints split into  L m_left + R m_right = 0  (ten eliminated monomials | [x y 1] (x) powers of z).  Only rows
    // 4..9 of X = L^-1 R are needed:  X_i = w_i R  with  L' w_i = e_i.  M = L' (M(r, k) = L(k, r): row = monomial, column = constraint) is
    // LU-factorised with partial pivoting: the ROW index of a swap is data dependent, everything else is static once the column loop is
    // unrolled.  Columns 2..9 live in the caller's strided storage (LDS on the device: element (r, k) at M[(8 r + k - 2) cs]), columns 0
    // and 1 in registers with their swaps as select chains — 80 elements = 40 KB per wavefront instead of 100 = 51 KB is a fourth
    // wavefront per CU for this kernel (round 6).  R stays in registers (statically indexed: it is never permuted).
    double *const M = store.C;
    const int cs = store.cs;
    double mc[2][10]; // columns 0 and 1 of M
#define M5(r, k) M[(8 * (r) + (k) - 2) * cs]
    auto mget = [&](int r, int k) __attribute__((always_inline)) -> double { return k < 2 ? mc[k][r] : M5(r, k); }; // r, k: constants after unrolling
    auto mset = [&](int r, int k, double v) __attribute__((always_inline)) { if (k < 2) mc[k][r] = v; else M5(r, k) = v; };
    double R[10][10];
    {
        double EEs[6][10], tr[10]; // E E' is symmetric: entry (i, j) at sym(i, j)
#define SYM3(i, j) ((i) <= (j) ? (i) * 3 - (i) * ((i) - 1) / 2 + (j) - (i) : (j) * 3 - (j) * ((j) - 1) / 2 + (i) - (j))
#pragma unroll
        for (int k = 0; k < 10; ++k) tr[k] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j) {
#pragma unroll
                for (int k = 0; k < 10; ++k) EEs[SYM3(i, j)][k] = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) lin_mul_add(El[i][k], El[j][k], 1.0, EEs[SYM3(i, j)]);
            }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 10; ++k) tr[k] += EEs[SYM3(i, i)][k];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double row[20];
#pragma unroll
                for (int k = 0; k < 20; ++k) row[k] = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) quad_lin_mul_add(EEs[SYM3(i, k)], El[k][j], 2.0, row);
                quad_lin_mul_add(tr, El[i][j], -1.0, row);
#pragma unroll
                for (int k = 0; k < 10; ++k) { mset(k, 3 * i + j, row[k]); R[3 * i + j][k] = row[10 + k]; }
            }
        double det[20], m[10];
#pragma unroll
        for (int k = 0; k < 20; ++k) det[k] = 0.0;
        for (int k = 0; k < 10; ++k) m[k] = 0.0;
        lin_mul_add(El[1][1], El[2][2], 1.0, m); lin_mul_add(El[1][2], El[2][1], -1.0, m); quad_lin_mul_add(m, El[0][0], 1.0, det);
        for (int k = 0; k < 10; ++k) m[k] = 0.0;
        lin_mul_add(El[1][0], El[2][2], 1.0, m); lin_mul_add(El[1][2], El[2][0], -1.0, m); quad_lin_mul_add(m, El[0][1], -1.0, det);
        for (int k = 0; k < 10; ++k) m[k] = 0.0;
        lin_mul_add(El[1][0], El[2][1], 1.0, m); lin_mul_add(El[1][1], El[2][0], -1.0, m); quad_lin_mul_add(m, El[0][2], 1.0, det);
#pragma unroll
        for (int k = 0; k < 10; ++k) { M5(k, 9) = det[k]; R[9][k] = det[10 + k]; }
#undef SYM3
    }
    // P M = Lo Up, partial pivoting; the multipliers stay in the eliminated positions; perm[i] = original row now at position i
    int perm[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) perm[i] = i;
#pragma unroll
    for (int col = 0; col < 10; ++col) {
        int piv = col;
        double pv = fabs(mget(col, col));
#pragma unroll
        for (int r = 0; r < 10; ++r) if (r > col) { const double v = fabs(mget(r, col)); if (v > pv) { pv = v; piv = r; } }
        if (!(pv > 0.0)) return false;
        if (piv != col) {
#pragma unroll
            for (int k = 2; k < 10; ++k) { const double t = M5(col, k); M5(col, k) = M5(piv, k); M5(piv, k) = t; }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double top = mc[c][col];
                double low = top;
#pragma unroll
                for (int r = 0; r < 10; ++r) if (r > col && r == piv) { low = mc[c][r]; mc[c][r] = top; }
                mc[c][col] = low;
            }
            int pp = 0;
#pragma unroll
            for (int i = 0; i < 10; ++i) if (i == piv) pp = perm[i];
            const int pc = perm[col];
#pragma unroll
            for (int i = 0; i < 10; ++i) perm[i] = (i == piv) ? pc : ((i == col) ? pp : perm[i]);
        }
        double prow[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) prow[k] = k >= col ? mget(col, k) : 0.0;
        const double inv = 1.0 / prow[col];
#pragma unroll
        for (int r = 0; r < 10; ++r)
            if (r > col) {
                const double f = mget(r, col) * inv;
                mset(r, col, f);
#pragma unroll
                for (int k = 0; k < 10; ++k) if (k > col) mset(r, k, mget(r, k) - f * prow[k]);
            }
    }
    // rows 4..9 of X, two at a time: forward / back substitution of e_t, then w R; each pair (<x2 z>, <x2>), (<xyz>, <xy>),
    // (<y2 z>, <y2>) gives one row  u - z v  of  B(z) [x y 1]' = 0
    double (&bx)[3][4] = red.bx, (&by)[3][4] = red.by, (&b1)[3][5] = red.b1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double uv[2][10];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = 4 + 2 * i + h;
            double y[10];
#pragma unroll
            for (int a = 0; a < 10; ++a) {
                double v = (perm[a] == t) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 10; ++k) if (k < a) v -= mget(a, k) * y[k];
                y[a] = v;
            }
#pragma unroll
            for (int a = 9; a >= 0; --a) {
                double v = y[a];
#pragma unroll
                for (int k = 0; k < 10; ++k) if (k > a) v -= mget(a, k) * y[k];
                y[a] = v / mget(a, a);
            }
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < 10; ++r) acc += y[r] * R[r][k];
                uv[h][k] = acc;
            }
        }
        const double *u = uv[0], *v = uv[1];
        bx[i][3] = -v[0]; bx[i][2] = u[0] - v[1]; bx[i][1] = u[1] - v[2]; bx[i][0] = u[2];
        by[i][3] = -v[3]; by[i][2] = u[3] - v[4]; by[i][1] = u[4] - v[5]; by[i][0] = u[5];
        b1[i][4] = -v[6]; b1[i][3] = u[6] - v[7]; b1[i][2] = u[7] - v[8]; b1[i][1] = u[8] - v[9]; b1[i][0] = u[9];
    }
#undef M5
    return true;
}

MDRP_HD bool relpose_5pt_reduce(const double (*x1h)[3], const double (*x2h)[3], const Solve5Store &store, Reduce5 &red) {
    relpose_5pt_nullspace(x1h, x2h, store.C, store.cs, red.El); // the 9 x 5 constraint matrix borrows the LU storage (dead before it is built)
#if defined(__HIP_DEVICE_COMPILE__)
    // The device runs the two parts as separate kernels (mdrp_classic.h): there the elimination reads the null space from memory.  Inlined behind the
    // null space, -ffp-contract=fast forms other fused multiply-adds in it (tools/ubench/solve5_split_bits.hip: the three rows of B(z) differ in their
    // last bits on 70 % of random samples); hiding where the 36 values come from makes this path compute what the kernels compute, bit for bit.
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(red.El[i][j][k]));
#endif
    return relpose_5pt_eliminate(store, red);
}

// the essential matrix of one solution (x, y, z) of the null-space coordinates, unit Frobenius norm
MDRP_HD void essential_from_xyz(const double El[3][3][4], double x, double y, double z, double e[9]) {
    double nrm = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { e[3 * i + j] = x * El[i][j][0] + y * El[i][j][1] + z * El[i][j][2] + El[i][j][3]; nrm += e[3 * i + j] * e[3 * i + j]; }
    nrm = 1.0 / sqrt(nrm);
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] *= nrm;
}

// det B(z) = 0 -> real roots z (Sturm, ascending), then (x, y) from two rows of B(z): emit_xyz(x, y, z) per solution, in the order of the roots
template <class EmitXyz>
MDRP_HD int relpose_5pt_roots_xyz(const double bx[3][4], const double by[3][4], const double b1[3][5], const RootStack &rs, EmitXyz &&emit_xyz) {
    double c[11];
    for (int k = 0; k < 11; ++k) c[k] = 0.0;
    constexpr int PERM[6][4] = {{0, 1, 2, 1}, {0, 2, 1, -1}, {1, 0, 2, -1}, {1, 2, 0, 1}, {2, 0, 1, 1}, {2, 1, 0, -1}};
    for (int p = 0; p < 6; ++p) { // sign * bx[r0] * by[r1] * b1[r2] over the permutations (r0, r1, r2)
        const int r0 = PERM[p][0], r1 = PERM[p][1], r2 = PERM[p][2];
        const double sgn = (double)PERM[p][3];
        double ab[7];
        for (int k = 0; k < 7; ++k) ab[k] = 0.0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) ab[i + j] += bx[r0][i] * by[r1][j];
        for (int i = 0; i < 7; ++i) for (int j = 0; j < 5; ++j) c[i + j] += sgn * ab[i] * b1[r2][j];
    }
    double roots[10];
    const int nr = real_roots_fast<10>(c, roots, rs);
    int n_out = 0;
    for (int s = 0; s < nr; ++s) {
        const double z = roots[s];
        double B[3][3];
        for (int i = 0; i < 3; ++i) {
            B[i][0] = ((bx[i][3] * z + bx[i][2]) * z + bx[i][1]) * z + bx[i][0];
            B[i][1] = ((by[i][3] * z + by[i][2]) * z + by[i][1]) * z + by[i][0];
            B[i][2] = (((b1[i][4] * z + b1[i][3]) * z + b1[i][2]) * z + b1[i][1]) * z + b1[i][0];
        }
        int r0 = 0, r1 = 1;
        double best = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int b = a + 1; b < 3; ++b) {
                const double dd = fabs(B[a][0] * B[b][1] - B[a][1] * B[b][0]);
                if (dd > best) { best = dd; r0 = a; r1 = b; }
            }
        if (!(best > 0.0)) continue;
        const double det = B[r0][0] * B[r1][1] - B[r0][1] * B[r1][0];
        const double x = (-B[r0][2] * B[r1][1] + B[r0][1] * B[r1][2]) / det;
        const double y = (-B[r0][0] * B[r1][2] + B[r0][2] * B[r1][0]) / det;
        emit_xyz(x, y, z);
        ++n_out;
    }
    return n_out;
}

// emit(const double E[9]) is called for every essential matrix, in the order of the roots: the device solver hands each one straight
// on to motion_from_essential and to the model slots (ten matrices and ten poses held in arrays cost the kernel 420 registers)
template <class Emit>
MDRP_HD int relpose_5pt_emit(const double (*x1h)[3], const double (*x2h)[3], const Solve5Store &store, Emit &&emit) {
    Reduce5 red;
    if (!relpose_5pt_reduce(x1h, x2h, store, red)) return 0;
    return relpose_5pt_roots_xyz(red.bx, red.by, red.b1, store.rs, [&](double x, double y, double z) {
        double e[9];
        essential_from_xyz(red.El, x, y, z, e);
        emit(e);
    });
}
MDRP_HD int relpose_5pt_E(const double (*x1h)[3], const double (*x2h)[3], double (*Es)[9], const Solve5Store &store) {
    int n = 0;
    return relpose_5pt_emit(x1h, x2h, store, [&](const double *e) { for (int k = 0; k < 9; ++k) Es[n][k] = e[k]; ++n; });
}

// ---------------------------------------------------------------- E -> poses (motion_from_essential @0x1dd540)
// check_cheirality @0x1dce00 on unit bearings, min_depth 0
MDRP_HD bool cheirality_bearing(const double R[9], const double t[3], const double x1[3], const double x2[3]) {
    const double u0 = R[0] * x1[0] + R[1] * x1[1] + R[2] * x1[2], u1 = R[3] * x1[0] + R[4] * x1[1] + R[5] * x1[2],
                 u2 = R[6] * x1[0] + R[7] * x1[1] + R[8] * x1[2];
    const double a = -(u0 * x2[0] + u1 * x2[1] + u2 * x2[2]);
    const double b1 = -(u0 * t[0] + u1 * t[1] + u2 * t[2]);
    const double b2 = x2[0] * t[0] + x2[1] * t[1] + x2[2] * t[2];
    return (b1 - a * b2) > 0.0 && (-a * b1 + b2) > 0.0;
}
template <class Emit>
MDRP_HD int motion_from_essential_emit(const double E[9], const double (*x1h)[3], const double (*x2h)[3], int npts, Emit &&emit) {
    const double c0[3] = {E[0], E[3], E[6]}, c1[3] = {E[1], E[4], E[7]}, c2[3] = {E[2], E[5], E[8]};
    double u12[3], u13[3], u23[3];
    cross3(c0, c1, u12); cross3(c0, c2, u13); cross3(c1, c2, u23);
    const double n12 = dot3(u12, u12), n13 = dot3(u13, u13), n23 = dot3(u23, u23);
    double UW[3][3], Vt[3][3]; // UW[col][.]
    const double *a, *u;
    double nu;
    if (n12 > n13) { if (n12 > n23) { a = c0; u = u12; nu = n12; } else { a = c1; u = u23; nu = n23; } }
    else { if (n13 > n23) { a = c0; u = u13; nu = n13; } else { a = c1; u = u23; nu = n23; } }
    const double na = 1.0 / sqrt(dot3(a, a)), su = 1.0 / sqrt(nu);
    for (int k = 0; k < 3; ++k) { UW[1][k] = a[k] * na; UW[2][k] = u[k] * su; }
    double t0[3];
    cross3(UW[2], UW[1], t0);
    for (int k = 0; k < 3; ++k) UW[0][k] = -t0[k];
    for (int j = 0; j < 3; ++j) {
        Vt[0][j] = UW[1][0] * E[j] + UW[1][1] * E[3 + j] + UW[1][2] * E[6 + j];
        Vt[1][j] = -(UW[0][0] * E[j] + UW[0][1] * E[3 + j] + UW[0][2] * E[6 + j]);
    }
    double n0 = 1.0 / sqrt(dot3(Vt[0], Vt[0]));
    for (int j = 0; j < 3; ++j) Vt[0][j] *= n0;
    const double d = dot3(Vt[0], Vt[1]);
    for (int j = 0; j < 3; ++j) Vt[1][j] -= d * Vt[0][j];
    n0 = 1.0 / sqrt(dot3(Vt[1], Vt[1]));
    for (int j = 0; j < 3; ++j) Vt[1][j] *= n0;
    cross3(Vt[0], Vt[1], Vt[2]);
    int n_out = 0;
    double t[3] = {UW[2][0], UW[2][1], UW[2][2]};
    for (int pass = 0; pass < 2; ++pass) { // candidates in the order (R1, t), (R1, -t), (R2, -t), (R2, t)
        double R[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[3 * i + j] = UW[0][i] * Vt[0][j] + UW[1][i] * Vt[1][j] + UW[2][i] * Vt[2][j];
        for (int sgn = 0; sgn < 2; ++sgn) {
            bool ok = true;
            for (int k = 0; k < npts; ++k) ok = ok && cheirality_bearing(R, t, x1h[k], x2h[k]);
            if (ok) {
                Model m;
                model_identity(m);
                R_to_quat(R, m.q);
                m.t[0] = t[0]; m.t[1] = t[1]; m.t[2] = t[2];
                emit(m);
                ++n_out;
            }
            for (int k = 0; k < 3; ++k) t[k] = -t[k];
        }
        // after two flips t is back at +t; the reference continues from -t: (R2, -t) then (R2, +t)
        for (int k = 0; k < 3; ++k) { t[k] = -t[k]; UW[0][k] = -UW[0][k]; UW[1][k] = -UW[1][k]; }
    }
    return n_out;
}

MDRP_HD int motion_from_essential(const double E[9], const double (*x1h)[3], const double (*x2h)[3], int npts, Model *out) {
    int n = 0;
    return motion_from_essential_emit(E, x1h, x2h, npts, [&](const Model &m) { out[n++] = m; });
}

constexpr int MAX_MODELS_5PT = 10; // one pose per essential matrix passes the cheirality of all five points (generic data)
// emit(const Model &, int k) for the k-th pose of the sample (k < MAX_MODELS_5PT), in the reference's order
template <class Emit>
MDRP_HD int solver_relpose_5pt_emit(const double (*x1h)[3], const double (*x2h)[3], const Solve5Store &store, Emit &&emit) {
    int n = 0;
    relpose_5pt_emit(x1h, x2h, store, [&](const double *e) {
        if (n < MAX_MODELS_5PT) motion_from_essential_emit(e, x1h, x2h, 5, [&](const Model &m) { if (n < MAX_MODELS_5PT) { emit(m, n); ++n; } });
    });
    return n;
}
MDRP_HD int solver_relpose_5pt(const double (*x1h)[3], const double (*x2h)[3], Model *out /*[MAX_MODELS_5PT]*/, const Solve5Store &store) {
    return solver_relpose_5pt_emit(x1h, x2h, store, [&](const Model &m, int k) { out[k] = m; });
}

// ---------------------------------------------------------------- 6-point, one shared unknown focal length
// relpose_6pt_shared_focal of the reference (SharedFocalRelativePoseEstimator::generate_models; estimate_shared_focal_relative_pose,
// /root/reference/eval_shared_f.py:161).  The binary uses a generated elimination template and a 15 x 15 action matrix; the solution
// SET belongs to the polynomial system, and this restates the published formulation (Stewenius et al. 2005; as a polynomial
// eigenvalue problem Kukelova, Bujnak, Pajdla 2008):  F = x N0 + y N1 + N2 on the null space of the six epipolar constraints,
// Q = diag(1, 1, w), w = 1 / f^2:  2 F Q F' Q F - tr(F Q F' Q) F = 0 and det F = 0 are ten cubics in (x, y), quadratic in w:
// (M0 + w M1 + w^2 M2) v = 0, v = (x3, x2y, xy2, y3, x2, xy, y2, x, y, 1).  In u = 1 / w = f^2 the leading matrix M0 is regular:
// the eigenvalues of the 20 x 20 companion matrix [[0, I], [-M0^-1 M2, -M0^-1 M1]] (Hessenberg reduction + shifted QR, below), five of
// them the spurious u = 0.  Real u > 0 -> f, null vector of M(w) -> (x, y) -> F -> E = diag(1,1,1/f) F diag(1,1,1/f) -> poses with the
// cheirality of all six points.  Solutions come out by ascending focal length (the binary's order is the order of Eigen's
// eigenvalues; pinned: the sets, and the estimator's trajectory — 48 / 48 identical results without that order).
// One lane per sample; the matrices (about 1100 doubles) are dynamically indexed and live in the lane's scratch memory.
// The 6-point solver is ~2000 lines of loop nests over small matrices when fully unrolled: 512 registers and 763 spilled VGPRs at one
// wavefront per SIMD.  Kept rolled it is a compact scalar program per lane whose arrays live in scratch / LDS, at full occupancy.
#define MDRP_ROLLED _Pragma("unroll 1")
MDRP_HD double sgn_of(double a, double b) { return b >= 0.0 ? fabs(a) : -fabs(a); }
// Square matrices of the 6-point solver behind an accessor (plain row-major storage here).  Measured and dropped: one matrix per lane
// in LDS with the lanes interleaved (element e of slot g at base[e * slots + g]).  The 20 x 20 companion matrix is 3.2 KB, a CU's LDS
// holds 48 of them, and 48 lanes per CU with every access ~100 cycles of exposed LDS latency are slower (220-310 pairs/s) than 512
// lanes per CU waiting for scratch memory (480 pairs/s): the sequential chain per sample is the problem, not where the matrix lives.
struct PlainMat {
    double *a; int n;
    MDRP_HD double &operator()(int i, int j) const { return a[i * n + j]; }
};
// real and imaginary parts of the eigenvalues of a (n x n, destroyed); false if the QR iteration did not converge
template <class Mat>
MDRP_HD bool hessenberg_qr_eigenvalues(Mat a, int n, double *wr, double *wi) {
#define HA(i, j) a((i), (j))
    MDRP_ROLLED for (int m = 1; m < n - 1; ++m) {
        double x = 0.0;
        int piv = m;
        MDRP_ROLLED for (int j = m; j < n; ++j)
            if (fabs(HA(j, m - 1)) > fabs(x)) { x = HA(j, m - 1); piv = j; }
        if (piv != m) {
            MDRP_ROLLED for (int j = m - 1; j < n; ++j) { const double t = HA(piv, j); HA(piv, j) = HA(m, j); HA(m, j) = t; }
            MDRP_ROLLED for (int j = 0; j < n; ++j) { const double t = HA(j, piv); HA(j, piv) = HA(j, m); HA(j, m) = t; }
        }
        if (x != 0.0)
            MDRP_ROLLED for (int i = m + 1; i < n; ++i) {
                double y = HA(i, m - 1);
                if (y != 0.0) {
                    y /= x;
                    HA(i, m - 1) = y;
                    MDRP_ROLLED for (int j = m; j < n; ++j) HA(i, j) -= y * HA(m, j);
                    MDRP_ROLLED for (int j = 0; j < n; ++j) HA(j, m) += y * HA(j, i);
                }
            }
    }
    MDRP_ROLLED for (int i = 2; i < n; ++i)
        MDRP_ROLLED for (int j = 0; j < i - 1; ++j) HA(i, j) = 0.0;
    int nn = n - 1, its = 0;
    double t = 0.0, anorm = 0.0, p = 0, q = 0, r = 0, s, w, x, y, z;
    MDRP_ROLLED for (int i = 0; i < n; ++i)
        MDRP_ROLLED for (int j = (i > 0 ? i - 1 : 0); j < n; ++j) anorm += fabs(HA(i, j));
    while (nn >= 0) {
        int l;
        MDRP_ROLLED for (l = nn; l >= 1; --l) {
            s = fabs(HA(l - 1, l - 1)) + fabs(HA(l, l));
            if (s == 0.0) s = anorm;
            if (fabs(HA(l, l - 1)) + s == s) { HA(l, l - 1) = 0.0; break; }
        }
        x = HA(nn, nn);
        if (l == nn) { wr[nn] = x + t; wi[nn] = 0.0; --nn; its = 0; continue; }
        y = HA(nn - 1, nn - 1);
        w = HA(nn, nn - 1) * HA(nn - 1, nn);
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
        if (its == 60) return false;
        if (its == 10 || its == 20 || its == 30 || its == 40) { // exceptional shift
            t += x;
            MDRP_ROLLED for (int i = 0; i <= nn; ++i) HA(i, i) -= x;
            s = fabs(HA(nn, nn - 1)) + fabs(HA(nn - 1, nn - 2));
            y = x = 0.75 * s;
            w = -0.4375 * s * s;
        }
        ++its;
        int m;
        MDRP_ROLLED for (m = nn - 2; m >= l; --m) {
            z = HA(m, m);
            r = x - z; s = y - z;
            p = (r * s - w) / HA(m + 1, m) + HA(m, m + 1);
            q = HA(m + 1, m + 1) - z - r - s;
            r = HA(m + 2, m + 1);
            s = fabs(p) + fabs(q) + fabs(r);
            p /= s; q /= s; r /= s;
            if (m == l) break;
            const double u = fabs(HA(m, m - 1)) * (fabs(q) + fabs(r));
            const double v = fabs(p) * (fabs(HA(m - 1, m - 1)) + fabs(z) + fabs(HA(m + 1, m + 1)));
            if (u + v == v) break;
        }
        MDRP_ROLLED for (int i = m + 2; i <= nn; ++i) { HA(i, i - 2) = 0.0; if (i != m + 2) HA(i, i - 3) = 0.0; }
        MDRP_ROLLED for (int k = m; k <= nn - 1; ++k) {
            if (k != m) {
                p = HA(k, k - 1); q = HA(k + 1, k - 1); r = 0.0;
                if (k != nn - 1) r = HA(k + 2, k - 1);
                if ((x = fabs(p) + fabs(q) + fabs(r)) != 0.0) { p /= x; q /= x; r /= x; }
            }
            if ((s = sgn_of(sqrt(p * p + q * q + r * r), p)) != 0.0) {
                if (k == m) { if (l != m) HA(k, k - 1) = -HA(k, k - 1); }
                else HA(k, k - 1) = -s * x;
                p += s; x = p / s; y = q / s; z = r / s; q /= p; r /= p;
                MDRP_ROLLED for (int j = k; j <= nn; ++j) {
                    p = HA(k, j) + q * HA(k + 1, j);
                    if (k != nn - 1) { p += r * HA(k + 2, j); HA(k + 2, j) -= p * z; }
                    HA(k + 1, j) -= p * y; HA(k, j) -= p * x;
                }
                const int mmin = nn < k + 3 ? nn : k + 3;
                MDRP_ROLLED for (int i = l; i <= mmin; ++i) {
                    p = x * HA(i, k) + y * HA(i, k + 1);
                    if (k != nn - 1) { p += z * HA(i, k + 2); HA(i, k + 2) -= p * r; }
                    HA(i, k + 1) -= p * q; HA(i, k) -= p;
                }
            }
        }
    }
    return true;
#undef HA
}

// polynomials in (x, y) of degree <= 3 as 10 coefficients in the order x3, x2y, xy2, y3, x2, xy, y2, x, y, 1
MDRP_HD constexpr int six_exp_x(int i) { return i == 0 ? 3 : (i == 1 || i == 4) ? 2 : (i == 2 || i == 5 || i == 7) ? 1 : 0; }
MDRP_HD constexpr int six_exp_y(int i) { return i == 3 ? 3 : (i == 2 || i == 6) ? 2 : (i == 1 || i == 5 || i == 8) ? 1 : 0; }
MDRP_HD constexpr int six_index(int a, int b) { return a + b == 3 ? 3 - a : (a + b == 2 ? 6 - a : (a + b == 1 ? 8 - a : 9)); }
MDRP_HD void six_mul_add(const double *a, const double *b, double s, double *out) { // out += s a b (degrees above 3 do not occur)
    MDRP_ROLLED for (int i = 0; i < 10; ++i) {
        if (a[i] == 0.0) continue;
        MDRP_ROLLED for (int j = 0; j < 10; ++j) {
            if (b[j] == 0.0) continue;
            const int ex = six_exp_x(i) + six_exp_x(j), ey = six_exp_y(i) + six_exp_y(j);
            if (ex + ey <= 3) out[six_index(ex, ey)] += s * a[i] * b[j];
        }
    }
}
// null vector of a 10 x 10 matrix (row-major, destroyed) by complete pivoting; returns |last pivot| / |first pivot|
template <class Mat>
MDRP_HD double six_null_vector(Mat Mm, double *v) {
    int cp[10];
    MDRP_ROLLED for (int i = 0; i < 10; ++i) cp[i] = i;
    double first = 0.0, last = 0.0;
    MDRP_ROLLED for (int k = 0; k < 10; ++k) {
        int pr = k, pc = k;
        double best = -1.0;
        MDRP_ROLLED for (int i = k; i < 10; ++i)
            MDRP_ROLLED for (int j = k; j < 10; ++j)
                if (fabs(Mm(i, j)) > best) { best = fabs(Mm(i, j)); pr = i; pc = j; }
        if (k == 0) first = best;
        if (k == 9) { last = best; break; }
        if (pr != k) MDRP_ROLLED for (int j = 0; j < 10; ++j) { const double t = Mm(pr, j); Mm(pr, j) = Mm(k, j); Mm(k, j) = t; }
        if (pc != k) {
            MDRP_ROLLED for (int i = 0; i < 10; ++i) { const double t = Mm(i, pc); Mm(i, pc) = Mm(i, k); Mm(i, k) = t; }
            const int t = cp[pc]; cp[pc] = cp[k]; cp[k] = t;
        }
        const double piv = Mm(k, k);
        if (piv == 0.0) break;
        MDRP_ROLLED for (int i = k + 1; i < 10; ++i) {
            const double f = Mm(i, k) / piv;
            if (f != 0.0) MDRP_ROLLED for (int j = k; j < 10; ++j) Mm(i, j) -= f * Mm(k, j);
        }
    }
    double y[10];
    y[9] = 1.0;
    MDRP_ROLLED for (int i = 8; i >= 0; --i) {
        double s = 0.0;
        MDRP_ROLLED for (int j = i + 1; j < 10; ++j) s += Mm(i, j) * y[j];
        y[i] = Mm(i, i) != 0.0 ? -s / Mm(i, i) : 0.0;
    }
    MDRP_ROLLED for (int i = 0; i < 10; ++i) v[cp[i]] = y[i];
    return first > 0.0 ? last / first : 1.0;
}

// Gauss-Newton on the ten equations in (x, y, w): three steps bring the residual of an eigenpair to rounding level
MDRP_HD double six_pow(double x, int a) { return a == 0 ? 1.0 : (a == 1 ? x : (a == 2 ? x * x : x * x * x)); }
MDRP_HD void six_polish(const double *M0, const double *M1, const double *M2, double &px, double &py, double &pw) {
    MDRP_ROLLED for (int it = 0; it < 3; ++it) {
        const double x = px, y = py, w = pw;
        double mono[10], dmx[10], dmy[10];
        MDRP_ROLLED for (int e = 0; e < 10; ++e) {
            const int a = six_exp_x(e), b = six_exp_y(e);
            mono[e] = six_pow(x, a) * six_pow(y, b);
            dmx[e] = a > 0 ? a * six_pow(x, a - 1) * six_pow(y, b) : 0.0;
            dmy[e] = b > 0 ? b * six_pow(x, a) * six_pow(y, b - 1) : 0.0;
        }
        double JtJ[9], Jtr[3] = {0, 0, 0};
        MDRP_ROLLED for (int a = 0; a < 9; ++a) JtJ[a] = 0.0;
        MDRP_ROLLED for (int r = 0; r < 10; ++r) {
            double g = 0, gx = 0, gy = 0, gw = 0;
            MDRP_ROLLED for (int e = 0; e < 10; ++e) {
                const double c = M0[r * 10 + e] + w * (M1[r * 10 + e] + w * M2[r * 10 + e]);
                g += c * mono[e]; gx += c * dmx[e]; gy += c * dmy[e];
                gw += (M1[r * 10 + e] + 2.0 * w * M2[r * 10 + e]) * mono[e];
            }
            const double J[3] = {gx, gy, gw};
            MDRP_ROLLED for (int a = 0; a < 3; ++a) { Jtr[a] += J[a] * g; MDRP_ROLLED for (int b = 0; b < 3; ++b) JtJ[3 * a + b] += J[a] * J[b]; }
        }
        double d[3];
        if (!solve3x3(JtJ, Jtr, d)) return;
        px = x - d[0]; py = y - d[1]; pw = w - d[2];
    }
}

constexpr int MAX_MODELS_6PT = 15;
constexpr double SIX_MIN_U = 1e-5; // the defective cluster of the five spurious u = 0 spreads to ~1e-7 on scale-normalised points; f < 0.003 is no camera
// emit(const Model &, int k) for the k-th model (pose, f1 = f2 = f) of the sample, k < MAX_MODELS_6PT
// storage of the companion matrix (and, once that is dead, of M(w) for the null vectors)
struct PlainStore6 {
    double c[400];
    MDRP_HD PlainMat mat20() { return PlainMat{c, 20}; }
    MDRP_HD PlainMat mat10() { return PlainMat{c, 10}; }
};
template <class Store, class Emit>
MDRP_HD int solver_relpose_6pt_emit(const double (*x1h)[3], const double (*x2h)[3], Store &&store, Emit &&emit) {
    double Nq[27]; // three null vectors, each the column-major vec of a 3 x 3 matrix: F(i, j) = N[3 j + i]
    {
        double A[54];
        epipolar_columns<6>(x1h, x2h, A, 1);
        fullpiv_nullspace<6>(A, 1, Nq);
    }
    double F[3][3][10];
    MDRP_ROLLED for (int i = 0; i < 3; ++i)
        MDRP_ROLLED for (int j = 0; j < 3; ++j) {
            MDRP_ROLLED for (int e = 0; e < 10; ++e) F[i][j][e] = 0.0;
            F[i][j][7] = Nq[3 * j + i]; F[i][j][8] = Nq[9 + 3 * j + i]; F[i][j][9] = Nq[18 + 3 * j + i];
        }
    double S0[3][3][10], S1[3][3][10], tr0[10], tr1[10], tr2[10];
    MDRP_ROLLED for (int i = 0; i < 3; ++i)
        MDRP_ROLLED for (int k = 0; k < 3; ++k) {
            MDRP_ROLLED for (int e = 0; e < 10; ++e) { S0[i][k][e] = 0.0; S1[i][k][e] = 0.0; }
            six_mul_add(F[i][0], F[k][0], 1.0, S0[i][k]); six_mul_add(F[i][1], F[k][1], 1.0, S0[i][k]);
            six_mul_add(F[i][2], F[k][2], 1.0, S1[i][k]);
        }
    MDRP_ROLLED for (int e = 0; e < 10; ++e) {
        tr0[e] = S0[0][0][e] + S0[1][1][e];
        tr1[e] = S1[0][0][e] + S1[1][1][e] + S0[2][2][e];
        tr2[e] = S1[2][2][e];
    }
    double M0[100], M1[100], M2[100];
    MDRP_ROLLED for (int e = 0; e < 100; ++e) { M0[e] = 0.0; M1[e] = 0.0; M2[e] = 0.0; }
    MDRP_ROLLED for (int i = 0; i < 3; ++i)
        MDRP_ROLLED for (int l = 0; l < 3; ++l) {
            double *t0 = M0 + (3 * i + l) * 10, *t1 = M1 + (3 * i + l) * 10, *t2 = M2 + (3 * i + l) * 10;
            six_mul_add(S0[i][0], F[0][l], 2.0, t0); six_mul_add(S0[i][1], F[1][l], 2.0, t0); six_mul_add(tr0, F[i][l], -1.0, t0);
            six_mul_add(S1[i][0], F[0][l], 2.0, t1); six_mul_add(S1[i][1], F[1][l], 2.0, t1); six_mul_add(S0[i][2], F[2][l], 2.0, t1);
            six_mul_add(tr1, F[i][l], -1.0, t1);
            six_mul_add(S1[i][2], F[2][l], 2.0, t2); six_mul_add(tr2, F[i][l], -1.0, t2);
        }
    { // det F
        double m[10], *d = M0 + 90;
        MDRP_ROLLED for (int e = 0; e < 10; ++e) m[e] = 0.0;
        six_mul_add(F[1][1], F[2][2], 1.0, m); six_mul_add(F[1][2], F[2][1], -1.0, m); six_mul_add(m, F[0][0], 1.0, d);
        MDRP_ROLLED for (int e = 0; e < 10; ++e) m[e] = 0.0;
        six_mul_add(F[1][0], F[2][2], 1.0, m); six_mul_add(F[1][2], F[2][0], -1.0, m); six_mul_add(m, F[0][1], -1.0, d);
        MDRP_ROLLED for (int e = 0; e < 10; ++e) m[e] = 0.0;
        six_mul_add(F[1][0], F[2][1], 1.0, m); six_mul_add(F[1][1], F[2][0], -1.0, m); six_mul_add(m, F[0][2], 1.0, d);
    }
    // X = M0^-1 [M2 | M1]: LU with partial pivoting on a copy of M0, 20 right-hand sides; then the companion matrix
    auto C = store.mat20();
    double wr[20], wi[20];
    {
        double L[100], B[200];
        MDRP_ROLLED for (int e = 0; e < 100; ++e) L[e] = M0[e];
        MDRP_ROLLED for (int i = 0; i < 10; ++i)
            MDRP_ROLLED for (int j = 0; j < 10; ++j) { B[i * 20 + j] = M2[i * 10 + j]; B[i * 20 + 10 + j] = M1[i * 10 + j]; }
        MDRP_ROLLED for (int k = 0; k < 10; ++k) {
            int piv = k;
            MDRP_ROLLED for (int i = k + 1; i < 10; ++i) if (fabs(L[i * 10 + k]) > fabs(L[piv * 10 + k])) piv = i;
            if (L[piv * 10 + k] == 0.0) return 0;
            if (piv != k) {
                MDRP_ROLLED for (int j = 0; j < 10; ++j) { const double t = L[piv * 10 + j]; L[piv * 10 + j] = L[k * 10 + j]; L[k * 10 + j] = t; }
                MDRP_ROLLED for (int j = 0; j < 20; ++j) { const double t = B[piv * 20 + j]; B[piv * 20 + j] = B[k * 20 + j]; B[k * 20 + j] = t; }
            }
            MDRP_ROLLED for (int i = k + 1; i < 10; ++i) {
                const double f = L[i * 10 + k] / L[k * 10 + k];
                if (f == 0.0) continue;
                MDRP_ROLLED for (int j = k; j < 10; ++j) L[i * 10 + j] -= f * L[k * 10 + j];
                MDRP_ROLLED for (int j = 0; j < 20; ++j) B[i * 20 + j] -= f * B[k * 20 + j];
            }
        }
        MDRP_ROLLED for (int i = 9; i >= 0; --i)
            MDRP_ROLLED for (int j = 0; j < 20; ++j) {
                double s = B[i * 20 + j];
                MDRP_ROLLED for (int k = i + 1; k < 10; ++k) s -= L[i * 10 + k] * B[k * 20 + j];
                B[i * 20 + j] = s / L[i * 10 + i];
            }
        MDRP_ROLLED for (int i = 0; i < 10; ++i) {
            MDRP_ROLLED for (int j = 0; j < 20; ++j) { C(i, j) = j == 10 + i ? 1.0 : 0.0; C(10 + i, j) = -B[i * 20 + j]; }
        }
    }
    if (!hessenberg_qr_eigenvalues(C, 20, wr, wi)) return 0;
    double us[20];
    int nu = 0;
    MDRP_ROLLED for (int i = 0; i < 20; ++i)
        if (fabs(wi[i]) <= 1e-9 * (fabs(wr[i]) + 1e-300) && wr[i] > SIX_MIN_U) us[nu++] = wr[i];
    MDRP_ROLLED for (int i = 1; i < nu; ++i) { const double t = us[i]; int j = i - 1; while (j >= 0 && us[j] > t) { us[j + 1] = us[j]; --j; } us[j + 1] = t; }
    int n_out = 0;
    MDRP_ROLLED for (int s = 0; s < nu && n_out < MAX_MODELS_6PT; ++s) {
        const double w = 1.0 / us[s];
        auto Mw = store.mat10(); // the companion matrix is dead: reuse its storage
        double v[10];
        MDRP_ROLLED for (int i = 0; i < 10; ++i)
            MDRP_ROLLED for (int j = 0; j < 10; ++j) Mw(i, j) = M0[i * 10 + j] + w * (M1[i * 10 + j] + w * M2[i * 10 + j]);
        const double res = six_null_vector(Mw, v);
        if (!(res < 1e-6) || !(fabs(v[9]) > 0.0)) continue;
        double x = v[7] / v[9], y = v[8] / v[9], wv = w;
        six_polish(M0, M1, M2, x, y, wv);
        if (!(wv > 0.0)) continue;
        const double f = sqrt(1.0 / wv), invf = 1.0 / f;
        double E[9], nrm = 0.0;
        MDRP_ROLLED for (int i = 0; i < 3; ++i)
            MDRP_ROLLED for (int j = 0; j < 3; ++j) {
                const double Fij = x * Nq[3 * j + i] + y * Nq[9 + 3 * j + i] + Nq[18 + 3 * j + i];
                E[3 * i + j] = Fij * (i == 2 ? invf : 1.0) * (j == 2 ? invf : 1.0);
                nrm += E[3 * i + j] * E[3 * i + j];
            }
        nrm = 1.0 / sqrt(nrm);
        MDRP_ROLLED for (int e = 0; e < 9; ++e) E[e] *= nrm;
        double b1[6][3], b2[6][3]; // bearings K^-1 x, unit length
        MDRP_ROLLED for (int p = 0; p < 6; ++p) {
            const double a0 = x1h[p][0] * invf, a1 = x1h[p][1] * invf, a2 = x1h[p][2];
            const double c0 = x2h[p][0] * invf, c1 = x2h[p][1] * invf, c2 = x2h[p][2];
            const double na = 1.0 / sqrt(a0 * a0 + a1 * a1 + a2 * a2), nb = 1.0 / sqrt(c0 * c0 + c1 * c1 + c2 * c2);
            b1[p][0] = a0 * na; b1[p][1] = a1 * na; b1[p][2] = a2 * na;
            b2[p][0] = c0 * nb; b2[p][1] = c1 * nb; b2[p][2] = c2 * nb;
        }
        motion_from_essential_emit(E, b1, b2, 6, [&](const Model &m) {
            if (n_out < MAX_MODELS_6PT) { Model o = m; o.f1 = f; o.f2 = f; emit(o, n_out); ++n_out; }
        });
    }
    return n_out;
}
MDRP_HD int solver_relpose_6pt(const double (*x1h)[3], const double (*x2h)[3], Model *out /*[MAX_MODELS_6PT]*/) {
    PlainStore6 st;
    return solver_relpose_6pt_emit(x1h, x2h, st, [&](const Model &m, int k) { out[k] = m; });
}

#undef MDRP_ROLLED
// plain local storage (host tests; not for the device: these arrays would land in scratch memory)
struct Solve5Local {
    double C[100], lo[12], hi[12], ilo[10], ihi[10];
    int cc[12];
    MDRP_HD Solve5Store store() { return Solve5Store{C, 1, RootStack{lo, hi, cc, ilo, ihi, 1, 1}}; }
};

// ---------------------------------------------------------------- 7-point
// A fundamental matrix travels in the first nine doubles of a Model, row-major (q[0..3], t[0..2], scale, shift1).
MDRP_HD double *model_F(Model &m) { return m.q; }
MDRP_HD const double *model_F(const Model &m) { return m.q; }
MDRP_HD int solver_fundamental_7pt(const double (*x1h)[3], const double (*x2h)[3], Model *out /*[3]*/, double *A /*36 doubles, strided*/, int as) {
    double N[18];
    epipolar_nullspace<7, 3>(x1h, x2h, A, as, N); // columns 0..3 of the 9 x 7 constraint matrix in A
    const double *N0 = N, *N1 = N + 9;
    // det(r A + B), A = mat(N0), B = mat(N1), vec index of (i, j) = 3 j + i
    double c[4] = {0, 0, 0, 0};
    constexpr int PERM[6][4] = {{0, 1, 2, 1}, {0, 2, 1, -1}, {1, 0, 2, -1}, {1, 2, 0, 1}, {2, 0, 1, 1}, {2, 1, 0, -1}};
    for (int p = 0; p < 6; ++p) {
        const int js[3] = {PERM[p][0], PERM[p][1], PERM[p][2]};
        const double sgn = (double)PERM[p][3];
        double l1[3], l0[3];
        for (int i = 0; i < 3; ++i) { l1[i] = N0[3 * js[i] + i]; l0[i] = N1[3 * js[i] + i]; }
        const double q0 = l0[0] * l0[1], q1 = l0[0] * l1[1] + l1[0] * l0[1], q2 = l1[0] * l1[1];
        c[0] += sgn * q0 * l0[2];
        c[1] += sgn * (q0 * l1[2] + q1 * l0[2]);
        c[2] += sgn * (q1 * l1[2] + q2 * l0[2]);
        c[3] += sgn * q2 * l1[2];
    }
    // solve_cubic_real of the binary: closed form, roots in DESCENDING order (cos(phi), cos(phi - 2 pi / 3), cos(phi - 4 pi / 3))
    double roots[3];
    int nr = 0;
    if (c[3] != 0.0) {
        const double inv = 1.0 / c[3];
        nr = solve_cubic_real(c[2] * inv, c[1] * inv, c[0] * inv, roots[0], roots[1], roots[2]);
    }
    for (int s = 0; s < nr; ++s) {
        const double r = roots[s];
        double f[9], nrm = 0.0;
        for (int k = 0; k < 9; ++k) { f[k] = N0[k] * r + N1[k]; nrm += f[k] * f[k]; }
        nrm = 1.0 / sqrt(nrm);
        Model m;
        model_identity(m);
        double *F = model_F(m);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) F[3 * i + j] = f[3 * j + i] * nrm;
        m.shift2 = 0.0; m.f1 = 1.0; m.f2 = 1.0;
        out[s] = m;
    }
    return nr;
}

} // namespace mdrp
