// Host build of mdrp_amd/csrc/mdrp_math.h for CPU unit tests of the per-lane arithmetic (no GPU in the build
// container).  Test scaffolding only: the shipped library has no CPU path.
#include "../../mdrp_amd/csrc/mdrp_math.h"
#include <cstring>
using namespace mdrp;

extern "C" {
int hm_solver(int solver, const double *x1h, const double *x2h, const double *d1, const double *d2, double *out /*4*12*/) {
    Sample3 s;
    for (int i = 0; i < 3; ++i) {
        s.x1[i][0] = x1h[3 * i]; s.x1[i][1] = x1h[3 * i + 1];
        s.x2[i][0] = x2h[3 * i]; s.x2[i][1] = x2h[3 * i + 1];
        s.d1[i] = d1[i]; s.d2[i] = d2[i];
    }
    Model m[4];
    int n = run_solver(solver, s, m);
    std::memcpy(out, m, sizeof(Model) * (n > 0 ? n : 0));
    return n;
}
void hm_point(int focal, const double *model12, double scale_reproj, const double *x1, const double *x2, double d1, double d2,
              double *r7, double *J55) {
    Model m;
    std::memcpy(&m, model12, sizeof m);
    LmState st;
    lm_state_from_model(m, focal != 0, st);
    double J[5][LM_NPAR];
    point_residuals<true, true>(st, sqrt(scale_reproj), x1[0], x1[1], x2[0], x2[1], d1, d2, r7, r7[5], r7[6], J);
    std::memcpy(J55, J, sizeof J);
}
void hm_step(int focal, int est_shift, const double *model12, const double *delta11, double *out12) {
    Model m, o;
    std::memcpy(&m, model12, sizeof m);
    lm_apply_step(m, delta11, focal != 0, est_shift != 0, o);
    std::memcpy(out12, &o, sizeof o);
}
void hm_chol9(const double *A, const double *b, double *x) { chol_solve<9>(A, b, x); }
void hm_draw(uint64_t n, uint64_t seed, int count, uint32_t *out) {
    uint64_t st = seed;
    for (int i = 0; i < count; ++i) draw_sample3(n, st, out[3 * i], out[3 * i + 1], out[3 * i + 2]);
}
double hm_loss(int type, double thr, double r2, int weight) { return weight ? loss_weight(type, thr, r2) : loss_value(type, thr, r2); }
double hm_loss_mu(int type, double thr, double r2, double mu) { return loss_weight(type, thr, r2, mu); }
}

// host emulation of k_samples' wave-speculative table generation (64 "lanes" per step) for the CPU test
extern "C" void hm_draw_wave(uint64_t n, uint64_t seed, int count, uint32_t *out, uint64_t *state_out) {
    const uint64_t GAMMA = 0x9e3779b97f4a7c15ULL;
    uint64_t state = seed;
    int done = 0;
    while (done < count) {
        uint64_t s_end[64];
        uint32_t smp[64][3];
        int first = 63;
        bool found = false;
        for (int lane = 0; lane < 64; ++lane) {
            uint64_t s = state + (uint64_t)(3 * lane) * GAMMA;
            const uint64_t s0 = s;
            draw_sample3(n, s, smp[lane][0], smp[lane][1], smp[lane][2]);
            s_end[lane] = s;
            if (!found && (s - s0) != 3 * GAMMA) { first = lane; found = true; }
        }
        int nvalid = first + 1 < count - done ? first + 1 : count - done;
        for (int lane = 0; lane < nvalid; ++lane)
            for (int k = 0; k < 3; ++k) out[3 * (done + lane) + k] = smp[lane][k];
        state = s_end[nvalid - 1];
        done += nvalid;
    }
    *state_out = state;
}

// conservative fp32 pre-filter of the sweep: returns the number of records the EXACT fp64 test accepts but the filter
// drops (must be 0); *kept = records the filter keeps, *exact = records the exact test accepts
extern "C" long hm_filter_check(const double *E, const double *x1, const double *x2, long n, double thr, long *kept, long *exact) {
    double box[4] = {0, 0, 0, 0};
    for (long i = 0; i < n; ++i) {
        box[0] = fmax(box[0], fabs(x1[2 * i])); box[1] = fmax(box[1], fabs(x1[2 * i + 1]));
        box[2] = fmax(box[2], fabs(x2[2 * i])); box[3] = fmax(box[3], fabs(x2[2 * i + 1]));
    }
    float Ef[9], tb;
    double dm;
    filter_setup(E, box, thr, Ef, tb, dm);
    long missed = 0, k = 0, ex = 0;
    for (long i = 0; i < n; ++i) {
        const double a = x1[2 * i], b = x1[2 * i + 1], c = x2[2 * i], d = x2[2 * i + 1];
        const bool keep = filter_keeps(Ef, tb, (float)a, (float)b, (float)c, (float)d);
        const long double e0 = (long double)E[0] * a + (long double)E[1] * b + E[2], e1 = (long double)E[3] * a + (long double)E[4] * b + E[5],
                          e2 = (long double)E[6] * a + (long double)E[7] * b + E[8];
        const long double g0 = (long double)E[0] * c + (long double)E[3] * d + E[6], g1 = (long double)E[1] * c + (long double)E[4] * d + E[7];
        const long double Cv = c * e0 + d * e1 + e2, den = e0 * e0 + e1 * e1 + g0 * g0 + g1 * g1;
        const bool inl = Cv * Cv < (long double)thr * (1.0L + 1e-12L) * den;
        k += keep; ex += inl;
        if (inl && !keep) ++missed;
    }
    *kept = k; *exact = ex;
    return missed;
}

// ---- k_count's filter, emulated: bf16 operands, exact products, fp32 accumulation in slot order, the clamp test.
// Returns the number of correspondences the EXACT fp64 test accepts but the filter calls definite outliers (must be 0);
// *kept = candidates of the filter, *exact = correspondences the exact test accepts.  order = 0: slots ascending,
// 1: descending, 2: pairwise tree — the hardware's accumulation order is not documented, the bound must hold for any.
extern "C" long hm_count_check(const double *E, const double *x1, const double *x2, long n, double thr, int order, long *kept, long *exact) {
    double box[4] = {0, 0, 0, 0};
    for (long i = 0; i < n; ++i) {
        box[0] = fmax(box[0], fabs(x1[2 * i])); box[1] = fmax(box[1], fabs(x1[2 * i + 1]));
        box[2] = fmax(box[2], fabs(x2[2 * i])); box[3] = fmax(box[3], fabs(x2[2 * i + 1]));
    }
    uint16_t eh[8], el[8], e8[3];
    float tb2;
    count_setup_scaled(E, box, thr, eh, el, e8, tb2);
    long missed = 0, k = 0, ex = 0;
    for (long i = 0; i < n; ++i) {
        const double a = x1[2 * i], b = x1[2 * i + 1], c = x2[2 * i], d = x2[2 * i + 1];
        double m[8];
        count_monomials(a, b, c, d, m);
        float prod[27];
        for (int j = 0; j < 8; ++j) {
            uint16_t mh, ml;
            bf16_split(m[j], mh, ml);
            prod[j] = bf16_value(eh[j]) * bf16_value(mh);       // exact in fp32 (8 x 8 significand bits)
            prod[8 + j] = bf16_value(eh[j]) * bf16_value(ml);
            prod[16 + j] = bf16_value(el[j]) * bf16_value(mh);
        }
        prod[24] = bf16_value(e8[0]); prod[25] = bf16_value(e8[1]); prod[26] = bf16_value(e8[2]);
        float Cs = 0.f;
        if (order == 0) for (int j = 0; j < 27; ++j) Cs += prod[j];
        else if (order == 1) for (int j = 26; j >= 0; --j) Cs += prod[j];
        else {
            float t[32];
            for (int j = 0; j < 32; ++j) t[j] = j < 27 ? prod[j] : 0.f;
            for (int w = 16; w >= 1; w >>= 1) for (int j = 0; j < w; ++j) t[j] += t[j + w];
            Cs = t[0];
        }
        const float dd = fmaf(-Cs, Cs, tb2);
        const bool keep = dd > 0.f; // clamp(dd) is 1 for every positive dd here (dd is either <= 0 or >= 2^17)
        if (dd > 0.f && dd < 1.f) return -1; // the scale must leave nothing between 0 and 1
        const long double e0 = (long double)E[0] * a + (long double)E[1] * b + E[2], e1 = (long double)E[3] * a + (long double)E[4] * b + E[5],
                          e2 = (long double)E[6] * a + (long double)E[7] * b + E[8];
        const long double g0 = (long double)E[0] * c + (long double)E[3] * d + E[6], g1 = (long double)E[1] * c + (long double)E[4] * d + E[7];
        const long double Cv = c * e0 + d * e1 + e2, den = e0 * e0 + e1 * e1 + g0 * g0 + g1 * g1;
        const bool inl = Cv * Cv < (long double)thr * (1.0L + 1e-12L) * den;
        k += keep; ex += inl;
        if (inl && !keep) ++missed;
    }
    *kept = k; *exact = ex;
    return missed;
}

// ---- k_bound's lower bound: sum over the correspondences of min(q, thr_dn), times (1 - BOUND_SLACK), against the exact
// MSAC score without cheirality (sum of min(r^2, thr)) in long double.  Returns 1 if lower bound <= exact (or the model is
// not judged), 0 otherwise; *lb, *exact_score out; *cnt_ub = correspondences with q below the (inflated) threshold.
extern "C" int hm_bound_check(const double *E, const double *x1, const double *x2, long n, double thr, double *lb, double *exact_score, long *cnt_ub, long *cnt_exact) {
    double box[4] = {0, 0, 0, 0};
    for (long i = 0; i < n; ++i) {
        box[0] = fmax(box[0], fabs(x1[2 * i])); box[1] = fmax(box[1], fabs(x1[2 * i + 1]));
        box[2] = fmax(box[2], fabs(x2[2 * i])); box[3] = fmax(box[3], fabs(x2[2 * i + 1]));
    }
    float Ef[9], eC, eD, thr_dn;
    const bool sane = bound_setup32(E, box, thr, Ef, eC, eD, thr_dn);
    const float thr_cnt = (float)(thr * (1.0 + 1e-5)) * (1.0f + 1e-6f);
    double total = 0;
    long double exact = 0;
    long cu = 0, ce = 0;
    for (long j0 = 0; j0 < n; j0 += 64) {
        float part = 0.f;
        for (long i = j0; i < n && i < j0 + 64; ++i) {
            const double a = x1[2 * i], b = x1[2 * i + 1], c = x2[2 * i], d = x2[2 * i + 1];
            const float q = bound_r2_32(Ef, eC, eD, (float)a, (float)b, (float)c, (float)d);
            part += q < thr_dn ? q : thr_dn;
            cu += q < thr_cnt;
            const long double e0 = (long double)E[0] * a + (long double)E[1] * b + E[2], e1 = (long double)E[3] * a + (long double)E[4] * b + E[5],
                              e2 = (long double)E[6] * a + (long double)E[7] * b + E[8];
            const long double g0 = (long double)E[0] * c + (long double)E[3] * d + E[6], g1 = (long double)E[1] * c + (long double)E[4] * d + E[7];
            const long double Cv = c * e0 + d * e1 + e2, den = e0 * e0 + e1 * e1 + g0 * g0 + g1 * g1;
            const long double r2 = Cv * Cv / den;
            if (r2 < (long double)thr) { exact += r2; ++ce; } else exact += thr;
        }
        total += (double)part;
    }
    *lb = total * (1.0 - BOUND_SLACK); *exact_score = (double)exact; *cnt_ub = cu; *cnt_exact = ce;
    if (!sane) return 1;
    return (*lb <= (double)exact) && (cu >= ce);
}

extern "C" int hm_first_chunk_wish(double r, int k) { return first_chunk_wish(r, k); }
// ---- non-monodepth baselines (mdrp_classic_math.h)
#include "../../mdrp_amd/csrc/mdrp_classic_math.h"
extern "C" {
int hm_relpose_5pt_E(const double *x1h, const double *x2h, double *out /*10*9*/) {
    double Es[10][9];
    Solve5Local loc;
    const int n = relpose_5pt_E(reinterpret_cast<const double(*)[3]>(x1h), reinterpret_cast<const double(*)[3]>(x2h), Es, loc.store());
    std::memcpy(out, Es, sizeof(double) * 9 * n);
    return n;
}
int hm_relpose_5pt(const double *x1h, const double *x2h, double *out /*10*12*/) {
    Model m[MAX_MODELS_5PT];
    Solve5Local loc;
    const int n = solver_relpose_5pt(reinterpret_cast<const double(*)[3]>(x1h), reinterpret_cast<const double(*)[3]>(x2h), m, loc.store());
    std::memcpy(out, m, sizeof(Model) * n);
    return n;
}
int hm_relpose_7pt(const double *x1h, const double *x2h, double *out /*3*12*/) {
    Model m[3];
    double A[63];
    const int n = solver_fundamental_7pt(reinterpret_cast<const double(*)[3]>(x1h), reinterpret_cast<const double(*)[3]>(x2h), m, A, 1);
    std::memcpy(out, m, sizeof(Model) * n);
    return n;
}
// the null space of M epipolar constraints, by the stored-matrix factorisation (which = 0) and by the one with trailing columns in registers (which = 1)
int hm_nullspace(int M, int which, const double *x1h, const double *x2h, double *N /*[9 - M][9]*/) {
    const double(*a)[3] = reinterpret_cast<const double(*)[3]>(x1h);
    const double(*b)[3] = reinterpret_cast<const double(*)[3]>(x2h);
    double A[63];
    if (which == 0) {
        if (M == 5) { epipolar_columns<5>(a, b, A, 1); fullpiv_nullspace<5>(A, 1, N); }
        else if (M == 7) { epipolar_columns<7>(a, b, A, 1); fullpiv_nullspace<7>(A, 1, N); }
        else return 1;
    } else {
        if (M == 5) epipolar_nullspace<5, 2>(a, b, A, 1, N);
        else if (M == 7) epipolar_nullspace<7, 3>(a, b, A, 1, N);
        else return 1;
    }
    return 0;
}
int hm_real_roots10(const double *c, double *roots) { return real_roots<10>(c, roots); }
int hm_real_roots10_fast(const double *c, double *roots) {
    Solve5Local loc;
    return real_roots_fast<10>(c, roots, loc.store().rs);
}
// the device's storage: the isolated intervals in the free end of the interval stack's own arrays (mdrp_classic.h lds_solve5_store)
int hm_real_roots10_fast_shared(const double *c, double *roots) {
    double lo[12], hi[12];
    int cc[12];
    return real_roots_fast<10>(c, roots, RootStack{lo, hi, cc, lo + 11, hi + 11, 1, -1});
}
}
