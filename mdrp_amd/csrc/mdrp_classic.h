// mdrp_classic.h — kernels of the non-monodepth baselines on the same phase-split LO-RANSAC:
//   kind 3  estimate_relative_pose @0x21f800      RelativePoseEstimator: 5-point samples, pose models, Sampson + cheirality
//   kind 5  estimate_fundamental   @0x221a00      FundamentalEstimator: 7-point samples, F models, Sampson
// The sampler, the three scoring sweeps (k_count on the matrix cores, k_bound, k_score), k_scan, k_lo_plan and k_walk are the
// monodepth path's own kernels (a fundamental matrix travels in the first nine doubles of a Model: RAWF instantiations);
// this file adds what differs: the normalisation (kc_prep), samples of K > 3 points (kc_samples), the minimal solvers
// (kc_solve, mdrp_classic_math.h), and the Sampson-only LM of refine_relpose @0x258f50 / refine_fundamental @0x2590d0 with
// the estimators' LO rules (kc_lo) and final stage (kc_final).
#pragma once
#include "mdrp_kernels.h"
#include "mdrp_classic_math.h"

namespace mdrp {

template <int CK> struct ClassicTraits;
template <> struct ClassicTraits<CLASSIC_RELPOSE> { static constexpr int K = 5, MPS = 12, MAXM = MAX_MODELS_5PT, NP = 5; static constexpr bool POSE = true; };
template <> struct ClassicTraits<CLASSIC_SHARED> { static constexpr int K = 6, MPS = 16, MAXM = MAX_MODELS_6PT, NP = 6; static constexpr bool POSE = false; };
template <> struct ClassicTraits<CLASSIC_FUND> { static constexpr int K = 7, MPS = 4, MAXM = 3, NP = 7; static constexpr bool POSE = false; };

// ------------------------------------------------------------------------------------------------ prep
// kind 3: Camera::unproject, threshold * (1/f1 + 1/f2) / 2 (estimate_relative_pose @0x21f800)
// kind 4: x <- (x - pp) / s with the same shared scale s (estimate_shared_focal_relative_pose @0x2205a0; pp in cam1[pair].p[0..1])
// kind 5: normalize_points(normalize_scale, normalize_centroid, shared_scale) @0x4f6ae0: x <- (x - centroid) / s,
//         s = sum(|x1 - c1| + |x2 - c2|) / (sqrt2 N); thresholds / s (estimate_fundamental @0x221a00)
MDRP_GLOBAL __launch_bounds__(256) void kc_prep(RunParams rp, const double *__restrict__ x1, const double *__restrict__ x2,
                                               const int32_t *__restrict__ n_per_pair, const int32_t *__restrict__ table_of_pair,
                                               const CamDev *__restrict__ cam1, const CamDev *__restrict__ cam2, double max_epi,
                                               double bundle_loss_scale, double *__restrict__ pts, PairState *__restrict__ st,
                                               uint4 *__restrict__ rfrag) {
    __shared__ double red[4][5];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int n = n_per_pair[pair];
    const size_t base = (size_t)pair * rp.n_max;
    double k = 1.0, norm = 1.0;
    double fx1 = 1, fy1 = 1, cx1 = 0, cy1 = 0, fx2 = 1, fy2 = 1, cx2 = 0, cy2 = 0;
    if (rp.kind == CLASSIC_RELPOSE) {
        const CamDev a = cam1[pair], b = cam2[pair];
        if (a.model_id == 1) { fx1 = a.p[0]; fy1 = a.p[1]; cx1 = a.p[2]; cy1 = a.p[3]; } else { fx1 = fy1 = a.p[0]; cx1 = a.p[1]; cy1 = a.p[2]; }
        if (b.model_id == 1) { fx2 = b.p[0]; fy2 = b.p[1]; cx2 = b.p[2]; cy2 = b.p[3]; } else { fx2 = fy2 = b.p[0]; cx2 = b.p[1]; cy2 = b.p[2]; }
        k = 0.5 * (1.0 / (0.5 * (fx1 + fy1)) + 1.0 / (0.5 * (fx2 + fy2)));
    } else {
        double s[4] = {0, 0, 0, 0};
        for (int i = tid; i < n; i += 256) {
            s[0] += x1[2 * (base + i)]; s[1] += x1[2 * (base + i) + 1]; s[2] += x2[2 * (base + i)]; s[3] += x2[2 * (base + i) + 1];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { s[q] = wave_sum(s[q]); if ((tid & 63) == 0) red[tid >> 6][q] = s[q]; }
        __syncthreads();
        const double inv_n = 1.0 / (double)(n > 0 ? n : 1);
        cx1 = (red[0][0] + red[1][0] + red[2][0] + red[3][0]) * inv_n; cy1 = (red[0][1] + red[1][1] + red[2][1] + red[3][1]) * inv_n;
        cx2 = (red[0][2] + red[1][2] + red[2][2] + red[3][2]) * inv_n; cy2 = (red[0][3] + red[1][3] + red[2][3] + red[3][3]) * inv_n;
        if (rp.kind == CLASSIC_SHARED) { // estimate_shared_focal_relative_pose: the caller's principal point instead of the centroids
            const CamDev a = cam1[pair];
            cx1 = cx2 = a.p[0]; cy1 = cy2 = a.p[1];
        }
        double acc = 0;
        for (int i = tid; i < n; i += 256) {
            const double a = x1[2 * (base + i)] - cx1, b = x1[2 * (base + i) + 1] - cy1, c = x2[2 * (base + i)] - cx2, d = x2[2 * (base + i) + 1] - cy2;
            acc += sqrt(a * a + b * b) + sqrt(c * c + d * d);
        }
        acc = wave_sum(acc);
        if ((tid & 63) == 0) red[tid >> 6][4] = acc;
        __syncthreads();
        norm = (red[0][4] + red[1][4] + red[2][4] + red[3][4]) / (1.4142135623730951 * (double)(n > 0 ? n : 1));
        k = 1.0 / norm;
        fx1 = fy1 = fx2 = fy2 = norm;
    }
    double bx[4] = {0, 0, 0, 0};
    for (int i = tid; i < n; i += 256) {
        const double a = (x1[2 * (base + i)] - cx1) / fx1, b = (x1[2 * (base + i) + 1] - cy1) / fy1;
        const double c = (x2[2 * (base + i)] - cx2) / fx2, d = (x2[2 * (base + i) + 1] - cy2) / fy2;
        bx[0] = fmax(bx[0], fabs(a)); bx[1] = fmax(bx[1], fabs(b)); bx[2] = fmax(bx[2], fabs(c)); bx[3] = fmax(bx[3], fabs(d));
        double *p = pts + (base + i) * PT_STRIDE;
        p[0] = a; p[1] = b; p[2] = c; p[3] = d;
        p[4] = 1.0 / sqrt(a * a + b * b + 1.0);
        p[5] = 1.0 / sqrt(c * c + d * d + 1.0);
        if (rfrag) store_record_fragment(rfrag + (size_t)pair * ((rp.n_max + 15) / 16) * 64, i, a, b, c, d);
    }
    if (rfrag) {
        const int g_end = ((n + 15) / 16) * 16;
        for (int i = n + tid; i < g_end; i += 256) clear_record_fragment(rfrag + (size_t)pair * ((rp.n_max + 15) / 16) * 64, i);
    }
    __shared__ double redbox[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = bx[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
        if ((tid & 63) == 0) redbox[tid >> 6][q] = v;
    }
    __syncthreads();
    if (tid == 0) {
        PairState s;
#pragma unroll
        for (int q = 0; q < 4; ++q) s.box[q] = fmax(fmax(redbox[0][q], redbox[1][q]), fmax(redbox[2][q], redbox[3][q]));
        const bool ok = n >= rp.sample_sz;
        s.n = ok ? n : 0;
        s.table = table_of_pair[pair];
        s.active = ok;
        s.n_triggers = 0;
        s.eps = max_epi * k;
        s.sq_thr = s.eps * s.eps;
        s.scale_reproj = 0.0;
        s.lo_loss_scale = s.eps;
        s.final_loss_scale = bundle_loss_scale * k;
        s.norm = norm;
        s.cen[0] = cx1; s.cen[1] = cy1; s.cen[2] = cx2; s.cen[3] = cy2;
        s.best_min_cnt = 0; s.best_min_score = DBL_MAX;
        s.dyn_max_iter = rp.max_iterations;
        s.iterations = 0; s.refinements = 0; s.num_inliers = 0;
        s.inlier_ratio = 0.0; s.model_score = DBL_MAX;
        model_identity(s.best);
        if (rp.kind == CLASSIC_FUND) { double *F = model_F(s.best); for (int q = 0; q < 9; ++q) F[q] = (q % 4 == 0) ? 1.0 : 0.0; }
        if (rp.score_initial && ok && rp.kind != CLASSIC_FUND) { // the reset identity pose has E = 0: no inliers, score N eps^2, one LO
            s.best_min_score = s.sq_thr * (double)n;
            s.model_score = s.best_min_score;
            s.refinements = 1;
        }
        st[pair] = s;
    }
}

// ------------------------------------------------------------------------------------------------ samples of K points
// k_samples' wave-speculative scheme with K raw draws per sample (RandomSampler @0x4f8970, draw_sample @0x4f87f0)
template <int K>
__device__ __forceinline__ void draw_sample_k(uint64_t n, uint64_t &state, uint32_t *out) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        bool dup;
        do {
            out[i] = (uint32_t)((uint64_t)(int64_t)splitmix_int(state) % n);
            dup = false;
#pragma unroll
            for (int j = 0; j < K; ++j) dup = dup || (j < i && out[j] == out[i]);
        } while (dup);
    }
}
template <int K>
__global__ __launch_bounds__(SAMP_THREADS) void kc_samples(int n_tables, const int32_t *__restrict__ table_n, uint64_t *__restrict__ table_state,
                                                           int chunk_len, uint32_t *__restrict__ samples /*[n_tables][chunk_len][K]*/) {
    const int t = blockIdx.x;
    if (t >= n_tables) return;
    const uint64_t n = (uint64_t)table_n[t];
    if (n < (uint64_t)K) return;
    uint64_t state = table_state[t];
    __syncthreads(); // every thread holds the state before thread 0 advances it
    samples_block<K>(n, state, chunk_len, samples + (size_t)t * chunk_len * K, [](uint64_t n_, uint64_t &s_, uint32_t *o) { draw_sample_k<K>(n_, s_, o); });
    if (threadIdx.x == 0) table_state[t] = state;
}

// LDS storage of the 5-point solver: columns 2..9 of the 10 x 10 matrix of its LU factorisation (columns 0 and 1 are in registers,
// mdrp_classic_math.h relpose_5pt_eliminate), one 64-lane column per element, and — in the same bytes, once the factorisation is dead —
// the root finder's interval stack (dynamic LDS of the launch: SOLVE5_LDS_BYTES = 40 KB; one wavefront per workgroup, four workgroups per CU)
constexpr size_t SOLVE5_LDS_BYTES = (size_t)64 * 80 * sizeof(double);
constexpr size_t SOLVE7_LDS_BYTES = (size_t)64 * 36 * sizeof(double); // columns 0..3 of the 9 x 7 constraint matrix of the 7-point null space (4..6: registers)
__device__ __forceinline__ Solve5Store lds_solve5_store() {
    extern __shared__ double solve5_lds[];
    const int lane = threadIdx.x & 63;
    double *C = solve5_lds + lane;
    double *lo = solve5_lds + lane, *hi = lo + 12 * 64;
    int *cc = reinterpret_cast<int *>(solve5_lds + 24 * 64) + lane; // 12 ints per lane = 6 double columns
    return Solve5Store{C, 64, RootStack{lo, hi, cc, lo + 11 * 64, hi + 11 * 64, 64, -64}}; // isolated intervals: down from the stack's last entry
}

// ------------------------------------------------------------------------------------------------ 5-point solver in two kernels (round 6)
// kc_solve<CLASSIC_RELPOSE> did everything per sample in one kernel: the elimination (10 x 10 LU in LDS, R in 200 registers) fixed it at 512 registers
// and 51 KB of LDS per wavefront = three wavefronts per CU for its whole length, while four fifths of its instructions (Sturm isolation of the
// degree-10 polynomial's roots, their polishing, the decomposition of up to ten essential matrices) need neither.
//   kc_solve5_null     one lane per sample: the null space of the five epipolar constraints as linear polynomials (36 doubles)
//   kc_solve5_reduce   one lane per sample: ten cubic constraints, LU (40 KB of LDS, 512 registers: four wavefronts per CU), the three rows of B(z)
//                      (39 doubles)  -> with the null space: Reduce5 in global
//                      memory, lane-interleaved per 64 samples (element k of lane l at [(76 block + k) 64 + l]: every access is one 512-byte row)
//   kc_solve5_roots    one lane per sample: det B(z), its real roots, (x, y, z) per root parked in LDS (the interval stack is dead by then);
//                      then the wavefront's solutions (0-10 per sample, ~3 on average) are decomposed by whichever lane is free, 64 at a time
//                      (round 4's pooling: the lock-step version ran max-over-lanes trips with a third of the lanes active): essential matrix from
//                      the owner's null space (read back from the Reduce5 block), motion_from_essential, 4 x 5 cheirality tests, model store.  A pose
//                      goes into slot `root index` of its sample (at most one decomposition of an essential matrix has all five points in front of
//                      both cameras): k_scan walks the slots of an iteration in order and skips empty ones, so the order of the reference is kept
//                      without counting.  17.5 KB of LDS per wavefront (eight per CU; 25.6 KB = six until the isolated intervals moved into the stack's arrays).
// Same expressions, same order as the one-kernel solver; the last bits of the elimination depend on whether the compiler sees where the null space
// comes from (-ffp-contract=fast), which is why the inlined path (unit entry point, tests) hides it: relpose_5pt_reduce, tools/ubench/solve5_split_bits.hip.
constexpr int RED5_STRIDE = REDUCE5_DOUBLES + 1;                                        // + 1: "the elimination succeeded"
// the root finder's stack and isolated intervals: lo 12 | hi 12 | cc 6 columns; then the solutions: 30 columns and 640 codes (5): 17.5 KB, eight wavefronts per CU
constexpr size_t SOLVE5B_LDS_BYTES = (size_t)64 * 35 * sizeof(double);
__device__ __forceinline__ void gather5(const uint32_t *__restrict__ sm, const double *__restrict__ pts_pair, double (*x1h)[3], double (*x2h)[3]) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double *p = pts_pair + (size_t)sm[k] * PT_STRIDE;
        const double2 p01 = *reinterpret_cast<const double2 *>(p), p23 = *reinterpret_cast<const double2 *>(p + 2),
                      p45 = *reinterpret_cast<const double2 *>(p + 4);
        x1h[k][0] = p01.x * p45.x; x1h[k][1] = p01.y * p45.x; x1h[k][2] = p45.x;
        x2h[k][0] = p23.x * p45.y; x2h[k][1] = p23.y * p45.y; x2h[k][2] = p45.y;
    }
}

// (the null space first, in a kernel of its own: a full-pivoting Householder QR of the 9 x 5 constraint matrix, three of its columns in 14 KB of
// LDS and two in registers — eight wavefronts per CU instead of the four the elimination behind it is held to)
constexpr size_t SOLVE5N_LDS_BYTES = (size_t)64 * 27 * sizeof(double); // columns 0..2 of the 9 x 5 constraint matrix (3, 4: registers)
MDRP_GLOBAL __launch_bounds__(64, 2) void kc_solve5_null(RunParams rp, const PairState *__restrict__ st, const uint32_t *__restrict__ samples,
                                                           const double *__restrict__ pts, double *__restrict__ red /*[pair][block][RED5_STRIDE][64]*/) {
    extern __shared__ double solve5_lds[];
    const int pair = blockIdx.y, lane = threadIdx.x;
    const int it = blockIdx.x * 64 + lane;
    const PairState &ps = st[pair];
    if (!ps.active || it >= rp.chunk_len) return;
    double x1h[5][3], x2h[5][3], El[3][3][4];
    gather5(samples + ((size_t)ps.table * rp.chunk_len + it) * 5, pts + (size_t)pair * rp.n_max * PT_STRIDE, x1h, x2h);
    relpose_5pt_nullspace(x1h, x2h, solve5_lds + lane, 64, El);
    double *dst = red + ((size_t)pair * gridDim.x + blockIdx.x) * RED5_STRIDE * 64 + lane;
    const double *src = &El[0][0][0];
#pragma unroll
    for (int k = 0; k < 36; ++k) dst[(size_t)k * 64] = src[k];
}

MDRP_GLOBAL __launch_bounds__(64) void kc_solve5_reduce(RunParams rp, const PairState *__restrict__ st, double *__restrict__ red /*[pair][block][RED5_STRIDE][64]*/) {
    const int pair = blockIdx.y, lane = threadIdx.x;
    const int it = blockIdx.x * 64 + lane;
    const PairState &ps = st[pair];
    if (!ps.active || it >= rp.chunk_len) return;
    double *blk = red + ((size_t)pair * gridDim.x + blockIdx.x) * RED5_STRIDE * 64 + lane;
    Reduce5 r5;
    static_assert(sizeof(Reduce5) == REDUCE5_DOUBLES * sizeof(double), "Reduce5 is 75 packed doubles");
    {
        double *de = &r5.El[0][0][0];
#pragma unroll
        for (int k = 0; k < 36; ++k) de[k] = blk[(size_t)k * 64];
    }
    const bool ok = relpose_5pt_eliminate(lds_solve5_store(), r5);
    const double *src = &r5.bx[0][0];
#pragma unroll
    for (int k = 36; k < REDUCE5_DOUBLES; ++k) blk[(size_t)k * 64] = src[k - 36];
    blk[(size_t)REDUCE5_DOUBLES * 64] = ok ? 1.0 : 0.0;
}

MDRP_GLOBAL __launch_bounds__(64, 2) void kc_solve5_roots(RunParams rp, const PairState *__restrict__ st, const uint32_t *__restrict__ samples,
                                                            const double *__restrict__ pts, const double *__restrict__ red, Model *__restrict__ models,
                                                            int32_t *__restrict__ slot_inl, uint32_t *__restrict__ tags, int32_t *__restrict__ model_count) {
    constexpr int MPS = ClassicTraits<CLASSIC_RELPOSE>::MPS;
    extern __shared__ double solve5_lds[];
    const int pair = blockIdx.y, lane = threadIdx.x;
    const int it = blockIdx.x * 64 + lane;
    const PairState &ps = st[pair];
    if (!ps.active) return;
    const bool live = it < rp.chunk_len;
    const size_t slot0 = (size_t)pair * rp.slot_stride + (size_t)(rp.chunk_off + it) * MPS;
    const size_t tag_base = (size_t)pair * rp.slot_stride;
    const double *blk = red + ((size_t)pair * gridDim.x + blockIdx.x) * RED5_STRIDE * 64;
    const double *pts_pair = pts + (size_t)pair * rp.n_max * PT_STRIDE;
    int ne = 0;
    if (live) {
        if (blk[(size_t)REDUCE5_DOUBLES * 64 + lane] != 0.0) {
            double bx[3][4], by[3][4], b1[3][5];
            {
                const double *src = blk + (size_t)36 * 64 + lane; // behind El
                double *dx = &bx[0][0], *dy = &by[0][0], *d1 = &b1[0][0];
#pragma unroll
                for (int k = 0; k < 12; ++k) dx[k] = src[(size_t)k * 64];
#pragma unroll
                for (int k = 0; k < 12; ++k) dy[k] = src[(size_t)(12 + k) * 64];
#pragma unroll
                for (int k = 0; k < 15; ++k) d1[k] = src[(size_t)(24 + k) * 64];
            }
            // (x, y) of a root are computed after the root finder has returned: its interval stack is dead, the solutions are parked straight in this
            // lane's LDS column for whichever lane decomposes them
            relpose_5pt_roots_xyz(bx, by, b1, lds_solve5_store().rs, [&](double x, double y, double z) {
                if (ne < MAX_MODELS_5PT) { solve5_lds[(3 * ne) * 64 + lane] = x; solve5_lds[(3 * ne + 1) * 64 + lane] = y; solve5_lds[(3 * ne + 2) * 64 + lane] = z; ++ne; }
            });
        }
#pragma unroll
        for (int g = 0; g < MPS / 4; ++g) *reinterpret_cast<int4 *>(slot_inl + slot0 + 4 * g) = make_int4(-1, -1, -1, -1);
    }
    int pre = ne;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(pre, o, 64);
        if (lane >= o) pre += v;
    }
    const int total = __shfl(pre, 63, 64);
    int *codes = reinterpret_cast<int *>(solve5_lds + 30 * 64); // 640 items: ten more columns behind the thirty of the solutions
    for (int r = 0; r < ne; ++r) codes[pre - ne + r] = lane * 16 + r;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i0 = 0; i0 < total; i0 += 64) {
        const int i = i0 + lane;
        bool found = false;
        int src_it = 0, root = 0;
        if (i < total) {
            const int code = codes[i];
            const int src = code >> 4;
            root = code & 15;
            src_it = blockIdx.x * 64 + src;
            double El[3][3][4], e[9], x1h[5][3], x2h[5][3];
            {
                const double *sp = blk + src;
                double *de = &El[0][0][0];
#pragma unroll
                for (int k = 0; k < 36; ++k) de[k] = sp[(size_t)k * 64];
            }
            essential_from_xyz(El, solve5_lds[(3 * root) * 64 + src], solve5_lds[(3 * root + 1) * 64 + src], solve5_lds[(3 * root + 2) * 64 + src], e);
            gather5(samples + ((size_t)ps.table * rp.chunk_len + src_it) * 5, pts_pair, x1h, x2h);
            const size_t dst = (size_t)pair * rp.slot_stride + (size_t)(rp.chunk_off + src_it) * MPS + root;
            motion_from_essential_emit(e, x1h, x2h, 5, [&](const Model &m) {
                if (!found) { models[dst] = m; slot_inl[dst] = -2; found = true; }
                // (a second decomposition with all five points in front of both cameras would need exactly singular geometry)
            });
        }
        const unsigned long long fb = __ballot(found);
        if (fb) {
            int base = 0;
            const int first = __ffsll((long long)fb) - 1;
            if (lane == first) base = atomicAdd(&model_count[2 * pair], __popcll(fb));
            base = __shfl(base, first, 64);
            if (found) tags[tag_base + base + __popcll(fb & ((1ull << lane) - 1ull))] = (uint32_t)((rp.chunk_off + src_it) * MPS + root);
        }
    }
}

// ------------------------------------------------------------------------------------------------ solve
// One lane per minimal sample (64-lane workgroups: the solvers live in scratch-backed arrays and their trip counts
// diverge with the number of real roots).  Same slot / tag conventions as k_solve with MPS slots per sample.
template <int CK>
__global__ __launch_bounds__(64) void kc_solve(RunParams rp, const PairState *__restrict__ st, const uint32_t *__restrict__ samples,
                                               const double *__restrict__ pts, Model *__restrict__ models, int32_t *__restrict__ slot_inl,
                                               uint32_t *__restrict__ tags, int32_t *__restrict__ model_count) {
    using TR = ClassicTraits<CK>;
    constexpr int K = TR::K, MPS = TR::MPS, MAXM = TR::MAXM;
    const int pair = blockIdx.y;
    const int it = blockIdx.x * 64 + threadIdx.x;
    const PairState &ps = st[pair];
    if (!ps.active) return;
    const bool live = it < rp.chunk_len;
    int n = 0;
    Model out[CK == CLASSIC_FUND ? MAXM : 1]; // the 5- and 6-point solvers write their poses straight into their slots
    const size_t slot0 = (size_t)pair * rp.slot_stride + (size_t)(rp.chunk_off + it) * MPS;
    const int lane = threadIdx.x & 63;
    const size_t tag_base = (size_t)pair * rp.slot_stride;
    auto gather = [&](int it_, double (*x1h)[3], double (*x2h)[3]) { // unit bearings of sample `it_` of this pair
        const uint32_t *sm = samples + ((size_t)ps.table * rp.chunk_len + it_) * K;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double *p = pts + ((size_t)pair * rp.n_max + sm[k]) * PT_STRIDE;
            const double2 p01 = *reinterpret_cast<const double2 *>(p), p23 = *reinterpret_cast<const double2 *>(p + 2),
                          p45 = *reinterpret_cast<const double2 *>(p + 4);
            x1h[k][0] = p01.x * p45.x; x1h[k][1] = p01.y * p45.x; x1h[k][2] = p45.x;
            x2h[k][0] = p23.x * p45.y; x2h[k][1] = p23.y * p45.y; x2h[k][2] = p45.y;
        }
    };
    static_assert(CK != CLASSIC_RELPOSE, "the 5-point solver runs as kc_solve5_reduce + kc_solve5_roots");
    if (live) {
        double x1h[K][3], x2h[K][3];
        gather(it, x1h, x2h);
        if (CK == CLASSIC_SHARED) { PlainStore6 st6; n = solver_relpose_6pt_emit(x1h, x2h, st6, [&](const Model &m, int k) { models[slot0 + k] = m; }); }
        else { extern __shared__ double solve5_lds[]; n = solver_fundamental_7pt(x1h, x2h, out, solve5_lds + (threadIdx.x & 63), 64); }
    }
    int pre = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(pre, o, 64);
        if (lane >= o) pre += v;
    }
    const int tot = __shfl(pre, 63, 64);
    int base = 0;
    if (lane == 63 && tot > 0) base = atomicAdd(&model_count[2 * pair], tot);
    base = __shfl(base, 63, 64);
    if (!live) return;
    int pos = base + pre - n;
#pragma unroll
    for (int g = 0; g < MPS / 4; ++g)
        *reinterpret_cast<int4 *>(slot_inl + slot0 + 4 * g) =
            make_int4(n > 4 * g ? -2 : -1, n > 4 * g + 1 ? -2 : -1, n > 4 * g + 2 ? -2 : -1, n > 4 * g + 3 ? -2 : -1);
    for (int k = 0; k < n; ++k) {
        if (CK == CLASSIC_FUND) models[slot0 + k] = out[k];
        tags[tag_base + pos] = (uint32_t)((rp.chunk_off + it) * MPS + k);
        ++pos;
    }
}

// ------------------------------------------------------------------------------------------------ Sampson-only LM
// refine_relpose @0x258f50: 5 parameters — R <- R exp([w]x), t <- t + B d with B an orthonormal basis of the tangent plane of t
//   (set up from the CURRENT t at every accumulate);  refine_fundamental @0x2590d0: F = U diag(1, sigma, 0) V' with rotations
//   U, V: U <- exp([a]x) U, V <- exp([b]x) V, sigma <- sigma + d.  Inside the LM a factorised F lives in the Model as
//   q = qU, (t[0], t[1], t[2], scale) = qV, shift1 = sigma.
template <int CK>
struct ClmState {
    LmState st;      // kinds 3 / 4
    double tb[6];
    double F[9], u1[3], v1[3]; // kind 5
};

__device__ __forceinline__ void clm_tangent_basis(const double *t, double *tb) {
    double e[3] = {0, 0, 0};
    if (fabs(t[0]) < fabs(t[1])) { if (fabs(t[0]) < fabs(t[2])) e[0] = 1; else e[2] = 1; }
    else { if (fabs(t[1]) < fabs(t[2])) e[1] = 1; else e[2] = 1; }
    cross3(t, e, tb);
    double n = 1.0 / sqrt(dot3(tb, tb));
#pragma unroll
    for (int i = 0; i < 3; ++i) tb[i] *= n;
    cross3(tb, t, tb + 3);
    n = 1.0 / sqrt(dot3(tb + 3, tb + 3));
#pragma unroll
    for (int i = 0; i < 3; ++i) tb[3 + i] *= n;
}

template <int CK>
__device__ __forceinline__ void clm_setup(const Model &m, ClmState<CK> &s) {
    if (CK == CLASSIC_FUND) {
        double U[9], V[9];
        const double qV[4] = {m.t[0], m.t[1], m.t[2], m.scale};
        quat_to_R(m.q, U);
        quat_to_R(qV, V);
#pragma unroll
        for (int i = 0; i < 3; ++i) { s.u1[i] = U[3 * i + 1]; s.v1[i] = V[3 * i + 1]; }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) s.F[3 * i + j] = U[3 * i] * V[3 * j] + m.shift1 * U[3 * i + 1] * V[3 * j + 1];
    } else {
        lm_state_from_model(m, CK == CLASSIC_SHARED, s.st);
        clm_tangent_basis(m.t, s.tb);
    }
}

// residual r = C / |J_C| and (WITH_J) the Jacobian row of the NP parameters
template <int CK, bool WITH_J>
__device__ __forceinline__ double clm_point(const ClmState<CK> &s, double a, double b, double c, double d, double *J) {
    if (CK != CLASSIC_FUND) {
        double r0, J0[LM_NPAR];
        lm_sampson_term<WITH_J, CK == CLASSIC_SHARED>(s.st, a, b, c, d, r0, J0);
        if (WITH_J) {
            J[0] = J0[0]; J[1] = J0[1]; J[2] = J0[2];
            J[3] = s.tb[0] * J0[3] + s.tb[1] * J0[4] + s.tb[2] * J0[5];
            J[4] = s.tb[3] * J0[3] + s.tb[4] * J0[4] + s.tb[5] * J0[5];
            if (CK == CLASSIC_SHARED) J[5] = J0[9] + J0[10];
        }
        return r0;
    }
    const double *F = s.F;
    const double Fh1_0 = F[0] * a + F[1] * b + F[2], Fh1_1 = F[3] * a + F[4] * b + F[5], Fh1_2 = F[6] * a + F[7] * b + F[8];
    const double Ft2_0 = F[0] * c + F[3] * d + F[6], Ft2_1 = F[1] * c + F[4] * d + F[7];
    const double C = c * Fh1_0 + d * Fh1_1 + Fh1_2;
    const double den = Fh1_0 * Fh1_0 + Fh1_1 * Fh1_1 + Ft2_0 * Ft2_0 + Ft2_1 * Ft2_1;
    const double isd = 1.0 / sqrt(den);
    const double r0 = C * isd;
    if (!WITH_J) return r0;
    const double h1[3] = {a, b, 1.0}, h2[3] = {c, d, 1.0};
    const double Fh1[3] = {Fh1_0, Fh1_1, Fh1_2}, Ft2[3] = {Ft2_0, Ft2_1, 0.0};
    const double k = C * isd * isd * isd;
    double G[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double g = h2[i] * h1[j] * isd;
            if (i < 2) g -= k * Fh1[i] * h1[j];
            if (j < 2) g -= k * Ft2[j] * h2[i];
            G[3 * i + j] = g;
        }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int q = (p + 1) % 3, w = (p + 2) % 3;
        double au = 0, av = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            au += -G[3 * q + j] * F[3 * w + j] + G[3 * w + j] * F[3 * q + j]; // [e_p]x F : row q = -F row w, row w = +F row q
            av += -G[3 * j + q] * F[3 * j + w] + G[3 * j + w] * F[3 * j + q]; // F [e_p]x': col q = -F col w, col w = +F col q
        }
        J[p] = au; J[3 + p] = av;
    }
    double gs = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) gs += G[3 * i + j] * s.u1[i] * s.v1[j];
    J[6] = gs;
    return r0;
}

// Work lists of the classic LM (round 4, the scheme of mdrp_kernels.h lm_cost / lm_accumulate): with a truncated loss a record beyond the
// threshold has IRLS weight zero and adds nothing to J'J, but inside a wavefront its lanes ride along through the ~300-instruction Jacobian
// code whenever one lane contributes.  The cost sweep — which runs for every candidate anyway — appends the indices of the contributing
// records to an LDS list, compacted per wavefront with a ballot; the normal-equation sweep of an accepted model walks its list with every
// lane busy.  Two lists: the current model's and the candidate's.  stride == 0 (pairs beyond LM_LIST_MAX_N records, the unit entry point): off.
struct ClmList {
    uint16_t *list; // dynamic LDS, 2 * stride entries
    int stride;
    int count[2][4];
};

template <int CK, int T>
__device__ double clm_cost(const Model &m, const double *__restrict__ pts, int n, const uint8_t *__restrict__ mask, const LmOpt &o,
                           double *scratch, ClmList &cl, int buf) {
    ClmState<CK> s;
    clm_setup<CK>(m, s);
    double cost = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool use_list = cl.stride > 0;
    // wavefront w sweeps the records [w * seg, (w + 1) * seg) in trips of 64; the next record is requested before the current one is consumed
    // (unconditional loads from a clamped index): a trip is ~50 instructions, the records come from L2 or beyond
    const int seg = ((n + T - 1) / T) * 64, lo = wave * seg, hi = n < lo + seg ? n : lo + seg;
    lds_u16 *list = lds_cast(cl.list) + (size_t)buf * cl.stride;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int last = n > 0 ? n - 1 : 0;
    auto fetch = [&](int i, double2 &p01, double2 &p23, bool &ok) {
        const int ic = i < last ? i : last;
        ok = (int)(i < hi) & (int)(mask ? mask[ic] != 0 : true);
        const double2 *P = reinterpret_cast<const double2 *>(pts + (size_t)ic * PT_STRIDE);
        p01 = P[0]; p23 = P[1];
    };
    double2 c01, c23;
    bool cok;
    int cnt = 0;
    fetch(lo + lane, c01, c23, cok);
    for (int base = lo; base < hi; base += 64) {
        double2 n01, n23;
        bool nok;
        fetch(base + 64 + lane, n01, n23, nok);
        bool contrib = false;
        if (cok) {
            const double r = clm_point<CK, false>(s, c01.x, c01.y, c23.x, c23.y, nullptr);
            cost += loss_value(o.loss, o.loss_scale, r * r);
            contrib = loss_weight(o.loss, o.loss_scale, r * r, o.mu) != 0.0;
        }
        if (use_list) {
            const unsigned long long ball = __ballot(contrib);
            if (contrib) list[lo + cnt + __popcll(ball & lt)] = (uint16_t)(base + lane);
            cnt += __popcll(ball);
        }
        c01 = n01; c23 = n23; cok = nok;
    }
    if (use_list && lane == 0) cl.count[buf][wave] = cnt;
    double v[1] = {cost};
    block_sum<1, T>(v, scratch);
    return v[0];
}

template <int CK, int T>
__device__ void clm_accumulate(const Model &m, const double *__restrict__ pts, int n, const uint8_t *__restrict__ mask, const LmOpt &o,
                               double *acc, double *tb_out, double *scratch, ClmList &cl, int buf) {
    constexpr int NP = ClassicTraits<CK>::NP, NT = NP * (NP + 1) / 2;
    ClmState<CK> s;
    clm_setup<CK>(m, s);
#pragma unroll
    for (int i = 0; i < 6; ++i) tb_out[i] = (CK == CLASSIC_FUND) ? 0.0 : s.tb[i];
#pragma unroll
    for (int i = 0; i < NT + NP; ++i) acc[i] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = ((n + T - 1) / T) * 64, lo = wave * seg, hi = n < lo + seg ? n : lo + seg;
    const bool use_list = cl.stride > 0;
    const lds_u16 *list = lds_cast(cl.list) + (size_t)buf * cl.stride;
    const int trips_end = use_list ? cl.count[buf][wave] : (hi > lo ? hi - lo : 0); // list entries, or records of the segment
    const int last = n > 0 ? n - 1 : 0;
    auto fetch = [&](int k, double2 &p01, double2 &p23, bool &ok) { // k: position in the list / in the segment; one record ahead, as in clm_cost
        ok = k < trips_end;
        int ic;
        if (use_list) ic = ok ? (int)list[lo + k] : last;
        else { ic = lo + k < last ? lo + k : last; ok = (int)ok & (int)(mask ? mask[ic] != 0 : true); }
        const double2 *P = reinterpret_cast<const double2 *>(pts + (size_t)ic * PT_STRIDE);
        p01 = P[0]; p23 = P[1];
    };
    double2 c01, c23;
    bool cok;
    fetch(lane, c01, c23, cok);
    for (int k0 = 0; k0 < trips_end; k0 += 64) {
        double2 n01, n23;
        bool nok;
        fetch(k0 + 64 + lane, n01, n23, nok);
        if (cok) {
            double J[NP];
            const double r = clm_point<CK, true>(s, c01.x, c01.y, c23.x, c23.y, J);
            const double w = loss_weight(o.loss, o.loss_scale, r * r, o.mu);
            if (w != 0.0) {
                int idx = 0;
#pragma unroll
                for (int a = 0; a < NP; ++a) {
                    const double wa = w * J[a];
#pragma unroll
                    for (int b = 0; b <= a; ++b) acc[idx++] += wa * J[b];
                }
#pragma unroll
                for (int a = 0; a < NP; ++a) acc[NT + a] += w * r * J[a];
            }
        }
        c01 = n01; c23 = n23; cok = nok;
    }
    block_sum<NT + NP, T>(acc, scratch);
}

__device__ __forceinline__ void clm_quat_exp(const double *w, double *q) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    double re, im;
    if (th > 1e-6) { re = cos(0.5 * th); im = sin(0.5 * th) / th; }
    else { re = 1.0 - th2 / 8.0; im = 0.5 - th2 / 48.0; const double nq = 1.0 / sqrt(re * re + im * im * th2); re *= nq; im *= nq; }
    q[0] = re; q[1] = im * w[0]; q[2] = im * w[1]; q[3] = im * w[2];
}
__device__ __forceinline__ void clm_quat_mul(const double *a, const double *b, double *o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

template <int CK>
__device__ __forceinline__ void clm_step(const Model &m, const double *dp, const double *tb, Model &o) {
    o = m;
    double dq[4];
    if (CK == CLASSIC_FUND) {
        double qn[4];
        const double qV[4] = {m.t[0], m.t[1], m.t[2], m.scale};
        clm_quat_exp(dp, dq); clm_quat_mul(dq, m.q, o.q);
        clm_quat_exp(dp + 3, dq); clm_quat_mul(dq, qV, qn);
        o.t[0] = qn[0]; o.t[1] = qn[1]; o.t[2] = qn[2]; o.scale = qn[3];
        o.shift1 = m.shift1 + dp[6];
    } else {
        clm_quat_exp(dp, dq); clm_quat_mul(m.q, dq, o.q);
#pragma unroll
        for (int i = 0; i < 3; ++i) o.t[i] = m.t[i] + tb[i] * dp[3] + tb[3 + i] * dp[4];
        if (CK == CLASSIC_SHARED) { o.f1 = m.f1 + dp[5]; o.f2 = o.f1; }
    }
}

// 3 x 3 SVD by one-sided Jacobi (FactorizedFundamentalMatrix(F): JacobiSVD, proper rotations, sigma = s1 / s0)
__device__ void clm_factorize(const double *Fin, Model &m) {
    double B[9], W[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
    for (int i = 0; i < 9; ++i) B[i] = Fin[i];
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            double a = 0, b = 0, c = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) { a += B[3 * i + p] * B[3 * i + p]; b += B[3 * i + q] * B[3 * i + q]; c += B[3 * i + p] * B[3 * i + q]; }
            off = fmax(off, fabs(c) / sqrt(fmax(a * b, 2.2250738585072014e-308)));
            if (fabs(c) <= 1e-300) continue;
            const double zeta = (b - a) / (2.0 * c);
            const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double bp = B[3 * i + p], bq = B[3 * i + q];
                B[3 * i + p] = cs * bp - sn * bq; B[3 * i + q] = sn * bp + cs * bq;
                const double wp = W[3 * i + p], wq = W[3 * i + q];
                W[3 * i + p] = cs * wp - sn * wq; W[3 * i + q] = sn * wp + cs * wq;
            }
        }
        if (off < 1e-16) break;
    }
    double nrm[3];
    int ord[3] = {0, 1, 2};
#pragma unroll
    for (int j = 0; j < 3; ++j) nrm[j] = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
    if (nrm[ord[1]] > nrm[ord[0]]) { const int t = ord[0]; ord[0] = ord[1]; ord[1] = t; }
    if (nrm[ord[2]] > nrm[ord[0]]) { const int t = ord[0]; ord[0] = ord[2]; ord[2] = t; }
    if (nrm[ord[2]] > nrm[ord[1]]) { const int t = ord[1]; ord[1] = ord[2]; ord[2] = t; }
    double U[9], V[9], s[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double nj = 0, bj[3], wj[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { bj[i] = ord[j] == 0 ? B[3 * i] : (ord[j] == 1 ? B[3 * i + 1] : B[3 * i + 2]); wj[i] = ord[j] == 0 ? W[3 * i] : (ord[j] == 1 ? W[3 * i + 1] : W[3 * i + 2]); }
        nj = ord[j] == 0 ? nrm[0] : (ord[j] == 1 ? nrm[1] : nrm[2]);
        s[j] = nj;
#pragma unroll
        for (int i = 0; i < 3; ++i) { V[3 * i + j] = wj[i]; U[3 * i + j] = nj > 0 ? bj[i] / nj : 0.0; }
    }
    if (s[2] <= 1e-12 * s[0]) { // rank 2: the third left vector from the first two
        const double u0[3] = {U[0], U[3], U[6]}, u1[3] = {U[1], U[4], U[7]};
        double u2[3];
        cross3(u0, u1, u2);
        const double nn = 1.0 / sqrt(dot3(u2, u2));
#pragma unroll
        for (int i = 0; i < 3; ++i) U[3 * i + 2] = u2[i] * nn;
    }
    const double dU = U[0] * (U[4] * U[8] - U[5] * U[7]) - U[1] * (U[3] * U[8] - U[5] * U[6]) + U[2] * (U[3] * U[7] - U[4] * U[6]);
    const double dV = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
    if (dU < 0) { for (int i = 0; i < 9; ++i) U[i] = -U[i]; }
    if (dV < 0) { for (int i = 0; i < 9; ++i) V[i] = -V[i]; }
    double qV[4];
    model_identity(m);
    R_to_quat(U, m.q);
    R_to_quat(V, qV);
    m.t[0] = qV[0]; m.t[1] = qV[1]; m.t[2] = qV[2]; m.scale = qV[3];
    m.shift1 = s[1] / s[0];
}
__device__ __forceinline__ void clm_compose(const Model &ff, Model &out) {
    ClmState<CLASSIC_FUND> s;
    clm_setup<CLASSIC_FUND>(ff, s);
    model_identity(out);
    double *F = model_F(out);
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i] = s.F[i];
    out.shift2 = 0.0;
}

// lm_impl<> loop, executed uniformly by all threads of the problem's workgroup; the two sweeps are distributed
template <int CK, int T>
__device__ void clm_refine(Model &model, const double *__restrict__ pts, int n, const uint8_t *__restrict__ mask, const LmOpt &o_in,
                           double *scratch, ClmList &cl) {
    constexpr int NP = ClassicTraits<CK>::NP, NT = NP * (NP + 1) / 2;
    LmOpt o = o_in;
    o.mu = 0.5;
    Model m = model;
    if (CK == CLASSIC_FUND) clm_factorize(model_F(model), m);
    int cur = 0; // list buffer of the current model
    double cost = clm_cost<CK, T>(m, pts, n, mask, o, scratch, cl, cur);
    double lambda = o.lambda0;
    bool recompute = true;
    double acc[NT + NP], A[NP * NP], g[NP], sol[NP], tb[6];
    for (int it = 0; it < o.max_it; ++it) {
        if (recompute) {
            clm_accumulate<CK, T>(m, pts, n, mask, o, acc, tb, scratch, cl, cur);
            double gn = 0;
            int idx = 0;
#pragma unroll
            for (int a = 0; a < NP; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) A[a * NP + b] = acc[idx++];
#pragma unroll
            for (int a = 0; a < NP; ++a) { g[a] = acc[NT + a]; gn += g[a] * g[a]; }
            if (sqrt(gn) < o.grad_tol) break;
        }
        double Ad[NP * NP];
#pragma unroll
        for (int a = 0; a < NP; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) Ad[a * NP + b] = A[a * NP + b] + (a == b ? lambda : 0.0);
        chol_solve<NP>(Ad, g, sol);
        double sn = 0;
#pragma unroll
        for (int a = 0; a < NP; ++a) { sol[a] = -sol[a]; sn += sol[a] * sol[a]; }
        if (sqrt(sn) < o.step_tol) break;
        Model cand;
        clm_step<CK>(m, sol, tb, cand);
        const double cost_new = clm_cost<CK, T>(cand, pts, n, mask, o, scratch, cl, cur ^ 1);
        if (cost_new < cost) {
            m = cand;
            cur ^= 1;
            lambda = fmax(o.lambda_min, lambda / 10.0);
            cost = cost_new;
            recompute = true;
        } else {
            lambda = fmin(o.lambda_max, lambda * 10.0);
            recompute = false;
        }
        o.mu *= 1.5;
    }
    if (CK == CLASSIC_FUND) clm_compose(m, model);
    else model = m;
}

// workgroup-wide exact MSAC score of one model (score_model of the estimators); optional inlier mask
template <int CK, int T>
__device__ void cblock_score(const Model &m, const double *__restrict__ pts, int n, double thr, double *scratch, double &score_out,
                             int &cnt_out, uint8_t *__restrict__ mask_out) {
    double R[9], E[9];
    if (CK == CLASSIC_FUND) {
#pragma unroll
        for (int i = 0; i < 9; ++i) { E[i] = model_F(m)[i]; R[i] = 0.0; }
    } else {
        double Em[9];
        quat_to_R(m.q, R);
        essential_from_Rt(R, m.t, Em);
        if (CK == CLASSIC_RELPOSE) {
#pragma unroll
            for (int i = 0; i < 9; ++i) E[i] = Em[i];
        } else fundamental_from_E(Em, m.f1, m.f2, E);
    }
    double score = 0;
    int cnt = 0;
    for (int i = threadIdx.x; i < n; i += T) {
        double s1 = 0;
        int c1 = 0;
        score_point<CK == CLASSIC_RELPOSE>(pts + (size_t)i * PT_STRIDE, E, R, m.t, thr, s1, c1);
        score += s1; cnt += c1;
        if (mask_out) mask_out[i] = (uint8_t)c1;
    }
    double v[2] = {score, (double)cnt};
    block_sum<2, T>(v, scratch);
    cnt_out = (int)v[1];
    score_out = v[0] + thr * (double)(n - cnt_out);
    if (mask_out) __syncthreads(); // the mask is read by other threads of the workgroup next
}

// refine_model of the estimators: RelativePoseEstimator refines on the inliers at 5 thr^2 of the incoming model (get_inliers,
// kept only if more than the sample size), FundamentalEstimator on all correspondences; 25 iterations, TRUNCATED at eps
template <int CK, int T>
__device__ void clm_lo(Model &m, const PairState &ps, const double *__restrict__ pp, uint8_t *__restrict__ wg_mask, double *scratch, ClmList &cl) {
    LmOpt o;
    o.max_it = 25; o.loss = 1; o.loss_scale = ps.lo_loss_scale;
    o.grad_tol = 1e-10; o.step_tol = 1e-8; o.lambda0 = 1e-3; o.lambda_min = 1e-10; o.lambda_max = 1e10;
    if (CK == CLASSIC_FUND) { clm_refine<CK, T>(m, pp, ps.n, nullptr, o, scratch, cl); return; }
    double sc;
    int ni;
    cblock_score<CK, T>(m, pp, ps.n, 5.0 * ps.sq_thr, scratch, sc, ni, wg_mask);
    if (ni > ClassicTraits<CK>::K) clm_refine<CK, T>(m, pp, ps.n, wg_mask, o, scratch, cl);
    __syncthreads(); // wg_mask is rewritten by the next problem
}

// ------------------------------------------------------------------------------------------------ LO
template <int CK, int T>
__global__ __launch_bounds__(T, 2) void kc_lo(RunParams rp, const PairState *__restrict__ st, const double *__restrict__ pts,
                                              const Model *__restrict__ models, Trigger *__restrict__ triggers, int trig_cap,
                                              const int32_t *__restrict__ plan, int32_t *__restrict__ head /*zeroed*/,
                                              int32_t *__restrict__ xheads /*zeroed, or null: lo_take (mdrp_kernels.h)*/,
                                              uint8_t *__restrict__ lo_mask /*[gridDim.x][n_max]*/, int list_stride /*2 * list_stride u16 of dynamic LDS, or 0*/,
                                              FuseTail fz /*ready == null: off*/) {
    extern __shared__ uint16_t clm_dyn_list[];
    __shared__ double scratch[4 * MAX_ACC];
    __shared__ ClmList cl;
    __shared__ int s_item;
    if (threadIdx.x == 0) { cl.list = clm_dyn_list; cl.stride = list_stride; }
    const int32_t *prefix = plan, *begin = plan + rp.batch + 1, *end = begin + rp.batch;
    const int total = plan[3 * (size_t)rp.batch + 1];
    uint8_t *wg_mask = lo_mask + (size_t)blockIdx.x * rp.n_max;
    if (fz.ready && threadIdx.x == 0) { // fused tail (mdrp_kernels.h FuseTail): pairs without a trigger in this launch are ready as they are
        for (int p = blockIdx.x; p < rp.batch; p += gridDim.x)
            if (end[p] == begin[p]) fuse_publish(fz, rp, p, models, triggers, trig_cap);
        atomicAdd(fz.ctl + 1, 1);
    }
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_item = lo_take(head, xheads, total);
        __syncthreads();
        const int w = s_item;
        if (w >= total) break;
        const int pair = plan_find(prefix, rp.batch, w);
        const int pos = begin[pair] + (w - prefix[pair]);
        const PairState &ps = st[pair];
        Trigger &tr = triggers[(size_t)pair * trig_cap + pos];
        const size_t slot_base = (size_t)pair * rp.slot_stride;
        Model m = models[slot_base + (size_t)tr.iter * rp.mps + tr.k_ref];
        const double *pp = pts + (size_t)pair * rp.n_max * PT_STRIDE;
        clm_lo<CK, T>(m, ps, pp, wg_mask, scratch, cl);
        double sc;
        int cn;
        cblock_score<CK, T>(m, pp, ps.n, ps.sq_thr, scratch, sc, cn, nullptr);
        if (threadIdx.x == 0) {
            tr.refined = m; tr.ref_score = sc; tr.ref_cnt = cn;
            if (fz.ready) {
                __threadfence(); // this trigger's results before the count
                if (atomicAdd(fz.done_cnt + pair, 1) + 1 == end[pair] - begin[pair]) {
                    __threadfence(); // the other triggers' results (written on other CUs / XCDs) before the replay reads them
                    fuse_publish(fz, rp, pair, models, triggers, trig_cap);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ final
// ransac<> tail + get_inliers + the estimator's inlier-only refinement with the user's BundleOptions (estimate_relative_pose
// @0x21f800: if more than 5 inliers; estimate_fundamental @0x221a00: more than 7, then F <- T2' F T1 / |.|)
template <int CK, int T>
__device__ void cfinal_pair(const RunParams &rp, const PairState &ps, const double *__restrict__ pts, uint8_t *__restrict__ mask_all,
                            ResultDev *__restrict__ results, int pair, double *scratch, ClmList &cl) {
    ResultDev res;
    res.model = ps.best;
    res.refinements = ps.refinements; res.iterations = ps.iterations; res.num_inliers = ps.num_inliers;
    res.inlier_ratio = ps.inlier_ratio; res.model_score = ps.model_score;
    uint8_t *mask = mask_all + (size_t)pair * rp.n_max;
    if (ps.n < ClassicTraits<CK>::K) {
        for (int i = threadIdx.x; i < rp.n_max; i += T) mask[i] = 0;
        if (threadIdx.x == 0) results[pair] = res;
        return;
    }
    const double *pp = pts + (size_t)pair * rp.n_max * PT_STRIDE;
    Model m = ps.best;
    for (int i = ps.n + threadIdx.x; i < rp.n_max; i += T) mask[i] = 0;
    clm_lo<CK, T>(m, ps, pp, mask, scratch, cl); // the output mask doubles as the LO's subset mask
    res.refinements++;
    double sc;
    int cn;
    cblock_score<CK, T>(m, pp, ps.n, ps.sq_thr, scratch, sc, cn, nullptr);
    Model best = ps.best;
    if (sc < ps.model_score) { best = m; res.num_inliers = (uint64_t)cn; } // score / ratio NOT updated (reference)
    cblock_score<CK, T>(best, pp, ps.n, ps.sq_thr, scratch, sc, cn, mask);
    if (res.num_inliers > (uint64_t)ClassicTraits<CK>::K) {
        LmOpt f;
        f.max_it = rp.final_max_it; f.loss = rp.final_loss; f.loss_scale = ps.final_loss_scale;
        f.grad_tol = rp.grad_tol; f.step_tol = rp.step_tol; f.lambda0 = rp.lambda0; f.lambda_min = rp.lambda_min; f.lambda_max = rp.lambda_max;
        clm_refine<CK, T>(best, pp, ps.n, mask, f, scratch, cl);
    }
    if (CK == CLASSIC_FUND) { // F <- T2' F T1, T = [1/s 0 -cx/s; 0 1/s -cy/s; 0 0 1], unit Frobenius norm
        double *F = model_F(best);
        const double is = 1.0 / ps.norm;
        const double T1[9] = {is, 0, -ps.cen[0] * is, 0, is, -ps.cen[1] * is, 0, 0, 1}, T2[9] = {is, 0, -ps.cen[2] * is, 0, is, -ps.cen[3] * is, 0, 0, 1};
        double M[9], O[9], nrm = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) M[3 * i + j] = F[3 * i] * T1[j] + F[3 * i + 1] * T1[3 + j] + F[3 * i + 2] * T1[6 + j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) { O[3 * i + j] = T2[i] * M[j] + T2[3 + i] * M[3 + j] + T2[6 + i] * M[6 + j]; nrm += O[3 * i + j] * O[3 * i + j]; }
        nrm = 1.0 / sqrt(nrm);
#pragma unroll
        for (int i = 0; i < 9; ++i) F[i] = O[i] * nrm;
    }
    if (CK == CLASSIC_SHARED) { best.f1 *= ps.norm; best.f2 *= ps.norm; } // back to pixels
    res.model = best;
    if (threadIdx.x == 0) results[pair] = res;
}
template <int CK, int T>
__global__ __launch_bounds__(T, 2) void kc_final(RunParams rp, PairState *__restrict__ st, const double *__restrict__ pts,
                                                 uint8_t *__restrict__ mask_all, ResultDev *__restrict__ results,
                                                 const int32_t *__restrict__ ready /*or null: pair = blockIdx.x*/,
                                                 int32_t *__restrict__ fin_done /*fused: set per refined pair; unfused: pairs to skip, or null*/,
                                                 unsigned long long ticks, unsigned long long *__restrict__ timeouts /*fused: expired bounded waits*/,
                                                 int list_stride /*2 * list_stride u16 of dynamic LDS, or 0*/) {
    extern __shared__ uint16_t clm_dyn_list[];
    __shared__ double scratch[4 * MAX_ACC];
    __shared__ ClmList cl;
    __shared__ int s_pair;
    if (threadIdx.x == 0) { cl.list = clm_dyn_list; cl.stride = list_stride; }
    __syncthreads();
    __shared__ __attribute__((aligned(16))) unsigned int s_ps[(sizeof(PairState) + 3) / 4];
    if (!ready) {
        if (fin_done && fin_done[blockIdx.x]) return; // (uniform) the pass behind a fused tail: only what that left undone
        cfinal_pair<CK, T>(rp, st[blockIdx.x], pts, mask_all, results, blockIdx.x, scratch, cl);
        return;
    }
    if (threadIdx.x == 0) { // fused tail: the blockIdx-th pair to become ready, bounded wait (k_final / k_gate, mdrp_kernels.h)
        int p;
        const unsigned long long t0 = wall_clock64();
        // polled RELAXED, one acquire fence when the pair is there (see k_final, mdrp_kernels.h)
        while ((p = __hip_atomic_load(ready + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0 && wall_clock64() - t0 < ticks)
            __builtin_amdgcn_s_sleep(16);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (p < 0 && timeouts) atomicAdd(timeouts, 1ull); // gave up: the pass behind the LO launch refines this pair (mdrp_stats.fuse_wait_timeouts)
        __threadfence();
        s_pair = p;
    }
    __syncthreads();
    if (s_pair < 0) return;
    {
        const volatile unsigned int *src = reinterpret_cast<const volatile unsigned int *>(st + s_pair);
        for (int i = threadIdx.x; i < (int)(sizeof(PairState) / 4); i += T) s_ps[i] = src[i];
    }
    __syncthreads();
    cfinal_pair<CK, T>(rp, *reinterpret_cast<const PairState *>(s_ps), pts, mask_all, results, s_pair, scratch, cl);
    if (threadIdx.x == 0) fin_done[s_pair] = 1;
}

// ------------------------------------------------------------------------------------------------ unit-parity kernels
template <int CK>
__global__ __launch_bounds__(64) void kc_solver_unit(int count, const double *__restrict__ x1h, const double *__restrict__ x2h, Model *__restrict__ out,
                               int32_t *__restrict__ n_out) {
    constexpr int K = ClassicTraits<CK>::K, MAXM = ClassicTraits<CK>::MAXM;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double a[K][3], b[K][3];
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < 3; ++c) { a[k][c] = x1h[(size_t)3 * K * i + 3 * k + c]; b[k][c] = x2h[(size_t)3 * K * i + 3 * k + c]; }
    Model m[MAXM];
    extern __shared__ double solve5_lds[];
    const int n = (CK == CLASSIC_RELPOSE) ? solver_relpose_5pt(a, b, m, lds_solve5_store())
                  : (CK == CLASSIC_SHARED ? solver_relpose_6pt(a, b, m) : solver_fundamental_7pt(a, b, m, solve5_lds + (threadIdx.x & 63), 64));
    n_out[i] = n;
    for (int k = 0; k < n; ++k) out[(size_t)MAXM * i + k] = m[k];
}

template <int CK, int T>
__global__ __launch_bounds__(T) void kc_refine_unit(int count, Model *__restrict__ models, const double *__restrict__ pts, int n, LmOpt o,
                                                    double *__restrict__ final_cost) {
    __shared__ double scratch[4 * MAX_ACC];
    __shared__ ClmList cl;
    const int i = blockIdx.x;
    if (i >= count) return;
    if (threadIdx.x == 0) { cl.list = nullptr; cl.stride = 0; } // the unit entry point sweeps every record (no work lists)
    __syncthreads();
    Model m = models[i];
    clm_refine<CK, T>(m, pts, n, nullptr, o, scratch, cl);
    Model f = m;
    if (CK == CLASSIC_FUND) clm_factorize(model_F(m), f);
    const double c = clm_cost<CK, T>(f, pts, n, nullptr, o, scratch, cl, 0);
    if (threadIdx.x == 0) { models[i] = m; if (final_cost) final_cost[i] = c; }
}

} // namespace mdrp
