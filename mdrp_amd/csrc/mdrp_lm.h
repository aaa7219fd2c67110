// mdrp_lm.h — phase-batched Levenberg-Marquardt engine for the LO / final refinements of the monodepth estimators
// (refine_monodepth_relpose @0x261030, ..._shared_focal_relpose @0x2592e0, ..._varying_focal_relpose @0x260fa0 + lm_impl<>).
//
// The reference runs one LM loop per refinement, start to end.  Rounds 1-2 mapped that loop onto one wavefront (or one
// workgroup) per problem: every problem dragged its own serial part (Cholesky, step, accept / reject) and its own sweeps
// through one wavefront, a launch ended with its longest problem (mean residency 1.15 wavefronts per SIMD of 2), and every
// LM iteration of every problem re-streamed its pair's records (55x re-fetch).  Here the LOOP is turned inside
// out: all problems of a phase advance together, one LM iteration per ROUND, and a round is three kernels
//
//   k_lme_cost   one wavefront per (pair, segment of correspondences): the records are loaded ONCE into registers and
//                evaluated against the candidate model of every problem of that pair that is still iterating (4-12 LO
//                problems share a pair; their expanded states are staged through LDS in one go): cost partial per
//                (problem, segment) and the segment's work list (correspondences with a non-zero IRLS weight, ballot-compacted);
//   k_lme_accum  one workgroup per problem: accept / reject of the candidate (cost partials summed in segment order), then
//                J'J | J'r over the work list of the accepted model — the dense list is rebuilt in LDS from the segment
//                lists, four wavefronts take every fourth trip of 64 entries, and the 35-54 accumulators are reduced across
//                lanes with the gfx950 half-/quarter-wave swaps (v_permlane32_swap / v_permlane16_swap: one add reduces two
//                accumulators) instead of 6 shuffles each;
//   k_lme_solve  one LANE per problem: gradient test, damped Cholesky solve, step test, next candidate and its expanded
//                state — the serial part of lm_impl<>, paid once per problem instead of once per wavefront.
//
// Everything a problem carries between rounds (model, candidate, lambda, J'J, lists) lives in HBM (about 1 KB + 2 n bytes);
// problems that converge simply stop taking part, so a round costs what its live problems cost and the chip is refilled
// every few microseconds instead of waiting for stragglers.  Summation orders are fixed by record index and list position
// alone, so results do not depend on the batch, the grid or the schedule.  The LM arithmetic itself (residuals, Jacobians,
// losses, Cholesky recurrences, step) is mdrp_math.h's, unchanged.
#pragma once
#include "mdrp_kernels.h"

namespace mdrp {

#ifndef MDRP_LME_RPT
#define MDRP_LME_RPT 4
#endif
#ifndef MDRP_LME_MINWAVES
#define MDRP_LME_MINWAVES 2 // workgroups of k_lme_accum per CU (2 = 256 VGPRs per lane)
#endif
constexpr int LME_RPT = MDRP_LME_RPT;  // records per lane in the cost sweep
constexpr int LME_SEG = 64 * LME_RPT;  // correspondences per cost-sweep segment (one wavefront)
constexpr int LME_T = 256;             // threads per problem in k_lme_accum
constexpr int LME_NW = LME_T / 64;
constexpr int LME_RING = 64;           // live counters of the rounds live in a ring (a round clears the slot 32 rounds ahead)
constexpr int LME_STAGE = 16;          // problem states staged in LDS per pass of the cost sweep
constexpr int LME_HEAD = 50;           // leading doubles of LmProb the sweeps need (state, loss parameters, flags)

struct alignas(64) LmProb {
    // --- read by the cost sweep: the first LME_HEAD doubles
    LmState cs;             // the model under evaluation, expanded (R, t, s, u, v, f1, f2, E, F, 1 / f1, 1 / f2): 37 doubles
    double sqrt_sr, ws, loss_scale, mu;
    int32_t status;         // 1 = iterating, 0 = finished (m is the result) or unused
    int32_t has_cand;       // 0 nothing to evaluate, 1 candidate step, 2 the initial model (its cost starts the loop)
    int32_t loss, cur;      // cur: list buffer that belongs to the current model m
    int32_t pair, n, it, max_it, recompute;
    int32_t need_acc;       // segment engine: set by k_lme_decide when the normal equations of m have to be (re)built this round
    unsigned long long ev_cost, ev_acc; // correspondences evaluated by the cost sweeps / the normal-equation sweeps of this problem (mdrp_stats)
    int32_t n_eff, pad3_;   // records a dense normal-equation sweep really visits (n, or the inlier count under a record mask)
    double pad2_;           // (the header is an even number of doubles: 16-byte rows in the LDS stage)
    // --- LM state
    double cost, lambda, grad_tol, step_tol, lambda_min, lambda_max;
    Model m, cand;
    double acc[MAX_ACC];    // J'J (lower triangle, row major) | J'r of the current model (kept for rejected steps)
};
static_assert(offsetof(LmProb, cost) == LME_HEAD * sizeof(double), "sweep header of LmProb");
// positions (in doubles) of the header fields inside the staged copy
constexpr int LME_O_SQRT_SR = offsetof(LmProb, sqrt_sr) / 8, LME_O_WS = offsetof(LmProb, ws) / 8, LME_O_LSC = offsetof(LmProb, loss_scale) / 8,
              LME_O_MU = offsetof(LmProb, mu) / 8, LME_O_STATUS = offsetof(LmProb, status) / 8 /*.x status .y has_cand*/,
              LME_O_LOSS = offsetof(LmProb, loss) / 8 /*.x loss .y cur*/, LME_O_NEED = offsetof(LmProb, recompute) / 8 /*.x recompute .y need_acc*/;
static_assert(offsetof(LmProb, status) % 8 == 0 && offsetof(LmProb, loss) % 8 == 0 && offsetof(LmProb, recompute) % 8 == 0 && sizeof(LmState) == 37 * 8, "header layout");
// the staged header of one problem -> its expanded state in scalar registers
__device__ __forceinline__ void lme_state_from_header(const double *S, LmState &stt) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { stt.R[i] = S[i]; stt.E[i] = S[17 + i]; stt.F[i] = S[26 + i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) stt.t[i] = S[9 + i];
    stt.s = S[12]; stt.u = S[13]; stt.v = S[14]; stt.f1 = S[15]; stt.f2 = S[16]; stt.if1 = S[35]; stt.if2 = S[36];
    lm_state_uniform(stt);
}

struct LmePhase {
    LmProb *probs;          // [cap]
    double *part;           // [cap][nseg] per (problem, segment): cost partial; squared-residual sum in the closing score sweep
    int32_t *ipart;         // [cap][nseg] inlier counts of the closing score sweep
    uint8_t *list;          // [cap][2][nseg][LME_SEG] work lists: segment-relative record indices, ascending
    uint16_t *list_cnt;     // [cap][2][nseg]
    const int32_t *pfx;     // [batch + 1] dense problem indices of pair p: [pfx[p], pfx[p + 1])
    const int32_t *total;   // number of problems of the phase (device side: only the scan knows how many triggers there are)
    int32_t *live;          // [LME_RING] problems still iterating after the round
    int32_t *pair_live;     // [2][batch] live problems per pair after round r in half r & 1 (k_lme_solve); lets the cost sweep of a
                            // pair without live problems leave after its first round trip
    int32_t *pair_acc;      // [2][batch] segment engine: problems of pair p whose normal equations are rebuilt in round r, in half r & 1 (k_lme_decide)
    double *accpart;        // [cap][nseg][MAX_ACC] segment engine: J'J | J'r partial of (problem, segment)
    int first, cap;         // this pass handles the dense problems [first, first + cap)
    int batch, n_max, nseg;
    const uint8_t *mask;    // [batch][n_max] record mask (the inlier-only final refinement) or null
};

__device__ __forceinline__ int lme_count(const LmePhase &ph) {
    const int t = *ph.total - ph.first;
    return t < 0 ? 0 : (t < ph.cap ? t : ph.cap);
}

// (the cross-lane sums — dpp_f64, swap32 / swap16, wave_sum_swap, wave_reduce_scatter — live in mdrp_kernels.h since round 4: block_sum uses them too)

// ------------------------------------------------------------------------------------------------ cost sweep
// grid (nseg, batch), 64 threads.  Record i of the segment sits in lane i & 63, slot i >> 6.
// LOSS: the phase's loss type when it is known at compile time (1 = TRUNCATED: every LO refinement), -1 = read per problem
// DENSE (segment engine): no work lists — the normal-equation sweep of that engine visits every record with its weight as a select.
// Without lists a wavefront may hold more records per lane (RPT is a template parameter; k_lme_decide takes the segment size):
// measured on varying focal, 1024 x 5000: RPT = 8 halves the per-problem overhead (header -> scalar registers, wave sum, partial store:
// ~150 of ~280 instructions per record at RPT = 4) but needs 195 VGPRs, two wavefronts per SIMD instead of four: 53.4 against 51.3 ms
// per step.  4 it is.
constexpr int LME_RPT_DENSE = 4;
template <int KIND, int LOSS, bool DENSE = false, int RPT = (DENSE ? LME_RPT_DENSE : LME_RPT)>
__global__ __launch_bounds__(64) void k_lme_cost(LmePhase ph, const PairState *__restrict__ st, const double *__restrict__ pts,
                                                 const double *__restrict__ dep, int live_half /*-1: the initial sweep, every problem is live*/) {
    __shared__ double s_state[LME_STAGE][LME_HEAD];
    __shared__ int s_idx[LME_STAGE];
    const int pair = blockIdx.y, seg = blockIdx.x, lane = threadIdx.x;
    const int n = st[pair].n;
    const int pl = live_half >= 0 ? ph.pair_live[(size_t)live_half * ph.batch + pair] : 1;
    if (seg * (64 * RPT) >= n || pl == 0) return;
    const int cnt = lme_count(ph);
    const int j0 = max(ph.pfx[pair] - ph.first, 0), j1 = min(ph.pfx[pair + 1] - ph.first, cnt);
    if (j0 >= j1) return;
    const unsigned long long lt = (1ull << lane) - 1ull;
    bool loaded = false;
    double ra[RPT], rb[RPT], rc[RPT], rd[RPT], e1[RPT], e2[RPT];
    bool ok[RPT];
    for (int base = j0; base < j1; base += LME_STAGE) {
        // One round trip: the headers (expanded state, loss parameters, flags) of the next LME_STAGE problems of this pair go to
        // LDS whether they are live or not — their addresses depend on the pair alone — together with the records; which of them
        // have a model waiting for its cost is read from the staged flags afterwards.
        const int kk = min(LME_STAGE, j1 - base);
        __syncthreads(); // previous pass done with the stage
        for (int e = lane; e < kk * LME_HEAD; e += 64) {
            const int r = e / LME_HEAD, f = e - r * LME_HEAD;
            s_state[r][f] = reinterpret_cast<const double *>(ph.probs + base + r)[f];
        }
        if (!loaded) { // the records, once, while the headers are in flight
            loaded = true;
            const double *pp = pts + (size_t)pair * ph.n_max * PT_STRIDE;
            const double *dd = dep + (size_t)pair * ph.n_max * 2;
            const uint8_t *mk = ph.mask ? ph.mask + (size_t)pair * ph.n_max : nullptr;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int i = seg * (64 * RPT) + r * 64 + lane;
                ok[r] = i < n && (!mk || mk[i]);
                ra[r] = rb[r] = rc[r] = rd[r] = 0; e1[r] = e2[r] = 1;
                if (ok[r]) {
                    const double2 *P = reinterpret_cast<const double2 *>(pp + (size_t)i * PT_STRIDE);
                    const double2 p01 = P[0], p23 = P[1];
                    const double2 d12 = *reinterpret_cast<const double2 *>(dd + 2 * (size_t)i);
                    ra[r] = p01.x; rb[r] = p01.y; rc[r] = p23.x; rd[r] = p23.y; e1[r] = d12.x; e2[r] = d12.y;
                }
            }
        }
        __syncthreads();
        bool live = false;
        if (lane < kk) { const int2 fl = *reinterpret_cast<const int2 *>(&s_state[lane][LME_O_STATUS]); live = fl.x != 0 && fl.y != 0; } // status, has_cand
        const unsigned long long lb = __ballot(live);
        const int k = __popcll(lb);
        if (k == 0) continue;
        if (live) s_idx[__popcll(lb & lt)] = lane;
        __syncthreads();
        for (int q = 0; q < k; ++q) {
            const int sl = s_idx[q], j = base + sl;
            const double *S = s_state[sl];
            LmState stt;
            lme_state_from_header(S, stt);
            const double sqrt_sr = uniform_f64(S[LME_O_SQRT_SR]), ws = uniform_f64(S[LME_O_WS]), lsc = uniform_f64(S[LME_O_LSC]), mu = uniform_f64(S[LME_O_MU]);
            const int2 lc = *reinterpret_cast<const int2 *>(S + LME_O_LOSS); // loss, cur
            const int loss = LOSS >= 0 ? LOSS : __builtin_amdgcn_readfirstlane(lc.x), buf = __builtin_amdgcn_readfirstlane(lc.y) ^ 1;
            double cost = 0;
            bool contrib[RPT];
            // straight-line over the lane's records (padding lanes hold a harmless record and are masked by selects): the
            // RPT residual chains are independent, so the scheduler interleaves them — a branch per record serialised them
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                double res[5], zf, zb;
                point_residuals<false, KIND != 0>(stt, sqrt_sr, ra[r], rb[r], rc[r], rd[r], e1[r], e2[r], res, zf, zb, nullptr);
                const double rs = res[0] * res[0], rf = res[1] * res[1] + res[2] * res[2], rbk = res[3] * res[3] + res[4] * res[4];
                const bool fwd = zf > 0, bwd = zb > 0;
                double c = ws * loss_value(loss, lsc, rs);
                c += fwd ? loss_value(loss, lsc, rf) : 0.0;
                c += bwd ? loss_value(loss, lsc, rbk) : 0.0;
                cost += ok[r] ? c : 0.0;
                if (!DENSE)
                    contrib[r] = ok[r] && ((sampson_row_weight<KIND>(loss, lsc, mu, ws, ws * ws, rs) != 0.0) || (fwd && loss_weight(loss, lsc, rf, mu) != 0.0) ||
                                           (bwd && loss_weight(loss, lsc, rbk, mu) != 0.0));
            }
            cost = wave_sum_swap(cost);
            if (DENSE) {
                if (lane == 0) ph.part[(size_t)j * ph.nseg + seg] = cost;
                continue;
            }
            uint8_t *L = ph.list + (((size_t)j * 2 + buf) * ph.nseg + seg) * (64 * RPT);
            int fill = 0;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const unsigned long long ball = __ballot(contrib[r]);
                if (contrib[r]) L[fill + __popcll(ball & lt)] = (uint8_t)(r * 64 + lane);
                fill += __popcll(ball);
            }
            if (lane == 0) {
                ph.part[(size_t)j * ph.nseg + seg] = cost;
                ph.list_cnt[((size_t)j * 2 + buf) * ph.nseg + seg] = (uint16_t)fill;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ accept / reject + J'J
// J'J | J'r of model m over the work list `buf` of problem j; the totals land in scratch[0 .. NA).
// LDS: int32 pre[nseg + 1] | u16 dense[] (n <= dense_cap), scratch[LME_NW][MAX_ACC].
template <int KIND, bool SHIFT>
__device__ void lme_accumulate(const Model &m, const double *__restrict__ pts, const double *__restrict__ dep, int n,
                               const LmePhase &ph, int j, int buf, double sqrt_sr, double ws, const LmOpt &o, double (*scratch)[MAX_ACC],
                               int32_t *pre, uint16_t *dense, bool use_dense) {
    constexpr int NP = LmTraits<KIND, SHIFT>::NP;
    constexpr int NA = NP * (NP + 1) / 2 + NP;
    LmState stt;
    lm_state_from_model(m, KIND != 0, stt);
    lm_state_uniform(stt);
    double acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg = (n + LME_SEG - 1) / LME_SEG;
    const uint16_t *cnt = ph.list_cnt + ((size_t)j * 2 + buf) * ph.nseg;
    const uint8_t *lst = ph.list + ((size_t)j * 2 + buf) * ph.nseg * LME_SEG;
    if (use_dense) {
        // exclusive prefix of the segment counts (wave 0: lane l owns K consecutive segments), then the dense list
        if (wave == 0) {
            const int K = (nseg + 63) / 64;
            int own = 0;
            for (int k = 0; k < K; ++k) { const int s = lane * K + k; if (s < nseg) own += cnt[s]; }
            int inc = own;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
            int run = inc - own;
            for (int k = 0; k < K; ++k) { const int s = lane * K + k; if (s < nseg) { pre[s] = run; run += cnt[s]; } }
            if (lane == 63) pre[nseg] = inc;
        }
        __syncthreads();
        for (int s = wave; s < nseg; s += LME_NW) {
            const int c = cnt[s], p0 = pre[s];
            for (int e = lane; e < c; e += 64) dense[p0 + e] = (uint16_t)(s * LME_SEG + lst[(size_t)s * LME_SEG + e]);
        }
        __syncthreads();
        const int total = pre[nseg];
        // trips of 64 list entries; wavefront w takes trips w, w + 4, ...; software-pipelined by one trip
        double2 n01 = make_double2(0, 0), n23 = n01, ndd = n01;
        auto fetch = [&](int k) {
            if (k < total) {
                const size_t i = (size_t)dense[k];
                const double2 *P = reinterpret_cast<const double2 *>(pts + i * PT_STRIDE);
                n01 = P[0]; n23 = P[1];
                ndd = *reinterpret_cast<const double2 *>(dep + 2 * i);
            }
        };
        fetch(wave * 64 + lane);
        for (int k = wave * 64 + lane; k < total; k += 64 * LME_NW) {
            const double2 c01 = n01, c23 = n23, cdd = ndd;
            fetch(k + 64 * LME_NW);
            lm_accumulate_point<KIND, SHIFT>(stt, c01, c23, cdd, sqrt_sr, ws, ws * ws, o, acc);
        }
    } else {
        // large pairs: walk the segment lists directly (wavefront w takes segments w, w + 4, ...)
        for (int s = wave; s < nseg; s += LME_NW) {
            const int c = cnt[s];
            for (int e = lane; e < c; e += 64) {
                const size_t i = (size_t)s * LME_SEG + lst[(size_t)s * LME_SEG + e];
                const double2 *P = reinterpret_cast<const double2 *>(pts + i * PT_STRIDE);
                lm_accumulate_point<KIND, SHIFT>(stt, P[0], P[1], *reinterpret_cast<const double2 *>(dep + 2 * i), sqrt_sr, ws, ws * ws, o, acc);
            }
        }
    }
    wave_reduce_scatter<NA>(acc, scratch[wave]);
    __syncthreads();
    if (threadIdx.x < NA) {
        double s = scratch[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < LME_NW; ++w) s += scratch[w][threadIdx.x];
        scratch[0][threadIdx.x] = s;
    }
    __syncthreads();
}

// lm_impl<>'s loop body from "the cost of the candidate is known" to "the normal equations of the current model are known"
// (upstream PoseLib convention, as lm_refine of round 2).  Workgroups stride over the problems of the pass.
template <int KIND, bool SHIFT, int LOSS>
__global__ __launch_bounds__(LME_T, MDRP_LME_MINWAVES) void k_lme_accum(LmePhase ph, const double *__restrict__ pts, const double *__restrict__ dep,
                                                                       int round, int dense_cap) {
    constexpr int NP = LmTraits<KIND, SHIFT>::NP;
    constexpr int NA = NP * (NP + 1) / 2 + NP;
    extern __shared__ int32_t lme_dyn[];
    __shared__ double scratch[LME_NW][MAX_ACC];
    int32_t *pre = lme_dyn;
    uint16_t *dense = reinterpret_cast<uint16_t *>(lme_dyn + ph.nseg + 1);
    const int count = lme_count(ph);
    if (blockIdx.x == 0 && threadIdx.x == 0) ph.live[(round + LME_RING / 2) & (LME_RING - 1)] = 0; // free: every earlier round has ended
    for (int j = blockIdx.x; j < count; j += gridDim.x) {
        LmProb *P = ph.probs + j;
        if (P->status == 0) continue;
        // ---- local copy of the loop state (uniform)
        Model m = P->m;
        double cost = P->cost, lambda = P->lambda, mu = P->mu;
        int it = P->it, cur = P->cur, recompute = 0;
        const int has_cand = P->has_cand, n = P->n, pair = P->pair, max_it = P->max_it;
        const double lambda_min = P->lambda_min, lambda_max = P->lambda_max;
        __syncthreads(); // every wavefront holds its copy before thread 0 writes anything back; LDS of the previous problem is free
        if (has_cand == 0) continue; // (a problem that is live always has a model waiting)
        // ---- accept / reject the candidate whose cost the sweep just produced
        {
            const int nseg = (n + LME_SEG - 1) / LME_SEG;
            double cost_new = 0;
            for (int s = 0; s < nseg; ++s) cost_new += ph.part[(size_t)j * ph.nseg + s];
            if (threadIdx.x == 0) P->ev_cost += (unsigned long long)n;
            if (has_cand == 2) { cost = cost_new; cur ^= 1; recompute = 1; }
            else {
                if (cost_new < cost) { m = P->cand; cur ^= 1; lambda = fmax(lambda_min, lambda / 10.0); cost = cost_new; recompute = 1; }
                else { lambda = fmin(lambda_max, lambda * 10.0); }
                mu *= 1.5; // TRUNCATED_LE_ZACH: the reference's per-iteration callback
                ++it;
            }
        }
        const bool done = it >= max_it;
        if (!done && recompute) {
            LmOpt o;
            o.max_it = max_it; o.loss = LOSS >= 0 ? LOSS : P->loss; o.loss_scale = P->loss_scale; o.mu = mu;
            const double *pp = pts + (size_t)pair * ph.n_max * PT_STRIDE;
            const double *dd = dep + (size_t)pair * ph.n_max * 2;
            lme_accumulate<KIND, SHIFT>(m, pp, dd, n, ph, j, cur, P->sqrt_sr, P->ws, o, scratch, pre, dense, n <= dense_cap);
            if (threadIdx.x < NA) P->acc[threadIdx.x] = scratch[0][threadIdx.x];
            if (threadIdx.x == LME_T - 1) {
                const uint16_t *lc = ph.list_cnt + ((size_t)j * 2 + cur) * ph.nseg;
                unsigned long long tot = 0;
                for (int s_ = 0; s_ < (n + LME_SEG - 1) / LME_SEG; ++s_) tot += lc[s_];
                P->ev_acc += tot;
            }
        }
        if (threadIdx.x == 0) {
            P->m = m; P->cost = cost; P->lambda = lambda; P->mu = mu; P->it = it; P->cur = cur; P->recompute = recompute;
            P->has_cand = 0;
            if (done) P->status = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ segment engine (dense sweeps)
// Where the loss cannot truncate enough to pay for a work list — the varying-focal LO leaves loss_scale at 1.0 on scale-normalised
// points, so every sweep of every problem covers all N records — the normal equations are built the way the cost sweep
// works: one wavefront per (pair, segment of 256 records) keeps the records in registers and visits the pair's problems that need
// new normal equations one after the other; the 35-54 sums of (problem, segment) are reduce-scattered across the wavefront and stored,
// and a second kernel adds the segments in order.  No lists, no list -> record indirection, every wavefront of a round has the same
// amount of work, and a pair's records are read once per round for all of its problems.  A round is
//     k_lme_cost<DENSE> | k_lme_decide | k_lme_accum_seg | k_lme_reduce | k_lme_solve
// k_lme_decide: one lane per problem — lm_impl<>'s accept / reject once the candidate's cost is known (the first half of k_lme_accum)
MDRP_GLOBAL void k_lme_decide(LmePhase ph, int round, int cost_seg /*records per cost-sweep segment*/) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) ph.live[(round + LME_RING / 2) & (LME_RING - 1)] = 0; // free: every earlier round has ended
    if (j < ph.batch) ph.pair_acc[(size_t)((round + 1) & 1) * ph.batch + j] = 0; // the half the NEXT round counts into (its reader, the previous round's accum sweep, is done)
    if (j >= lme_count(ph)) return;
    LmProb *P = ph.probs + j;
    if (P->status == 0 || P->has_cand == 0) { P->need_acc = 0; return; }
    Model m = P->m;
    double cost = P->cost, lambda = P->lambda, mu = P->mu;
    int it = P->it, cur = P->cur, recompute = 0;
    const int has_cand = P->has_cand, n = P->n, max_it = P->max_it;
    const int nseg = (n + cost_seg - 1) / cost_seg;
    double cost_new = 0;
    for (int s = 0; s < nseg; ++s) cost_new += ph.part[(size_t)j * ph.nseg + s];
    P->ev_cost += (unsigned long long)n;
    if (has_cand == 2) { cost = cost_new; cur ^= 1; recompute = 1; }
    else {
        if (cost_new < cost) { m = P->cand; cur ^= 1; lambda = fmax(P->lambda_min, lambda / 10.0); cost = cost_new; recompute = 1; }
        else { lambda = fmin(P->lambda_max, lambda * 10.0); }
        mu *= 1.5; // TRUNCATED_LE_ZACH: the reference's per-iteration callback
        ++it;
    }
    const bool done = it >= max_it;
    const int need = (!done && recompute) ? 1 : 0;
    // (P->cs is the expanded state of the model just evaluated: of m whenever `recompute` is set)
    P->m = m; P->cost = cost; P->lambda = lambda; P->mu = mu; P->it = it; P->cur = cur; P->recompute = recompute;
    P->has_cand = 0; P->need_acc = need;
    if (done) P->status = 0;
    if (need) { P->ev_acc += (unsigned long long)P->n_eff; atomicAdd(ph.pair_acc + (size_t)(round & 1) * ph.batch + P->pair, 1); }
}

// J'J | J'r of (problem, segment) for every problem of the pair flagged by k_lme_decide.  grid (nseg, batch), 64 threads.
template <int KIND, bool SHIFT, int LOSS>
__global__ __launch_bounds__(64, MDRP_LME_MINWAVES) void k_lme_accum_seg(LmePhase ph, const PairState *__restrict__ st, const double *__restrict__ pts,
                                                                         const double *__restrict__ dep, int round) {
    constexpr int NP = LmTraits<KIND, SHIFT>::NP;
    constexpr int NA = NP * (NP + 1) / 2 + NP;
    __shared__ double s_state[LME_STAGE][LME_HEAD];
    __shared__ int s_idx[LME_STAGE];
    const int pair = blockIdx.y, seg = blockIdx.x, lane = threadIdx.x;
    if (ph.pair_acc[(size_t)(round & 1) * ph.batch + pair] == 0) return;
    const int n = st[pair].n;
    if (seg * LME_SEG >= n) return;
    const int cnt = lme_count(ph);
    const int j0 = max(ph.pfx[pair] - ph.first, 0), j1 = min(ph.pfx[pair + 1] - ph.first, cnt);
    if (j0 >= j1) return;
    const unsigned long long lt = (1ull << lane) - 1ull;
    bool loaded = false;
    double2 r01[LME_RPT], r23[LME_RPT], rdd[LME_RPT];
    bool ok[LME_RPT];
    for (int base = j0; base < j1; base += LME_STAGE) {
        const int kk = min(LME_STAGE, j1 - base);
        __syncthreads();
        for (int e = lane; e < kk * LME_HEAD; e += 64) {
            const int r = e / LME_HEAD, f = e - r * LME_HEAD;
            s_state[r][f] = reinterpret_cast<const double *>(ph.probs + base + r)[f];
        }
        if (!loaded) {
            loaded = true;
            const double *pp = pts + (size_t)pair * ph.n_max * PT_STRIDE;
            const double *dd = dep + (size_t)pair * ph.n_max * 2;
            const uint8_t *mk = ph.mask ? ph.mask + (size_t)pair * ph.n_max : nullptr;
#pragma unroll
            for (int r = 0; r < LME_RPT; ++r) {
                const int i = seg * LME_SEG + r * 64 + lane;
                ok[r] = i < n && (!mk || mk[i]);
                r01[r] = r23[r] = make_double2(0, 0); rdd[r] = make_double2(1, 1);
                if (ok[r]) {
                    const double2 *P = reinterpret_cast<const double2 *>(pp + (size_t)i * PT_STRIDE);
                    r01[r] = P[0]; r23[r] = P[1];
                    rdd[r] = *reinterpret_cast<const double2 *>(dd + 2 * (size_t)i);
                }
            }
        }
        __syncthreads();
        bool want = false;
        if (lane < kk) {
            const int status = reinterpret_cast<const int2 *>(&s_state[lane][LME_O_STATUS])->x, need = reinterpret_cast<const int2 *>(&s_state[lane][LME_O_NEED])->y;
            want = status != 0 && need != 0;
        }
        const unsigned long long wb = __ballot(want);
        const int k = __popcll(wb);
        if (k == 0) continue;
        if (want) s_idx[__popcll(wb & lt)] = lane;
        __syncthreads();
        for (int q = 0; q < k; ++q) {
            const int sl = s_idx[q], j = base + sl;
            const double *S = s_state[sl];
            LmState stt;
            lme_state_from_header(S, stt);
            const double sqrt_sr = uniform_f64(S[LME_O_SQRT_SR]), ws = uniform_f64(S[LME_O_WS]);
            LmOpt o;
            o.max_it = 0; o.loss = LOSS >= 0 ? LOSS : __builtin_amdgcn_readfirstlane(reinterpret_cast<const int2 *>(S + LME_O_LOSS)->x);
            o.loss_scale = uniform_f64(S[LME_O_LSC]); o.mu = uniform_f64(S[LME_O_MU]);
            o.grad_tol = o.step_tol = o.lambda0 = o.lambda_min = o.lambda_max = 0;
            double acc[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = 0;
#pragma unroll
            for (int r = 0; r < LME_RPT; ++r)
                if (ok[r]) lm_accumulate_point<KIND, SHIFT, LOSS>(stt, r01[r], r23[r], rdd[r], sqrt_sr, ws, ws * ws, o, acc);
            wave_reduce_scatter<NA>(acc, ph.accpart + ((size_t)j * ph.nseg + seg) * MAX_ACC);
        }
    }
}

// the segments of a problem in order: one wavefront per problem, lane a owns accumulator a
template <int KIND, bool SHIFT>
__global__ __launch_bounds__(256) void k_lme_reduce(LmePhase ph) {
    constexpr int NP = LmTraits<KIND, SHIFT>::NP;
    constexpr int NA = NP * (NP + 1) / 2 + NP;
    const int lane = threadIdx.x & 63;
    const int count = lme_count(ph);
    for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < count; j += gridDim.x * 4) {
        LmProb *P = ph.probs + j;
        if (P->status == 0 || P->need_acc == 0) continue;
        const int nseg = (P->n + LME_SEG - 1) / LME_SEG;
        if (lane < NA) {
            const double *src = ph.accpart + (size_t)j * ph.nseg * MAX_ACC + lane;
            double s = 0;
            for (int k = 0; k < nseg; ++k) s += src[(size_t)k * MAX_ACC];
            P->acc[lane] = s;
        }
    }
}

// Cholesky solve of the damped normal equations, packed lower triangle (row major) in place: the recurrences and their
// summation order are chol_solve<>'s (mdrp_math.h), only the storage differs (one lane per problem: registers matter)
template <int N>
__device__ __forceinline__ void chol_solve_packed(double *L /*in: A + lambda I, out: factor*/, const double *b, double *x) {
#define LME_TRI(i, j) ((i) * ((i) + 1) / 2 + (j))
    double inv[N]; // 1 / L(j, j) as in chol_solve<> (one reciprocal square root per column, products instead of quotients)
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = L[LME_TRI(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[LME_TRI(i, k)] * L[LME_TRI(j, k)];
            if (i == j) { inv[i] = lm_rsqrt(s); L[LME_TRI(i, i)] = s * inv[i]; }
            else L[LME_TRI(i, j)] = s * inv[j];
        }
    double y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[LME_TRI(i, k)] * y[k];
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s -= L[LME_TRI(k, i)] * x[k];
        x[i] = s * inv[i];
    }
#undef LME_TRI
}

// one lane per problem: gradient test, damped solve, step test, next candidate
template <int KIND, bool SHIFT>
__global__ __launch_bounds__(64) void k_lme_solve(LmePhase ph, int round) {
    constexpr int NP = LmTraits<KIND, SHIFT>::NP;
    constexpr int NT = NP * (NP + 1) / 2;
    const int j = blockIdx.x * 64 + threadIdx.x;
    bool alive = false;
    if (j < ph.batch) ph.pair_live[(size_t)((round + 1) & 1) * ph.batch + j] = 0; // the half the NEXT round counts into (its reader, the cost sweep two kernels back, is done)
    if (j < lme_count(ph)) {
        LmProb *P = ph.probs + j;
        if (P->status != 0) {
            bool done = false;
            double L[NT], g[NP], sol[NP];
#pragma unroll
            for (int a = 0; a < NP; ++a) g[a] = P->acc[NT + a];
            if (P->recompute) {
                double gn = 0;
#pragma unroll
                for (int a = 0; a < NP; ++a) gn += g[a] * g[a];
                if (sqrt(gn) < P->grad_tol) done = true;
            }
            if (!done) {
                const double lambda = P->lambda;
                int idx = 0;
#pragma unroll
                for (int a = 0; a < NP; ++a) {
#pragma unroll
                    for (int b = 0; b <= a; ++b, ++idx) L[idx] = P->acc[idx] + (a == b ? lambda : 0.0);
                }
                chol_solve_packed<NP>(L, g, sol);
                double sn = 0;
#pragma unroll
                for (int a = 0; a < NP; ++a) { sol[a] = -sol[a]; sn += sol[a] * sol[a]; }
                if (sqrt(sn) < P->step_tol) done = true;
            }
            if (done) { P->status = 0; P->has_cand = 0; }
            else {
                double full[LM_NPAR];
#pragma unroll
                for (int q = 0; q < LM_NPAR; ++q) full[q] = 0;
#pragma unroll
                for (int q = 0; q < NP; ++q) full[lm_col<KIND, SHIFT>(q)] = sol[q];
                if (KIND == 1) full[10] = full[9];
                const Model m = P->m;
                Model cand;
                lm_apply_step(m, full, KIND != 0, KIND == 0 && SHIFT, cand);
                P->cand = cand;
                LmState cs;
                lm_state_from_model(cand, KIND != 0, cs);
                P->cs = cs;
                P->has_cand = 1;
                alive = true;
                atomicAdd(ph.pair_live + (size_t)(round & 1) * ph.batch + P->pair, 1);
            }
            P->recompute = 0;
        }
    }
    const unsigned long long lb = __ballot(alive);
    if (threadIdx.x == 0 && lb) atomicAdd(ph.live + (round & (LME_RING - 1)), __popcll(lb));
}

// ------------------------------------------------------------------------------------------------ closing score sweep
// exact MSAC score (score_model of the estimators) of the current model of every problem, per (problem, segment);
// mask_out: get_inliers mask of the model (the final phase has one problem per pair)
template <int KIND>
__global__ __launch_bounds__(64) void k_lme_score(LmePhase ph, const PairState *__restrict__ st, const double *__restrict__ pts,
                                                  uint8_t *__restrict__ mask_out) {
    const int pair = blockIdx.y, seg = blockIdx.x, lane = threadIdx.x;
    const PairState &ps = st[pair];
    const int n = ps.n;
    if (seg * LME_SEG >= n) return;
    const int cnt = lme_count(ph);
    const int j0 = max(ph.pfx[pair] - ph.first, 0), j1 = min(ph.pfx[pair + 1] - ph.first, cnt);
    const double *pp = pts + (size_t)pair * ph.n_max * PT_STRIDE;
    const double thr = ps.sq_thr;
    for (int j = j0; j < j1; ++j) {
        const Model m = ph.probs[j].m;
        double R[9], E[9], Em[9];
        quat_to_R(m.q, R);
        essential_from_Rt(R, m.t, Em);
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) E[i] = Em[i];
        } else fundamental_from_E(Em, m.f1, m.f2, E);
        double score = 0;
        int c = 0;
#pragma unroll
        for (int r = 0; r < LME_RPT; ++r) {
            const int i = seg * LME_SEG + r * 64 + lane;
            if (i < n) {
                double s1 = 0;
                int c1 = 0;
                score_point<KIND == 0>(pp + (size_t)i * PT_STRIDE, E, R, m.t, thr, s1, c1);
                score += s1; c += c1;
                if (mask_out) mask_out[(size_t)pair * ph.n_max + i] = (uint8_t)c1;
            }
        }
        score = wave_sum_swap(score);
        c = wave_sum_swap_i(c);
        if (lane == 0) { ph.part[(size_t)j * ph.nseg + seg] = score; ph.ipart[(size_t)j * ph.nseg + seg] = c; }
    }
}

__device__ __forceinline__ void lme_score_total(const LmePhase &ph, int j, int n, double thr, double &score, int &cnt) {
    const int nseg = (n + LME_SEG - 1) / LME_SEG;
    double s = 0;
    int c = 0;
    for (int k = 0; k < nseg; ++k) { s += ph.part[(size_t)j * ph.nseg + k]; c += ph.ipart[(size_t)j * ph.nseg + k]; }
    cnt = c;
    score = s + thr * (double)(n - c);
}

__device__ __forceinline__ void lme_start(LmProb &P, const Model &m0, bool focal, int pair, int n, double scale_reproj, double ws,
                                          const LmOpt &o) {
    P.m = m0; P.cand = m0;
    lm_state_from_model(m0, focal, P.cs);
    P.sqrt_sr = sqrt(scale_reproj); P.ws = ws; P.loss_scale = o.loss_scale; P.mu = 0.5;
    P.status = 1; P.has_cand = 2; P.loss = o.loss; P.cur = 0;
    P.pair = pair; P.n = n; P.it = 0; P.max_it = o.max_it; P.recompute = 0; P.need_acc = 0; P.n_eff = n; P.pad3_ = 0;
    P.cost = 0; P.lambda = o.lambda0; P.grad_tol = o.grad_tol; P.step_tol = o.step_tol; P.lambda_min = o.lambda_min; P.lambda_max = o.lambda_max;
}

// the evaluation counters of a problem go into stats[0 .. 1] (one atomic per wavefront)
__device__ __forceinline__ void lme_flush_evals(const LmProb *P /*or null*/, unsigned long long *__restrict__ stats) {
    if (!stats) return;
    unsigned long long a = P ? P->ev_cost : 0ull, b = P ? P->ev_acc : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { if (a) atomicAdd(stats, a); if (b) atomicAdd(stats + 1, b); }
}

// ------------------------------------------------------------------------------------------------ LO phase (refine_model)
// one thread per trigger of the chunk's frozen plan (k_lo_plan): 25 iterations, TRUNCATED at the epipolar threshold
MDRP_GLOBAL void k_lme_lo_init(LmePhase ph, RunParams rp, const PairState *__restrict__ st, const Model *__restrict__ models,
                              const Trigger *__restrict__ triggers, int trig_cap, const int32_t *__restrict__ plan) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lme_count(ph)) return;
    const int w = ph.first + j;
    const int32_t *prefix = plan, *begin = plan + rp.batch + 1;
    const int pair = plan_find(prefix, rp.batch, w);
    const int pos = begin[pair] + (w - prefix[pair]);
    const PairState &ps = st[pair];
    const Trigger &tr = triggers[(size_t)pair * trig_cap + pos];
    const Model m0 = models[(size_t)pair * rp.slot_stride + (size_t)tr.iter * rp.mps + tr.k_ref];
    LmOpt o;
    o.max_it = 25; o.loss = 1; o.loss_scale = ps.lo_loss_scale;
    o.grad_tol = 1e-10; o.step_tol = 1e-8; o.lambda0 = 1e-3; o.lambda_min = 1e-10; o.lambda_max = 1e10;
    lme_start(ph.probs[j], m0, rp.kind != 0, pair, ps.n, ps.scale_reproj, rp.weight_sampson, o);
    ph.probs[j].ev_cost = 0; ph.probs[j].ev_acc = 0;
}

MDRP_GLOBAL void k_lme_lo_finish(LmePhase ph, RunParams rp, const PairState *__restrict__ st, Trigger *__restrict__ triggers, int trig_cap,
                                const int32_t *__restrict__ plan, unsigned long long *__restrict__ lm_stats) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = j < lme_count(ph);
    lme_flush_evals(mine ? ph.probs + j : nullptr, lm_stats);
    if (!mine) return;
    const int w = ph.first + j;
    const int32_t *prefix = plan, *begin = plan + rp.batch + 1;
    const int pair = plan_find(prefix, rp.batch, w);
    const int pos = begin[pair] + (w - prefix[pair]);
    const PairState &ps = st[pair];
    Trigger &tr = triggers[(size_t)pair * trig_cap + pos];
    double sc;
    int cn;
    lme_score_total(ph, j, ps.n, ps.sq_thr, sc, cn);
    tr.refined = ph.probs[j].m; tr.ref_score = sc; tr.ref_cnt = cn;
}

// ------------------------------------------------------------------------------------------------ final phase
// ransac<> tail (@0x22f1d0-0x22f295) + get_inliers (@0x4f7a10/@0x4f77f0) + the estimator's inlier-only refinement
// (@0x2247c3 / @0x223815) + focal un-normalisation, one problem per pair.
// stage 0: result header, mask rows of short pairs, LO from the best model (one wavefront per pair)
MDRP_GLOBAL __launch_bounds__(64) void k_lme_fin_init(LmePhase ph, RunParams rp, const PairState *__restrict__ st, uint8_t *__restrict__ mask_all,
                                                     ResultDev *__restrict__ results) {
    const int pair = blockIdx.x;
    const PairState &ps = st[pair];
    uint8_t *mask = mask_all + (size_t)pair * rp.n_max;
    for (int i = (ps.n < 3 ? 0 : ps.n) + threadIdx.x; i < rp.n_max; i += 64) mask[i] = 0;
    if (threadIdx.x != 0) return;
    ResultDev res;
    res.model = ps.best;
    res.refinements = ps.refinements; res.iterations = ps.iterations; res.num_inliers = ps.num_inliers;
    res.inlier_ratio = ps.inlier_ratio; res.model_score = ps.model_score;
    results[pair] = res;
    LmProb &P = ph.probs[pair];
    P.ev_cost = 0; P.ev_acc = 0;
    if (ps.n < 3) { P.status = 0; P.has_cand = 0; P.n = 0; P.pair = pair; P.m = ps.best; return; }
    LmOpt o;
    o.max_it = 25; o.loss = 1; o.loss_scale = ps.lo_loss_scale;
    o.grad_tol = 1e-10; o.step_tol = 1e-8; o.lambda0 = 1e-3; o.lambda_min = 1e-10; o.lambda_max = 1e10;
    lme_start(P, ps.best, rp.kind != 0, pair, ps.n, ps.scale_reproj, rp.weight_sampson, o);
}

// stage 1 (after the LO and its score sweep): adopt the refined model if it scores better; the model whose inliers the
// mask sweep then marks is left in P.m
MDRP_GLOBAL void k_lme_fin_select(LmePhase ph, RunParams rp, const PairState *__restrict__ st, ResultDev *__restrict__ results) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= rp.batch) return;
    const PairState &ps = st[pair];
    if (ps.n < 3) return;
    LmProb &P = ph.probs[pair];
    double sc;
    int cn;
    lme_score_total(ph, pair, ps.n, ps.sq_thr, sc, cn);
    results[pair].refinements = ps.refinements + 1;
    if (sc < ps.model_score) results[pair].num_inliers = (uint64_t)cn; // score / ratio NOT updated (reference)
    else P.m = ps.best;
}

// stage 2 (after the mask sweep): the inlier-only refinement with the user's bundle options
MDRP_GLOBAL void k_lme_fin_init2(LmePhase ph, RunParams rp, const PairState *__restrict__ st, const ResultDev *__restrict__ results) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= rp.batch) return;
    const PairState &ps = st[pair];
    if (ps.n < 3 || !(results[pair].num_inliers > (rp.kind == 2 ? 7u : 3u))) return; // (the wrappers' thresholds: k_final)
    LmOpt f;
    f.max_it = rp.final_max_it; f.loss = rp.final_loss; f.loss_scale = ps.final_loss_scale;
    f.grad_tol = rp.grad_tol; f.step_tol = rp.step_tol; f.lambda0 = rp.lambda0; f.lambda_min = rp.lambda_min; f.lambda_max = rp.lambda_max;
    const Model m0 = ph.probs[pair].m;
    lme_start(ph.probs[pair], m0, rp.kind != 0, pair, ps.n, ps.scale_reproj, rp.weight_sampson, f);
    ph.probs[pair].n_eff = (int32_t)results[pair].num_inliers; // the record mask of this phase = the inliers
}

MDRP_GLOBAL void k_lme_fin_write(LmePhase ph, RunParams rp, const PairState *__restrict__ st, ResultDev *__restrict__ results,
                                unsigned long long *__restrict__ lm_stats) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    lme_flush_evals(pair < rp.batch ? ph.probs + pair : nullptr, lm_stats);
    if (pair >= rp.batch) return;
    const PairState &ps = st[pair];
    if (ps.n < 3) return;
    Model best = ph.probs[pair].m;
    if (rp.kind != 0) { best.f1 *= ps.norm; best.f2 *= ps.norm; }
    results[pair].model = best;
}

// ------------------------------------------------------------------------------------------------ unit path (mdrp_refine_models)
MDRP_GLOBAL void k_lme_unit_init(LmePhase ph, int count, const Model *__restrict__ models, int kind, int n, double scale_reproj, double ws, LmOpt o) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    lme_start(ph.probs[j], models[j], kind != 0, 0, n, scale_reproj, ws, o);
    ph.probs[j].ev_cost = 0; ph.probs[j].ev_acc = 0;
}
MDRP_GLOBAL void k_lme_unit_finish(LmePhase ph, int count, Model *__restrict__ models, double *__restrict__ final_cost) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    models[j] = ph.probs[j].m;
    if (final_cost) final_cost[j] = ph.probs[j].cost;
}
// pfx = 0, 1, 2, ... (one problem per pair) or 0, count (all problems on pair 0); total
MDRP_GLOBAL void k_lme_iota(int32_t *pfx, int entries, int32_t *total, int total_value, int unit_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < entries) pfx[i] = unit_count >= 0 ? (i == 0 ? 0 : unit_count) : i;
    if (i == 0) *total = total_value;
}

} // namespace mdrp
