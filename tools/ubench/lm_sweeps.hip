// The two LM sweeps of mdrp_kernels.h in isolation (gfx950): one wavefront per workgroup runs `reps` cost sweeps (lm_cost) and `reps`
// normal-equation sweeps (lm_accumulate) over the records of one of `pairs` synthetic pairs; every SIMD of the chip holds `waves`
// wavefronts.  Reports microseconds per sweep, nanoseconds per record trip and shader cycles per trip — what a sweep costs when nothing
// else is in the way (no solver next door, records in L2 when `pairs` is small), to compare with the trace of the real kernel
// (tools/lo_trace.py).  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast [-DMDRP_LM_COST_DEPTH=2 ...] -o lm_sweeps lm_sweeps.hip
// Run on the GPU box: ./lm_sweeps [n = 2000] [pairs = 8] [waves per SIMD = 1] [reps = 20]
#include "../../mdrp_amd/csrc/mdrp_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <algorithm>
using namespace mdrp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int KIND, bool SHIFT, int MODE>
__global__ __launch_bounds__(64, MDRP_LM_MINWAVES) void k_sweeps(const double *__restrict__ pts, const double *__restrict__ dep, int n, int pairs, int reps,
                                                                Model m, double thr, int stride, double *out, unsigned long long *ticks) {
    extern __shared__ uint16_t dyn_list[];
    __shared__ LmShared sh;
    if (threadIdx.x == 0) { sh.list = dyn_list; sh.stride = stride; sh.stats = nullptr; sh.ev[0] = 0; sh.ev[1] = 0; }
    __syncthreads();
    const int pair = blockIdx.x % pairs;
    const double *pp = pts + (size_t)pair * n * PT_STRIDE, *dd = dep + (size_t)pair * n * 2;
    LmOpt o;
    o.max_it = 25; o.loss = 1; o.loss_scale = thr; o.grad_tol = 1e-10; o.step_tol = 1e-8; o.lambda0 = 1e-3; o.lambda_min = 1e-10; o.lambda_max = 1e10;
    constexpr int NP = LmTraits<KIND, SHIFT>::NP;
    double acc[NP * (NP + 1) / 2 + NP];
    double s = 0;
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        m.t[0] += 1e-9; // (a different model every time: nothing is hoisted out of the repetition)
        if (MODE == 0) s += lm_cost<KIND, 64, 1>(m, pp, dd, n, nullptr, 1.0, 1.0, o, sh, r & 1);
        else {
            if (r == 0) s += lm_cost<KIND, 64, 1>(m, pp, dd, n, nullptr, 1.0, 1.0, o, sh, 0); // the work list
            lm_accumulate<KIND, SHIFT, 64, 1>(m, pp, dd, n, nullptr, 1.0, 1.0, o, acc, sh, 0);
            s += acc[0] + acc[NP];
        }
    }
    const unsigned long long w1 = wall_clock64(), c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = s; ticks[2 * blockIdx.x] = w1 - w0; ticks[2 * blockIdx.x + 1] = c1 - c0; out[gridDim.x + blockIdx.x] = (double)sh.count[0][0]; }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2000, pairs = argc > 2 ? atoi(argv[2]) : 8, waves = argc > 3 ? atoi(argv[3]) : 1, reps = argc > 4 ? atoi(argv[4]) : 20;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 4 * waves;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> u(-0.5, 0.5), dz(2.0, 6.0);
    // a scene the model fits: X1 = d1 (x1, 1), X2 = R X1 + t, x2 = X2.xy / X2.z (half the records), random x2 for the rest
    const double t[3] = {0.3, -0.1, 0.05};
    std::vector<double> pts((size_t)pairs * n * PT_STRIDE), dep((size_t)pairs * n * 2);
    for (size_t i = 0; i < (size_t)pairs * n; ++i) {
        const double x = u(rng), y = u(rng), d1 = dz(rng);
        const double X2[3] = {d1 * x + t[0], d1 * y + t[1], d1 + t[2]};
        const bool in = (i & 1) == 0;
        double *p = pts.data() + i * PT_STRIDE;
        p[0] = x; p[1] = y; p[2] = in ? X2[0] / X2[2] + 1e-4 * u(rng) : u(rng); p[3] = in ? X2[1] / X2[2] + 1e-4 * u(rng) : u(rng); p[4] = 0; p[5] = 0;
        dep[2 * i] = d1; dep[2 * i + 1] = in ? X2[2] : dz(rng);
    }
    Model m{};
    m.q[0] = 1; m.t[0] = t[0]; m.t[1] = t[1]; m.t[2] = t[2]; m.scale = 1; m.f1 = 1; m.f2 = 1;
    double *d_pts, *d_dep, *d_out;
    unsigned long long *d_ticks;
    CK(hipMalloc(&d_pts, pts.size() * 8)); CK(hipMalloc(&d_dep, dep.size() * 8)); CK(hipMalloc(&d_out, 2 * blocks * 8)); CK(hipMalloc(&d_ticks, 2 * blocks * 8));
    CK(hipMemcpy(d_pts, pts.data(), pts.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_dep, dep.data(), dep.size() * 8, hipMemcpyHostToDevice));
    const int stride = ((n + 63) / 64) * 64;
    const size_t smem = 2 * (size_t)stride * sizeof(uint16_t);
    std::vector<unsigned long long> ticks(2 * blocks);
    std::vector<double> out(2 * blocks);
    const char *names[2] = {"cost sweep (lm_cost)", "normal equations (lm_accumulate)"};
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { // (the first launch warms the caches and the clock)
            if (mode == 0) hipLaunchKernelGGL((k_sweeps<0, false, 0>), dim3(blocks), dim3(64), smem, 0, d_pts, d_dep, n, pairs, reps, m, 0.01, stride, d_out, d_ticks);
            else hipLaunchKernelGGL((k_sweeps<0, false, 1>), dim3(blocks), dim3(64), smem, 0, d_pts, d_dep, n, pairs, reps, m, 0.01, stride, d_out, d_ticks);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(ticks.data(), d_ticks, ticks.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> us(blocks), cyc(blocks);
        for (int b = 0; b < blocks; ++b) { us[b] = ticks[2 * b] / 100.0 / reps; cyc[b] = (double)ticks[2 * b + 1] / reps; }
        std::sort(us.begin(), us.end());
        const double med = us[blocks / 2], listed = out[blocks];
        const double trips = mode == 0 ? (n + 63) / 64 : (listed + 63) / 64;
        printf("%-34s n %d, %d pairs, %d wave(s)/SIMD: median %.2f us per sweep (p10 %.2f, p90 %.2f), %.0f record trips -> %.0f ns per trip; list %0.f records\n",
               names[mode], n, pairs, waves, med, us[blocks / 10], us[blocks * 9 / 10], trips, 1e3 * med / trips, listed);
    }
    return 0;
}
