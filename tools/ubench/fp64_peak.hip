// fp64 VALU microbenchmark (gfx950): independent v_fma_f64 chains per lane on every CU; reports achieved TFLOP/s,
// the in-kernel clock (s_memtime / s_memrealtime) and cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return; } } while (0)
template <int CH>
__global__ __launch_bounds__(256) void fma_chains(double *out, unsigned long long *stamps, int iters, double a, double b) {
    double x[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) x[i] = threadIdx.x * 1e-9 + i;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) x[i] = fma(x[i], a, b);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int CH>
void run(int blocks_per_cu, int iters) {
    const int ncu = 256, nb = ncu * blocks_per_cu;
    double *out; unsigned long long *st;
    CK(hipMalloc(&out, sizeof(double) * nb * 256));
    CK(hipMalloc(&st, sizeof(unsigned long long) * 2 * nb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) fma_chains<CH><<<nb, 256>>>(out, st, iters, 1.0000001, 1e-9);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    fma_chains<CH><<<nb, 256>>>(out, st, iters, 1.0000001, 1e-9);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * nb);
    CK(hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost));
    std::vector<double> clk(nb), cyc(nb);
    for (int i = 0; i < nb; ++i) { clk[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 100e6; cyc[i] = (double)h[2 * i]; }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double fmas = (double)nb * 256 * (double)iters * CH;
    // waves per SIMD = blocks_per_cu (256 threads = 4 waves = 1 per SIMD)
    printf("chains %2d waves/SIMD %d: %7.2f ms  %6.2f TFLOP/s  clock %.2f GHz  %.2f cycles per wave-FMA per SIMD (in-kernel)\n", CH, blocks_per_cu, ms,
           2 * fmas / (ms * 1e-3) / 1e12, clk[nb / 2] / 1e9, cyc[nb / 2] / ((double)iters * CH * blocks_per_cu));
    CK(hipFree(out)); CK(hipFree(st));
}
int main() {
    run<1>(1, 400000); run<2>(1, 400000); run<4>(1, 400000); run<8>(1, 400000); run<16>(1, 200000);
    run<4>(2, 400000); run<8>(2, 400000); run<4>(4, 200000); run<8>(4, 200000); run<4>(8, 100000); run<8>(8, 100000);
    return 0;
}
