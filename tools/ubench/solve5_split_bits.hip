// Does the 5-point elimination give the same bits when its null space runs in a kernel of its own (mdrp_classic.h kc_solve5_null) as when both
// halves are inlined into one kernel?  Random samples, every one of Reduce5's 75 doubles compared bit by bit.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -o solve5_split_bits solve5_split_bits.hip ; run on the GPU box: ./solve5_split_bits [samples]
#include "../../mdrp_amd/csrc/mdrp_classic.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
using namespace mdrp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void load5(const double *__restrict__ in, int i, double (*x1h)[3], double (*x2h)[3]) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double *p = in + ((size_t)i * 5 + k) * 6;
        x1h[k][0] = p[0] * p[4]; x1h[k][1] = p[1] * p[4]; x1h[k][2] = p[4];
        x2h[k][0] = p[2] * p[5]; x2h[k][1] = p[3] * p[5]; x2h[k][2] = p[5];
    }
}
__global__ __launch_bounds__(64) void k_mono(const double *in, int n, double *out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    double x1h[5][3], x2h[5][3];
    load5(in, i, x1h, x2h);
    Reduce5 r5;
    const bool ok = relpose_5pt_reduce(x1h, x2h, lds_solve5_store(), r5);
    const double *src = &r5.El[0][0][0];
    for (int k = 0; k < 75; ++k) out[(size_t)i * 76 + k] = src[k];
    out[(size_t)i * 76 + 75] = ok;
}
__global__ __launch_bounds__(64, 2) void k_null(const double *in, int n, double *out) {
    extern __shared__ double solve5_lds[];
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    double x1h[5][3], x2h[5][3], El[3][3][4];
    load5(in, i, x1h, x2h);
    relpose_5pt_nullspace(x1h, x2h, solve5_lds + threadIdx.x, 64, El);
    const double *src = &El[0][0][0];
    for (int k = 0; k < 36; ++k) out[(size_t)i * 76 + k] = src[k];
}
__global__ __launch_bounds__(64) void k_elim(int n, double *out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    Reduce5 r5;
    double *de = &r5.El[0][0][0];
    for (int k = 0; k < 36; ++k) de[k] = out[(size_t)i * 76 + k];
    const bool ok = relpose_5pt_eliminate(lds_solve5_store(), r5);
    const double *src = &r5.bx[0][0];
    for (int k = 36; k < 75; ++k) out[(size_t)i * 76 + k] = src[k - 36];
    out[(size_t)i * 76 + 75] = ok;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1 << 16;
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> u(-0.6, 0.6);
    std::vector<double> in((size_t)n * 30);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 5; ++k) {
            double *p = &in[((size_t)i * 5 + k) * 6];
            p[0] = u(g); p[1] = u(g); p[2] = p[0] + 0.1 * u(g); p[3] = p[1] + 0.1 * u(g);
            p[4] = 1.0 / sqrt(p[0] * p[0] + p[1] * p[1] + 1.0); p[5] = 1.0 / sqrt(p[2] * p[2] + p[3] * p[3] + 1.0);
        }
    double *d_in, *d_a, *d_b;
    CK(hipMalloc(&d_in, in.size() * 8)); CK(hipMalloc(&d_a, (size_t)n * 76 * 8)); CK(hipMalloc(&d_b, (size_t)n * 76 * 8));
    CK(hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice));
    const int blocks = (n + 63) / 64;
    hipLaunchKernelGGL(k_mono, dim3(blocks), dim3(64), SOLVE5_LDS_BYTES, 0, d_in, n, d_a);
    hipLaunchKernelGGL(k_null, dim3(blocks), dim3(64), SOLVE5N_LDS_BYTES, 0, d_in, n, d_b);
    hipLaunchKernelGGL(k_elim, dim3(blocks), dim3(64), SOLVE5_LDS_BYTES, 0, n, d_b);
    CK(hipDeviceSynchronize());
    std::vector<double> a((size_t)n * 76), b((size_t)n * 76);
    CK(hipMemcpy(a.data(), d_a, a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_b, b.size() * 8, hipMemcpyDeviceToHost));
    long diff_el = 0, diff_rest = 0, shown = 0;
    for (int i = 0; i < n; ++i) {
        bool de = false, dr = false;
        int first = -1;
        for (int k = 0; k < 76; ++k)
            if (memcmp(&a[(size_t)i * 76 + k], &b[(size_t)i * 76 + k], 8)) { (k < 36 ? de : dr) = true; if (first < 0) first = k; }
        diff_el += de; diff_rest += (dr && !de);
        if ((de || dr) && shown++ < 5) printf("sample %d: first differing element %d  mono %.17g  split %.17g\n", i, first, a[(size_t)i * 76 + first], b[(size_t)i * 76 + first]);
    }
    printf("samples %d: null space differs on %ld, elimination alone differs on %ld\n", n, diff_el, diff_rest);
    return 0;
}
