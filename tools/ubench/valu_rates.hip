// VALU issue-rate microbenchmark (gfx950): cycles per wave-instruction per SIMD for the instruction kinds the scoring kernels
// use — v_fma_f32, v_pk_fma_f32, v_pk_add_f32, v_fma_f64, v_cmp + v_addc (through an SGPR pair), v_perm_b32, v_bcnt — and
// for v_mfma_f32_16x16x32_bf16 alone and interleaved with packed VALU work (does the vector pipe run beside the matrix pipe?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *stamps, int iters, float a, float b) {
    f32x2 x[8];
    double d[8];
    unsigned u[8];
    f32x4 acc[4];
    const uint4 raw = make_uint4(0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    const bf16x8 fa = __builtin_bit_cast(bf16x8, raw), fb = fa;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = (f32x2){threadIdx.x * 1e-6f + i, 1.0f + i}; d[i] = threadIdx.x * 1e-9 + i; u[i] = threadIdx.x + i; }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x2 av = {a, a}, bv = {b, b};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(a), "v"(b));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(bv));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"((double)a), "v"((double)b));
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                unsigned long long k0, k1, co;
                asm volatile("v_cmp_gt_f32_e64 %2, |%5|, %7\n\tv_cmp_gt_f32_e64 %3, |%6|, %7\n\t"
                             "v_addc_co_u32_e64 %0, %4, %0, 0, %2\n\tv_addc_co_u32_e64 %1, %4, %1, 0, %3"
                             : "+v"(u[i]), "+v"(u[i + 1]), "=&s"(k0), "=&s"(k1), "=&s"(co) : "v"(x[i].x), "v"(x[i + 1].x), "v"(a));
            }
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "s"(0x07030c0cu));
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        } else if (MODE == 7) { // MFMA alone: 4 independent accumulators
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
        } else if (MODE == 8) { // per MFMA: 4 packed VALU (the k_count mix)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
                asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3" : "+v"(x[2 * i]), "+v"(x[2 * i + 1]) : "v"(av), "v"(bv));
                asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %2" : "+v"(x[2 * i]), "+v"(x[2 * i + 1]) : "v"(bv));
            }
        } else if (MODE == 9) { // per MFMA: 4 plain fp32 VALU
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
                asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x[2 * i].x), "+v"(x[2 * i + 1].x) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x[2 * i].y), "+v"(x[2 * i + 1].y) : "v"(a), "v"(b));
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y + (float)d[i] + (float)u[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) stamps[blockIdx.x] = c1 - c0;
}

template <int MODE>
void run(const char *name, int blocks_per_cu, int iters, int per_iter) {
    const int nb = 256 * blocks_per_cu;
    float *out; unsigned long long *st;
    CK(hipMalloc(&out, sizeof(float) * nb * 256));
    CK(hipMalloc(&st, sizeof(unsigned long long) * nb));
    k<MODE><<<nb, 256>>>(out, st, iters, 1.0000001f, 1e-9f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    k<MODE><<<nb, 256>>>(out, st, iters, 1.0000001f, 1e-9f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nb);
    CK(hipMemcpy(h.data(), st, sizeof(unsigned long long) * nb, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%-34s waves/SIMD %d: %7.3f ms   %6.2f cycles per instruction per SIMD (in-kernel, %d instructions per trip)\n", name, blocks_per_cu, ms,
           (double)h[nb / 2] / ((double)iters * per_iter * blocks_per_cu), per_iter);
    CK(hipFree(out)); CK(hipFree(st));
}
int main() {
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_fma_f32", 1, 100000, 8); run<1>("v_pk_fma_f32", 1, 100000, 8); run<2>("v_pk_add_f32", 1, 100000, 8); run<3>("v_fma_f64", 1, 100000, 8);
                      run<4>("v_cmp + v_addc (SGPR pair)", 1, 100000, 16); run<5>("v_perm_b32", 1, 100000, 8); run<6>("v_bcnt_u32_b32", 1, 100000, 8);
                      run<7>("v_mfma_f32_16x16x32_bf16", 1, 100000, 4); run<8>("mfma + 4 v_pk (per MFMA)", 1, 100000, 4); run<9>("mfma + 4 v_fma_f32 (per MFMA)", 1, 100000, 4); }
        if (w == 2) { run<0>("v_fma_f32", 2, 100000, 8); run<1>("v_pk_fma_f32", 2, 100000, 8); run<2>("v_pk_add_f32", 2, 100000, 8); run<3>("v_fma_f64", 2, 100000, 8);
                      run<4>("v_cmp + v_addc (SGPR pair)", 2, 100000, 16); run<5>("v_perm_b32", 2, 100000, 8); run<6>("v_bcnt_u32_b32", 2, 100000, 8);
                      run<7>("v_mfma_f32_16x16x32_bf16", 2, 100000, 4); run<8>("mfma + 4 v_pk (per MFMA)", 2, 100000, 4); run<9>("mfma + 4 v_fma_f32 (per MFMA)", 2, 100000, 4); }
        if (w == 4) { run<0>("v_fma_f32", 4, 50000, 8); run<1>("v_pk_fma_f32", 4, 50000, 8); run<2>("v_pk_add_f32", 4, 50000, 8); run<3>("v_fma_f64", 4, 50000, 8);
                      run<4>("v_cmp + v_addc (SGPR pair)", 4, 50000, 16); run<5>("v_perm_b32", 4, 50000, 8); run<6>("v_bcnt_u32_b32", 4, 50000, 8);
                      run<7>("v_mfma_f32_16x16x32_bf16", 4, 50000, 4); run<8>("mfma + 4 v_pk (per MFMA)", 4, 50000, 4); run<9>("mfma + 4 v_fma_f32 (per MFMA)", 4, 50000, 4); }
    }
    return 0;
}
