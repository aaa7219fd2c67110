// Kernel instantiations that live in their own translation units (compiled in parallel by mdrp_amd/build.py; mdrp_tu.hip defines them,
// mdrp_capi.hip declares them `extern template`).  MDRP_INST is `extern` in the declaring unit and empty in the defining one.
//   group 1 / 2: k_final<KIND, SHIFT, T = 64 / 256, FLOSS> for the four LM estimators x the six loss types of BundleOptions
//   group 3:     the kernels of the 5- / 6- / 7-point baselines (mdrp_classic.h)
#pragma once
namespace mdrp {
#define MDRP_FINAL_PARAMS RunParams, PairState *, const double *, const double *, uint8_t *, ResultDev *, int, int, unsigned long long *, const int32_t *, int32_t *, \
                          unsigned long long, unsigned long long *
#define MDRP_FINAL_LOSSES(X, K, S, T) X(K, S, T, 0) X(K, S, T, 1) X(K, S, T, 2) X(K, S, T, 3) X(K, S, T, 4) X(K, S, T, 5)
#define MDRP_FINAL_KINDS(X, T) MDRP_FINAL_LOSSES(X, 0, false, T) MDRP_FINAL_LOSSES(X, 0, true, T) MDRP_FINAL_LOSSES(X, 1, false, T) MDRP_FINAL_LOSSES(X, 2, false, T)
#define MDRP_FINAL_ONE(K, S, T, L) MDRP_INST template __global__ void k_final<K, S, T, L>(MDRP_FINAL_PARAMS);

#define MDRP_CLASSIC_KINDS_T(X, T) X(CLASSIC_RELPOSE, T) X(CLASSIC_SHARED, T) X(CLASSIC_FUND, T)
#define MDRP_KC_LO_ONE(CK, T) MDRP_INST template __global__ void kc_lo<CK, T>(RunParams, const PairState *, const double *, const Model *, Trigger *, int, const int32_t *, int32_t *, \
                                                                               int32_t *, uint8_t *, int, FuseTail);
#define MDRP_KC_FINAL_ONE(CK, T) MDRP_INST template __global__ void kc_final<CK, T>(RunParams, PairState *, const double *, uint8_t *, ResultDev *, const int32_t *, int32_t *, \
                                                                                     unsigned long long, unsigned long long *, int);
#define MDRP_KC_REFINE_ONE(CK, T) MDRP_INST template __global__ void kc_refine_unit<CK, T>(int, Model *, const double *, int, LmOpt, double *);
#define MDRP_KC_SOLVE_ONE(CK) MDRP_INST template __global__ void kc_solve<CK>(RunParams, const PairState *, const uint32_t *, const double *, Model *, int32_t *, uint32_t *, int32_t *);
#define MDRP_KC_UNIT_ONE(CK) MDRP_INST template __global__ void kc_solver_unit<CK>(int, const double *, const double *, Model *, int32_t *);
#define MDRP_KC_SAMPLES_ONE(K) MDRP_INST template __global__ void kc_samples<K>(int, const int32_t *, uint64_t *, int, uint32_t *);

#define MDRP_INSTANCES_FINAL_64 MDRP_FINAL_KINDS(MDRP_FINAL_ONE, 64)
#define MDRP_INSTANCES_FINAL_256 MDRP_FINAL_KINDS(MDRP_FINAL_ONE, 256)
#define MDRP_INSTANCES_CLASSIC                                                                                                       \
    MDRP_CLASSIC_KINDS_T(MDRP_KC_LO_ONE, 64) MDRP_CLASSIC_KINDS_T(MDRP_KC_LO_ONE, 256)                                              \
    MDRP_CLASSIC_KINDS_T(MDRP_KC_FINAL_ONE, 64) MDRP_CLASSIC_KINDS_T(MDRP_KC_FINAL_ONE, 256)                                        \
    MDRP_CLASSIC_KINDS_T(MDRP_KC_REFINE_ONE, 64) MDRP_CLASSIC_KINDS_T(MDRP_KC_REFINE_ONE, 256)                                      \
    MDRP_KC_SOLVE_ONE(CLASSIC_SHARED) MDRP_KC_SOLVE_ONE(CLASSIC_FUND) /* (the 5-point solver: kc_solve5_reduce + kc_solve5_roots, main unit) */ \
    MDRP_KC_UNIT_ONE(CLASSIC_RELPOSE) MDRP_KC_UNIT_ONE(CLASSIC_SHARED) MDRP_KC_UNIT_ONE(CLASSIC_FUND)                               \
    MDRP_KC_SAMPLES_ONE(5) MDRP_KC_SAMPLES_ONE(6) MDRP_KC_SAMPLES_ONE(7)
} // namespace mdrp
