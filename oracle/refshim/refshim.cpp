// refshim — flat C wrappers around the REFERENCE's own compiled PoseLib (binary only).
//
// TEST INFRASTRUCTURE, THIS CONTAINER ONLY.  The reference arithmetic for the RePoseD hot path
// lives in a third-party binary shipped inside /root/reference/demo/poselib-2.0.5-cp312-*.whl
// (PoseLib 2.0.5, kocurvik/PoseLib@pr-mdrp).  No source for it exists under /root/reference, so
// it cannot be compiled; it CAN be dlopen()ed.  This file is our own
// code: it declares ABI-compatible plain structs and calls the exported C++ entry points by their
// mangled names.  It is used to (1) pin oracle/mdrp_oracle.c function by function and (2) emit the
// golden vectors committed under tests/golden/ (tests/tools/gen_golden.py).  Neither the wheel nor its
// .so is ever copied into this repository; the built shim lands in oracle/_ref/ (git-ignored) and
// is useless on the GPU box (the wheel does not travel).
//
// Build: oracle/build_ref.sh   (needs /root/reference; extracts the wheel to /tmp/mdrp_ref_whl)
#include <dlfcn.h>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

struct V3 { double v[3]; };
struct alignas(16) V2 { double v[2]; };
struct alignas(32) CameraPose { double q[4]; double t[3]; };
struct alignas(32) MDG { CameraPose pose; double scale, shift1, shift2; };
struct Camera { int model_id, width, height; std::vector<double> params; };
struct alignas(32) MDIP { MDG geometry; Camera camera1, camera2; };
struct RansacOptions {
    size_t max_iterations, min_iterations;
    double dyn_num_trials_mult, success_prob, max_reproj_error, max_epipolar_error;
    unsigned long seed;
    bool progressive_sampling;
    size_t max_prosac_iterations;
    bool real_focal_check, score_initial_model, monodepth_estimate_shift;
    float monodepth_weight_sampson;
};
struct BundleOptions {
    size_t max_iterations;
    int loss_type;
    double loss_scale, gradient_tol, step_tol, initial_lambda, min_lambda, max_lambda;
    bool verbose;
};
struct RansacStats { size_t refinements, iterations, num_inliers; double inlier_ratio, model_score; };
struct BundleStats { size_t iterations; double initial_cost, cost, lambda; size_t invalid_steps; double step_norm, grad_norm; };

typedef std::vector<V3> VV3;
typedef std::vector<V2> VV2;
typedef std::vector<double> VD;

typedef int (*solver3_t)(const VV3 &, const VV3 &, const VD &, const VD &, std::vector<MDG> *);
typedef int (*solverf_t)(const VV3 &, const VV3 &, const VD &, const VD &, std::vector<MDIP> *);
typedef int (*p3p_t)(const VV3 &, const VV3 &, std::vector<CameraPose> *);
typedef double (*msac_pose_t)(const CameraPose &, const VV2 &, const VV2 &, double, size_t *);
typedef double (*msac_F_t)(const double *, const VV2 &, const VV2 &, double, size_t *);
typedef int (*inl_pose_t)(const CameraPose &, const VV2 &, const VV2 &, double, std::vector<char> *);
typedef int (*inl_F_t)(const double *, const VV2 &, const VV2 &, double, std::vector<char> *);
typedef bool (*cheir_t)(const CameraPose &, const V3 &, const V3 &, double);
typedef BundleStats (*refine_calib_t)(const VV2 &, const VV2 &, const VD &, const VD &, MDG *, double, double,
                                      const BundleOptions &, bool, const VD &);
typedef BundleStats (*refine_focal_t)(const VV2 &, const VV2 &, const VD &, const VD &, MDIP *, double, double,
                                      const BundleOptions &, const VD &);
typedef RansacStats (*ransac_calib_t)(const VV2 &, const VV2 &, const VD &, const VD &, const RansacOptions &, MDG *,
                                      std::vector<char> *);
typedef RansacStats (*ransac_focal_t)(const VV2 &, const VV2 &, const VD &, const VD &, const RansacOptions &, MDIP *,
                                      std::vector<char> *);
typedef RansacStats (*est_calib_t)(const VV2 &, const VV2 &, const VD &, const VD &, const Camera &, const Camera &,
                                   const RansacOptions &, const BundleOptions &, MDG *, std::vector<char> *);
typedef RansacStats (*est_focal_t)(const VV2 &, const VV2 &, const VD &, const VD &, const RansacOptions &,
                                   const BundleOptions &, MDIP *, std::vector<char> *);
typedef void (*draw_sample_t)(size_t, size_t, std::vector<size_t> *, unsigned long &);
// non-monodepth baselines of the same binary: 5-point, 7-point, 6-point shared focal
struct M33 { double m[9]; }; // Eigen::Matrix3d, column-major
struct alignas(32) ImagePair { CameraPose pose; Camera camera1, camera2; };
typedef std::vector<M33> VM33;
typedef int (*solver5E_t)(const VV3 &, const VV3 &, VM33 *);
typedef int (*solver5_t)(const VV3 &, const VV3 &, std::vector<CameraPose> *);
typedef int (*solver6_t)(const VV3 &, const VV3 &, std::vector<ImagePair> *);
typedef void (*motion_t)(const M33 &, const VV3 &, const VV3 &, std::vector<CameraPose> *);
typedef BundleStats (*refine_pose_t)(const VV2 &, const VV2 &, CameraPose *, const BundleOptions &, const VD &);
typedef BundleStats (*refine_F_t)(const VV2 &, const VV2 &, M33 *, const BundleOptions &, const VD &);
typedef BundleStats (*refine_ip_t)(const VV2 &, const VV2 &, ImagePair *, const BundleOptions &, const VD &);
typedef RansacStats (*ransac_pose_t)(const VV2 &, const VV2 &, const RansacOptions &, CameraPose *, std::vector<char> *);
typedef RansacStats (*ransac_F_t)(const VV2 &, const VV2 &, const RansacOptions &, M33 *, std::vector<char> *);
typedef RansacStats (*ransac_ip_t)(const VV2 &, const VV2 &, const RansacOptions &, ImagePair *, std::vector<char> *);
typedef RansacStats (*est_pose_t)(const VV2 &, const VV2 &, const Camera &, const Camera &, const RansacOptions &,
                                  const BundleOptions &, CameraPose *, std::vector<char> *);
typedef RansacStats (*est_F_t)(const VV2 &, const VV2 &, const RansacOptions &, const BundleOptions &, M33 *, std::vector<char> *);
typedef RansacStats (*est_ip_t)(const VV2 &, const VV2 &, const V2 &, const RansacOptions &, const BundleOptions &, ImagePair *,
                                std::vector<char> *);

void *H = nullptr;
solver3_t f_solver_calib;
solverf_t f_solver_shared, f_solver_varying;
p3p_t f_p3p;
msac_pose_t f_msac_pose;
msac_F_t f_msac_F;
inl_pose_t f_inl_pose;
inl_F_t f_inl_F;
cheir_t f_cheir;
refine_calib_t f_refine_calib;
refine_focal_t f_refine_shared, f_refine_varying;
ransac_calib_t f_ransac_calib;
ransac_focal_t f_ransac_shared, f_ransac_varying;
est_calib_t f_est_calib;
est_focal_t f_est_shared, f_est_varying;
draw_sample_t f_draw;
solver5E_t f_5pt_E, f_7pt;
solver5_t f_5pt;
solver6_t f_6pt;
motion_t f_motion;
refine_pose_t f_refine_relpose;
refine_F_t f_refine_F;
refine_ip_t f_refine_sf;
ransac_pose_t f_ransac_relpose;
ransac_F_t f_ransac_F;
ransac_ip_t f_ransac_sf;
est_pose_t f_est_relpose;
est_F_t f_est_F;
est_ip_t f_est_sf;

template <typename T> bool sym(T &f, const char *name) {
    f = (T)dlsym(H, name);
    if (!f) fprintf(stderr, "refshim: missing symbol %s\n", name);
    return f != nullptr;
}

VV2 mk2(const double *x, int n) {
    VV2 v(n);
    for (int i = 0; i < n; ++i) { v[i].v[0] = x[2 * i]; v[i].v[1] = x[2 * i + 1]; }
    return v;
}
VV3 mk3(const double *x, int n) {
    VV3 v(n);
    for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) v[i].v[k] = x[3 * i + k];
    return v;
}
VD mkd(const double *x, int n) { return VD(x, x + n); }

// flat geometry: q[4] t[3] scale shift1 shift2  (10)  ; image pair: + f1 f2 (12)
void geom_in(const double *g, MDG *m) {
    memcpy(m->pose.q, g, 4 * 8); memcpy(m->pose.t, g + 4, 3 * 8);
    m->scale = g[7]; m->shift1 = g[8]; m->shift2 = g[9];
}
void geom_out(const MDG &m, double *g) {
    memcpy(g, m.pose.q, 4 * 8); memcpy(g + 4, m.pose.t, 3 * 8);
    g[7] = m.scale; g[8] = m.shift1; g[9] = m.shift2;
}
void pair_in(const double *g, MDIP *p) {
    geom_in(g, &p->geometry);
    p->camera1.model_id = 0; p->camera1.width = 0; p->camera1.height = 0; p->camera1.params = {g[10], 0.0, 0.0};
    p->camera2.model_id = 0; p->camera2.width = 0; p->camera2.height = 0; p->camera2.params = {g[11], 0.0, 0.0};
}
void pair_out(const MDIP &p, double *g) {
    geom_out(p.geometry, g);
    g[10] = p.camera1.params.empty() ? 0.0 : p.camera1.params[0];
    g[11] = p.camera2.params.empty() ? 0.0 : p.camera2.params[0];
}
// RansacOptions::score_initial_model (+0x49) for the calls that follow: the model passed in is then scored first
// (ransac<> branch @0x22f2c8); set through ref_set_score_initial()
bool g_score_initial = false;
// flat ransac opt: max_it,min_it,dyn_mult,success_prob,max_reproj,max_epi,seed,estimate_shift,weight_sampson (9)
RansacOptions ropt_in(const double *o) {
    RansacOptions r;
    memset(&r, 0, sizeof r);
    r.max_iterations = (size_t)o[0]; r.min_iterations = (size_t)o[1];
    r.dyn_num_trials_mult = o[2]; r.success_prob = o[3]; r.max_reproj_error = o[4]; r.max_epipolar_error = o[5];
    r.seed = (unsigned long)o[6]; r.progressive_sampling = false; r.max_prosac_iterations = 100000;
    r.real_focal_check = false; r.score_initial_model = g_score_initial; r.monodepth_estimate_shift = o[7] != 0.0;
    r.monodepth_weight_sampson = (float)o[8];
    return r;
}
// flat bundle opt: max_it,loss_type,loss_scale,grad_tol,step_tol,lambda0,min_lambda,max_lambda (8)
BundleOptions bopt_in(const double *o) {
    BundleOptions b;
    memset(&b, 0, sizeof b);
    b.max_iterations = (size_t)o[0]; b.loss_type = (int)o[1]; b.loss_scale = o[2]; b.gradient_tol = o[3];
    b.step_tol = o[4]; b.initial_lambda = o[5]; b.min_lambda = o[6]; b.max_lambda = o[7]; b.verbose = false;
    return b;
}
void bstats_out(const BundleStats &s, double *o) {
    o[0] = (double)s.iterations; o[1] = s.initial_cost; o[2] = s.cost; o[3] = s.lambda;
    o[4] = (double)s.invalid_steps; o[5] = s.step_norm; o[6] = s.grad_norm;
}
void rstats_out(const RansacStats &s, double *o) {
    o[0] = (double)s.refinements; o[1] = (double)s.iterations; o[2] = (double)s.num_inliers;
    o[3] = s.inlier_ratio; o[4] = s.model_score;
}
Camera cam_in(const double *c) { // model_id, width, height, nparams, params...
    Camera cam;
    cam.model_id = (int)c[0]; cam.width = (int)c[1]; cam.height = (int)c[2];
    int np = (int)c[3];
    cam.params.assign(c + 4, c + 4 + np);
    return cam;
}
void mask_out(const std::vector<char> &m, unsigned char *out, int n) {
    for (int i = 0; i < n; ++i) out[i] = (i < (int)m.size()) ? (unsigned char)(m[i] != 0) : 0;
}

} // namespace

extern "C" {

int ref_init(const char *so_path) {
    if (H) return 0;
    H = dlopen(so_path, RTLD_LAZY | RTLD_GLOBAL);
    if (!H) { fprintf(stderr, "refshim: dlopen failed: %s\n", dlerror()); return -1; }
    bool ok = true;
    ok &= sym(f_solver_calib, "_ZN7poselib21relpose_monodepth_3ptERKSt6vectorIN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEESaIS3_EES7_RKS0_IdSaIdEESB_PS0_INS_24MonoDepthTwoViewGeometryESaISC_EE");
    ok &= sym(f_solver_shared, "_ZN7poselib34relpose_monodepth_3pt_shared_focalERKSt6vectorIN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEESaIS3_EES7_RKS0_IdSaIdEESB_PS0_INS_18MonoDepthImagePairESaISC_EE");
    ok &= sym(f_solver_varying, "_ZN7poselib35relpose_monodepth_3pt_varying_focalERKSt6vectorIN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEESaIS3_EES7_RKS0_IdSaIdEESB_PS0_INS_18MonoDepthImagePairESaISC_EE");
    ok &= sym(f_p3p, "_ZN7poselib3p3pERKSt6vectorIN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEESaIS3_EES7_PS0_INS_10CameraPoseESaIS8_EE");
    ok &= sym(f_msac_pose, "_ZN7poselib26compute_sampson_msac_scoreERKNS_10CameraPoseERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS6_EESA_dPm");
    ok &= sym(f_msac_F, "_ZN7poselib26compute_sampson_msac_scoreERKN5Eigen6MatrixIdLi3ELi3ELi0ELi3ELi3EEERKSt6vectorINS1_IdLi2ELi1ELi0ELi2ELi1EEESaIS6_EESA_dPm");
    ok &= sym(f_inl_pose, "_ZN7poselib11get_inliersERKNS_10CameraPoseERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS6_EESA_dPS3_IcSaIcEE");
    ok &= sym(f_inl_F, "_ZN7poselib11get_inliersERKN5Eigen6MatrixIdLi3ELi3ELi0ELi3ELi3EEERKSt6vectorINS1_IdLi2ELi1ELi0ELi2ELi1EEESaIS6_EESA_dPS5_IcSaIcEE");
    ok &= sym(f_cheir, "_ZN7poselib16check_cheiralityERKNS_10CameraPoseERKN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEES7_d");
    ok &= sym(f_refine_calib, "_ZN7poselib24refine_monodepth_relposeERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKS0_IdSaIdEESB_PNS_24MonoDepthTwoViewGeometryEddRKNS_13BundleOptionsEbSB_");
    ok &= sym(f_refine_shared, "_ZN7poselib37refine_monodepth_shared_focal_relposeERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKS0_IdSaIdEESB_PNS_18MonoDepthImagePairEddRKNS_13BundleOptionsESB_");
    ok &= sym(f_refine_varying, "_ZN7poselib38refine_monodepth_varying_focal_relposeERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKS0_IdSaIdEESB_PNS_18MonoDepthImagePairEddRKNS_13BundleOptionsESB_");
    ok &= sym(f_ransac_calib, "_ZN7poselib24ransac_monodepth_relposeERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKS0_IdSaIdEESB_RKNS_13RansacOptionsEPNS_24MonoDepthTwoViewGeometryEPS0_IcSaIcEE");
    ok &= sym(f_ransac_shared, "_ZN7poselib37ransac_shared_focal_monodepth_relposeERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKS0_IdSaIdEESB_RKNS_13RansacOptionsEPNS_18MonoDepthImagePairEPS0_IcSaIcEE");
    ok &= sym(f_ransac_varying, "_ZN7poselib38ransac_varying_focal_monodepth_relposeERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKS0_IdSaIdEESB_RKNS_13RansacOptionsEPNS_18MonoDepthImagePairEPS0_IcSaIcEE");
    ok &= sym(f_est_calib, "_ZN7poselib32estimate_monodepth_relative_poseERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKS0_IdSaIdEESB_RKNS_6CameraESE_RKNS_13RansacOptionsERKNS_13BundleOptionsEPNS_24MonoDepthTwoViewGeometryEPS0_IcSaIcEE");
    ok &= sym(f_est_shared, "_ZN7poselib45estimate_shared_focal_monodepth_relative_poseERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKS0_IdSaIdEESB_RKNS_13RansacOptionsERKNS_13BundleOptionsEPNS_18MonoDepthImagePairEPS0_IcSaIcEE");
    ok &= sym(f_est_varying, "_ZN7poselib46estimate_varying_focal_monodepth_relative_poseERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKS0_IdSaIdEESB_RKNS_13RansacOptionsERKNS_13BundleOptionsEPNS_18MonoDepthImagePairEPS0_IcSaIcEE");
    ok &= sym(f_draw, "_ZN7poselib11draw_sampleEmmPSt6vectorImSaImEERm");
    ok &= sym(f_5pt_E, "_ZN7poselib11relpose_5ptERKSt6vectorIN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEESaIS3_EES7_PS0_INS2_IdLi3ELi3ELi0ELi3ELi3EEESaIS8_EE");
    ok &= sym(f_5pt, "_ZN7poselib11relpose_5ptERKSt6vectorIN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEESaIS3_EES7_PS0_INS_10CameraPoseESaIS8_EE");
    ok &= sym(f_7pt, "_ZN7poselib11relpose_7ptERKSt6vectorIN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEESaIS3_EES7_PS0_INS2_IdLi3ELi3ELi0ELi3ELi3EEESaIS8_EE");
    ok &= sym(f_6pt, "_ZN7poselib24relpose_6pt_shared_focalERKSt6vectorIN5Eigen6MatrixIdLi3ELi1ELi0ELi3ELi1EEESaIS3_EES7_PS0_INS_9ImagePairESaIS8_EE");
    ok &= sym(f_motion, "_ZN7poselib21motion_from_essentialERKN5Eigen6MatrixIdLi3ELi3ELi0ELi3ELi3EEERKSt6vectorINS1_IdLi3ELi1ELi0ELi3ELi1EEESaIS6_EESA_PS5_INS_10CameraPoseESaISB_EE");
    ok &= sym(f_refine_relpose, "_ZN7poselib14refine_relposeERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_PNS_10CameraPoseERKNS_13BundleOptionsERKS0_IdSaIdEE");
    ok &= sym(f_refine_F, "_ZN7poselib18refine_fundamentalERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_PNS2_IdLi3ELi3ELi0ELi3ELi3EEERKNS_13BundleOptionsERKS0_IdSaIdEE");
    ok &= sym(f_refine_sf, "_ZN7poselib27refine_shared_focal_relposeERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_PNS_9ImagePairERKNS_13BundleOptionsERKS0_IdSaIdEE");
    ok &= sym(f_ransac_relpose, "_ZN7poselib14ransac_relposeERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKNS_13RansacOptionsEPNS_10CameraPoseEPS0_IcSaIcEE");
    ok &= sym(f_ransac_F, "_ZN7poselib18ransac_fundamentalERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKNS_13RansacOptionsEPNS2_IdLi3ELi3ELi0ELi3ELi3EEEPS0_IcSaIcEE");
    ok &= sym(f_ransac_sf, "_ZN7poselib27ransac_shared_focal_relposeERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKNS_13RansacOptionsEPNS_9ImagePairEPS0_IcSaIcEE");
    ok &= sym(f_est_relpose, "_ZN7poselib22estimate_relative_poseERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKNS_6CameraESA_RKNS_13RansacOptionsERKNS_13BundleOptionsEPNS_10CameraPoseEPS0_IcSaIcEE");
    ok &= sym(f_est_F, "_ZN7poselib20estimate_fundamentalERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKNS_13RansacOptionsERKNS_13BundleOptionsEPNS2_IdLi3ELi3ELi0ELi3ELi3EEEPS0_IcSaIcEE");
    ok &= sym(f_est_sf, "_ZN7poselib35estimate_shared_focal_relative_poseERKSt6vectorIN5Eigen6MatrixIdLi2ELi1ELi0ELi2ELi1EEESaIS3_EES7_RKS3_RKNS_13RansacOptionsERKNS_13BundleOptionsEPNS_9ImagePairEPS0_IcSaIcEE");
    return ok ? 0 : -2;
}

// ---- non-monodepth baselines.  Flat models: pose = q[4] t[3]; shared focal = pose + f (8); F = 9 doubles column-major.
void ref_draw_samples_k(unsigned long seed, size_t N, int k, int count, long long *out) {
    unsigned long state = seed;
    std::vector<size_t> s(k);
    for (int i = 0; i < count; ++i) {
        f_draw((size_t)k, N, &s, state);
        for (int j = 0; j < k; ++j) out[(size_t)k * i + j] = (long long)s[j];
    }
}
int ref_relpose_5pt_E(const double *x1h, const double *x2h, double *out /*10*9*/) {
    VM33 Es;
    int n = f_5pt_E(mk3(x1h, 5), mk3(x2h, 5), &Es);
    for (size_t i = 0; i < Es.size() && i < 10; ++i) memcpy(out + 9 * i, Es[i].m, 72);
    return n;
}
int ref_relpose_5pt(const double *x1h, const double *x2h, double *out /*40*7*/) {
    std::vector<CameraPose> poses;
    int n = f_5pt(mk3(x1h, 5), mk3(x2h, 5), &poses);
    for (size_t i = 0; i < poses.size() && i < 40; ++i) { memcpy(out + 7 * i, poses[i].q, 32); memcpy(out + 7 * i + 4, poses[i].t, 24); }
    return n;
}
int ref_motion_from_essential(const double *E_colmajor, const double *x1h, const double *x2h, int npts, double *out /*4*7*/) {
    M33 E; memcpy(E.m, E_colmajor, 72);
    std::vector<CameraPose> poses;
    f_motion(E, mk3(x1h, npts), mk3(x2h, npts), &poses);
    for (size_t i = 0; i < poses.size() && i < 4; ++i) { memcpy(out + 7 * i, poses[i].q, 32); memcpy(out + 7 * i + 4, poses[i].t, 24); }
    return (int)poses.size();
}
int ref_relpose_7pt(const double *x1h, const double *x2h, double *out /*3*9*/) {
    VM33 Fs;
    int n = f_7pt(mk3(x1h, 7), mk3(x2h, 7), &Fs);
    for (size_t i = 0; i < Fs.size() && i < 3; ++i) memcpy(out + 9 * i, Fs[i].m, 72);
    return n;
}
int ref_relpose_6pt(const double *x1h, const double *x2h, double *out /*60*8*/) {
    std::vector<ImagePair> ips;
    int n = f_6pt(mk3(x1h, 6), mk3(x2h, 6), &ips);
    for (size_t i = 0; i < ips.size() && i < 60; ++i) {
        memcpy(out + 8 * i, ips[i].pose.q, 32); memcpy(out + 8 * i + 4, ips[i].pose.t, 24);
        out[8 * i + 7] = ips[i].camera1.params.empty() ? 0.0 : ips[i].camera1.params[0];
    }
    return n;
}
namespace {
void ip_in(const double *g, ImagePair *p) {
    memcpy(p->pose.q, g, 32); memcpy(p->pose.t, g + 4, 24);
    p->camera1.model_id = 0; p->camera1.width = 0; p->camera1.height = 0; p->camera1.params = {g[7], 0.0, 0.0};
    p->camera2 = p->camera1;
}
void ip_out(const ImagePair &p, double *g) {
    memcpy(g, p.pose.q, 32); memcpy(g + 4, p.pose.t, 24);
    g[7] = p.camera1.params.empty() ? 0.0 : p.camera1.params[0];
    g[8] = p.camera2.params.empty() ? 0.0 : p.camera2.params[0];
}
}
// kind: 3 = relative pose (model 7), 4 = shared focal (model 9: pose, f1, f2), 5 = fundamental (model 9, column-major)
void ref_refine_classic(int kind, const double *x1, const double *x2, int n, double *model, const double *bopt8,
                        const double *weights, int nw, double *stats7) {
    BundleOptions b = bopt_in(bopt8);
    VD w = nw ? mkd(weights, nw) : VD();
    BundleStats s;
    if (kind == 3) {
        CameraPose p; memcpy(p.q, model, 32); memcpy(p.t, model + 4, 24);
        s = f_refine_relpose(mk2(x1, n), mk2(x2, n), &p, b, w);
        memcpy(model, p.q, 32); memcpy(model + 4, p.t, 24);
    } else if (kind == 4) {
        ImagePair ip; ip_in(model, &ip);
        s = f_refine_sf(mk2(x1, n), mk2(x2, n), &ip, b, w);
        ip_out(ip, model);
    } else {
        M33 F; memcpy(F.m, model, 72);
        s = f_refine_F(mk2(x1, n), mk2(x2, n), &F, b, w);
        memcpy(model, F.m, 72);
    }
    bstats_out(s, stats7);
}
void ref_ransac_classic(int kind, const double *x1, const double *x2, int n, const double *ropt9, double *model,
                        double *stats5, unsigned char *mask) {
    RansacOptions r = ropt_in(ropt9);
    std::vector<char> inl;
    RansacStats s;
    if (kind == 3) {
        CameraPose p; memcpy(p.q, model, 32); memcpy(p.t, model + 4, 24);
        s = f_ransac_relpose(mk2(x1, n), mk2(x2, n), r, &p, &inl);
        memcpy(model, p.q, 32); memcpy(model + 4, p.t, 24);
    } else if (kind == 4) {
        ImagePair ip; ip_in(model, &ip);
        s = f_ransac_sf(mk2(x1, n), mk2(x2, n), r, &ip, &inl);
        ip_out(ip, model);
    } else {
        M33 F; memcpy(F.m, model, 72);
        s = f_ransac_F(mk2(x1, n), mk2(x2, n), r, &F, &inl);
        memcpy(model, F.m, 72);
    }
    rstats_out(s, stats5);
    mask_out(inl, mask, n);
}
// cam1/cam2 (flat, as ref_estimate) for kind 3; pp[2] for kind 4
void ref_estimate_classic(int kind, const double *x1, const double *x2, int n, const double *cam1, const double *cam2,
                          const double *pp, const double *ropt9, const double *bopt8, double *model, double *stats5,
                          unsigned char *mask) {
    RansacOptions r = ropt_in(ropt9);
    BundleOptions b = bopt_in(bopt8);
    std::vector<char> inl;
    RansacStats s;
    if (kind == 3) {
        CameraPose p; memcpy(p.q, model, 32); memcpy(p.t, model + 4, 24);
        Camera c1 = cam_in(cam1), c2 = cam_in(cam2);
        s = f_est_relpose(mk2(x1, n), mk2(x2, n), c1, c2, r, b, &p, &inl);
        memcpy(model, p.q, 32); memcpy(model + 4, p.t, 24);
    } else if (kind == 4) {
        ImagePair ip; ip_in(model, &ip);
        V2 c; c.v[0] = pp[0]; c.v[1] = pp[1];
        s = f_est_sf(mk2(x1, n), mk2(x2, n), c, r, b, &ip, &inl);
        ip_out(ip, model);
    } else {
        M33 F; memcpy(F.m, model, 72);
        s = f_est_F(mk2(x1, n), mk2(x2, n), r, b, &F, &inl);
        memcpy(model, F.m, 72);
    }
    rstats_out(s, stats5);
    mask_out(inl, mask, n);
}

// count samples of size 3 drawn consecutively from rng state `seed`; out: count*3 indices
void ref_draw_samples(unsigned long seed, size_t N, int count, long long *out) {
    unsigned long state = seed;
    std::vector<size_t> s(3);
    for (int i = 0; i < count; ++i) {
        f_draw(3, N, &s, state);
        for (int k = 0; k < 3; ++k) out[3 * i + k] = (long long)s[k];
    }
}

int ref_p3p(const double *x, const double *X, double *out /*4*7*/) {
    VV3 xv = mk3(x, 3), Xv = mk3(X, 3);
    std::vector<CameraPose> poses;
    int n = f_p3p(xv, Xv, &poses);
    for (size_t i = 0; i < poses.size() && i < 4; ++i) { memcpy(out + 7 * i, poses[i].q, 32); memcpy(out + 7 * i + 4, poses[i].t, 24); }
    return n;
}

int ref_solver_calib(const double *x1h, const double *x2h, const double *d1, const double *d2, double *out /*4*10*/) {
    std::vector<MDG> sols;
    int n = f_solver_calib(mk3(x1h, 3), mk3(x2h, 3), mkd(d1, 3), mkd(d2, 3), &sols);
    for (size_t i = 0; i < sols.size() && i < 4; ++i) geom_out(sols[i], out + 10 * i);
    return n;
}
int ref_solver_shared(const double *x1h, const double *x2h, const double *d1, const double *d2, double *out /*4*12*/) {
    std::vector<MDIP> sols;
    int n = f_solver_shared(mk3(x1h, 3), mk3(x2h, 3), mkd(d1, 3), mkd(d2, 3), &sols);
    for (size_t i = 0; i < sols.size() && i < 4; ++i) pair_out(sols[i], out + 12 * i);
    return n;
}
int ref_solver_varying(const double *x1h, const double *x2h, const double *d1, const double *d2, double *out /*4*12*/) {
    std::vector<MDIP> sols;
    int n = f_solver_varying(mk3(x1h, 3), mk3(x2h, 3), mkd(d1, 3), mkd(d2, 3), &sols);
    for (size_t i = 0; i < sols.size() && i < 4; ++i) pair_out(sols[i], out + 12 * i);
    return n;
}

double ref_msac_pose(const double *pose7, const double *x1, const double *x2, int n, double sq_thr, long long *cnt) {
    CameraPose p; memcpy(p.q, pose7, 32); memcpy(p.t, pose7 + 4, 24);
    size_t c = 0;
    double s = f_msac_pose(p, mk2(x1, n), mk2(x2, n), sq_thr, &c);
    *cnt = (long long)c;
    return s;
}
double ref_msac_F(const double *F_colmajor, const double *x1, const double *x2, int n, double sq_thr, long long *cnt) {
    alignas(32) double F[9]; memcpy(F, F_colmajor, 72);
    size_t c = 0;
    double s = f_msac_F(F, mk2(x1, n), mk2(x2, n), sq_thr, &c);
    *cnt = (long long)c;
    return s;
}
int ref_inliers_pose(const double *pose7, const double *x1, const double *x2, int n, double sq_thr, unsigned char *mask) {
    CameraPose p; memcpy(p.q, pose7, 32); memcpy(p.t, pose7 + 4, 24);
    std::vector<char> m;
    int r = f_inl_pose(p, mk2(x1, n), mk2(x2, n), sq_thr, &m);
    mask_out(m, mask, n);
    return r;
}
int ref_inliers_F(const double *F_colmajor, const double *x1, const double *x2, int n, double sq_thr, unsigned char *mask) {
    alignas(32) double F[9]; memcpy(F, F_colmajor, 72);
    std::vector<char> m;
    int r = f_inl_F(F, mk2(x1, n), mk2(x2, n), sq_thr, &m);
    mask_out(m, mask, n);
    return r;
}
int ref_check_cheirality(const double *pose7, const double *x1, const double *x2, double min_depth) {
    CameraPose p; memcpy(p.q, pose7, 32); memcpy(p.t, pose7 + 4, 24);
    V3 a, b; memcpy(a.v, x1, 24); memcpy(b.v, x2, 24);
    return f_cheir(p, a, b, min_depth) ? 1 : 0;
}

void ref_refine_calib(const double *x1, const double *x2, const double *d1, const double *d2, int n, double *geom10,
                      double scale_reproj, double weight_sampson, const double *bopt8, int estimate_shift,
                      const double *weights, int nw, double *stats7) {
    MDG g; geom_in(geom10, &g);
    BundleOptions b = bopt_in(bopt8);
    BundleStats s = f_refine_calib(mk2(x1, n), mk2(x2, n), mkd(d1, n), mkd(d2, n), &g, scale_reproj, weight_sampson, b,
                                   estimate_shift != 0, nw ? mkd(weights, nw) : VD());
    geom_out(g, geom10);
    bstats_out(s, stats7);
}
void ref_refine_focal(int varying, const double *x1, const double *x2, const double *d1, const double *d2, int n,
                      double *pair12, double scale_reproj, double weight_sampson, const double *bopt8,
                      const double *weights, int nw, double *stats7) {
    MDIP p; pair_in(pair12, &p);
    BundleOptions b = bopt_in(bopt8);
    BundleStats s = (varying ? f_refine_varying : f_refine_shared)(mk2(x1, n), mk2(x2, n), mkd(d1, n), mkd(d2, n), &p,
                                                                  scale_reproj, weight_sampson, b,
                                                                  nw ? mkd(weights, nw) : VD());
    pair_out(p, pair12);
    bstats_out(s, stats7);
}

void ref_set_score_initial(int on) { g_score_initial = on != 0; }

// kind: 0 calibrated (normalised inputs), 1 shared, 2 varying. model: 10 (calib) or 12 doubles, in/out.
void ref_ransac(int kind, const double *x1, const double *x2, const double *d1, const double *d2, int n,
                const double *ropt9, double *model, double *stats5, unsigned char *mask) {
    RansacOptions r = ropt_in(ropt9);
    std::vector<char> inl;
    RansacStats s;
    if (kind == 0) {
        MDG g; geom_in(model, &g);
        s = f_ransac_calib(mk2(x1, n), mk2(x2, n), mkd(d1, n), mkd(d2, n), r, &g, &inl);
        geom_out(g, model);
    } else {
        MDIP p; pair_in(model, &p);
        s = (kind == 1 ? f_ransac_shared : f_ransac_varying)(mk2(x1, n), mk2(x2, n), mkd(d1, n), mkd(d2, n), r, &p, &inl);
        pair_out(p, model);
    }
    rstats_out(s, stats5);
    mask_out(inl, mask, n);
}

// kind as above; cam1/cam2 only for kind 0 (flat: model_id,width,height,nparams,params...)
void ref_estimate(int kind, const double *x1, const double *x2, const double *d1, const double *d2, int n,
                  const double *cam1, const double *cam2, const double *ropt9, const double *bopt8, double *model,
                  double *stats5, unsigned char *mask) {
    RansacOptions r = ropt_in(ropt9);
    BundleOptions b = bopt_in(bopt8);
    std::vector<char> inl;
    RansacStats s;
    if (kind == 0) {
        MDG g; geom_in(model, &g);
        Camera c1 = cam_in(cam1), c2 = cam_in(cam2);
        s = f_est_calib(mk2(x1, n), mk2(x2, n), mkd(d1, n), mkd(d2, n), c1, c2, r, b, &g, &inl);
        geom_out(g, model);
    } else {
        MDIP p; pair_in(model, &p);
        s = (kind == 1 ? f_est_shared : f_est_varying)(mk2(x1, n), mk2(x2, n), mkd(d1, n), mkd(d2, n), r, b, &p, &inl);
        pair_out(p, model);
    }
    rstats_out(s, stats5);
    mask_out(inl, mask, n);
}

} // extern "C"
