"""ctypes loader for oracle/libmdrp_oracle.so — the CPU restatement of the reference algorithm.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
(as the checker / reported baseline).  The product package mdrp_amd never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmdrp_oracle.so")
_lib = None
_dp = C.POINTER(C.c_double)

MODEL_W = 12  # q[4] t[3] scale shift1 shift2 f1 f2
CALIB, SHARED, VARYING = 0, 1, 2


class RansacOpt(C.Structure):
    _fields_ = [("max_iterations", C.c_uint64), ("min_iterations", C.c_uint64), ("dyn_num_trials_mult", C.c_double),
                ("success_prob", C.c_double), ("max_reproj_error", C.c_double), ("max_epipolar_error", C.c_double),
                ("seed", C.c_uint64), ("estimate_shift", C.c_int), ("weight_sampson", C.c_double), ("score_initial_model", C.c_int)]


class BundleOpt(C.Structure):
    _fields_ = [("max_iterations", C.c_uint64), ("loss_type", C.c_int), ("loss_scale", C.c_double),
                ("gradient_tol", C.c_double), ("step_tol", C.c_double), ("initial_lambda", C.c_double),
                ("min_lambda", C.c_double), ("max_lambda", C.c_double)]


class RansacStats(C.Structure):
    _fields_ = [("refinements", C.c_uint64), ("iterations", C.c_uint64), ("num_inliers", C.c_uint64),
                ("inlier_ratio", C.c_double), ("model_score", C.c_double)]


class BundleStats(C.Structure):
    _fields_ = [("iterations", C.c_uint64), ("initial_cost", C.c_double), ("cost", C.c_double), ("lambda_", C.c_double),
                ("invalid_steps", C.c_uint64), ("step_norm", C.c_double), ("grad_norm", C.c_double)]


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(_SO)
        l.orc_msac_pose.restype = C.c_double
        l.orc_msac_F.restype = C.c_double
        for name, rt in (("orc_refine", BundleStats), ("orc_ransac", RansacStats), ("orc_estimate", RansacStats),
                         ("orc_refine_classic", BundleStats), ("orc_ransac_classic", RansacStats), ("orc_estimate_classic", RansacStats)):
            if hasattr(l, name):
                getattr(l, name).restype = rt
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ransac_opt(max_iterations=100000, min_iterations=1000, dyn_num_trials_mult=3.0, success_prob=0.9999,
               max_reproj_error=12.0, max_epipolar_error=1.0, seed=0, estimate_shift=False, weight_sampson=1.0, score_initial_model=False):
    return RansacOpt(int(max_iterations), int(min_iterations), dyn_num_trials_mult, success_prob, max_reproj_error,
                     max_epipolar_error, int(seed), int(bool(estimate_shift)), float(np.float32(weight_sampson)),
                     int(bool(score_initial_model)))


def bundle_opt(max_iterations=100, loss_type=3, loss_scale=1.0, gradient_tol=1e-8, step_tol=1e-8,
               initial_lambda=1e-3, min_lambda=1e-10, max_lambda=1e10):
    return BundleOpt(int(max_iterations), int(loss_type), loss_scale, gradient_tol, step_tol, initial_lambda,
                     min_lambda, max_lambda)


def new_model():
    m = np.zeros(MODEL_W)
    m[0] = 1.0
    m[7] = 1.0
    m[10] = m[11] = 1.0
    return m


def draw_samples(seed, n, count):
    st = C.c_uint64(seed)
    out = np.zeros((count, 3), dtype=np.uint64)
    buf = (C.c_uint64 * 3)()
    l = lib()
    for i in range(count):
        l.orc_draw_sample(C.c_uint64(n), C.byref(st), buf)
        out[i] = buf[:]
    return out.astype(np.int64)


def _solver(fn, *arrs):
    arrs = [f64(a) for a in arrs]
    out = np.zeros((4, MODEL_W))
    n = fn(*[_p(a) for a in arrs], _p(out))
    return out[:n]


def p3p(x, X):
    return _solver(lib().orc_p3p, x, X)


def solver_calib_shift(x1h, x2h, d1, d2):
    return _solver(lib().orc_solver_calib_shift, x1h, x2h, d1, d2)


def solver_calib_p3p(x1h, x2h, d1, d2):
    return _solver(lib().orc_solver_calib_p3p, x1h, x2h, d1, d2)


def solver_shared(x1h, x2h, d1, d2):
    return _solver(lib().orc_solver_shared, x1h, x2h, d1, d2)


def solver_varying(x1h, x2h, d1, d2):
    return _solver(lib().orc_solver_varying, x1h, x2h, d1, d2)


def quat_to_rotmat(q):
    R = np.zeros(9)
    lib().orc_quat_to_rotmat(_p(f64(q)), _p(R))
    return R.reshape(3, 3)


def essential(model):
    E = np.zeros(9)
    lib().orc_essential(_p(f64(model)), _p(E))
    return E.reshape(3, 3)


def fundamental(model):
    F = np.zeros(9)
    lib().orc_fundamental(_p(f64(model)), _p(F))
    return F.reshape(3, 3)


def msac_pose(model, x1, x2, sq_thr):
    model, x1, x2 = f64(model), f64(x1), f64(x2)
    cnt = C.c_uint64(0)
    s = lib().orc_msac_pose(_p(model), _p(x1), _p(x2), C.c_int(len(x1)), C.c_double(sq_thr), C.byref(cnt))
    return s, cnt.value


def msac_F(F, x1, x2, sq_thr):
    F, x1, x2 = f64(np.asarray(F).reshape(-1)), f64(x1), f64(x2)
    cnt = C.c_uint64(0)
    s = lib().orc_msac_F(_p(F), _p(x1), _p(x2), C.c_int(len(x1)), C.c_double(sq_thr), C.byref(cnt))
    return s, cnt.value


def inliers_pose(model, x1, x2, sq_thr):
    model, x1, x2 = f64(model), f64(x1), f64(x2)
    m = np.zeros(len(x1), dtype=np.uint8)
    lib().orc_inliers_pose(_p(model), _p(x1), _p(x2), C.c_int(len(x1)), C.c_double(sq_thr), m.ctypes.data_as(C.c_void_p))
    return m


def inliers_F(F, x1, x2, sq_thr):
    F, x1, x2 = f64(np.asarray(F).reshape(-1)), f64(x1), f64(x2)
    m = np.zeros(len(x1), dtype=np.uint8)
    lib().orc_inliers_F(_p(F), _p(x1), _p(x2), C.c_int(len(x1)), C.c_double(sq_thr), m.ctypes.data_as(C.c_void_p))
    return m


def refine(kind, x1, x2, d1, d2, model, scale_reproj, weight_sampson, bopt, estimate_shift=False, weights=None):
    x1, x2, d1, d2 = f64(x1), f64(x2), f64(d1), f64(d2)
    m = f64(model).copy()
    w = f64(weights) if weights is not None else None
    st = lib().orc_refine(C.c_int(kind), _p(x1), _p(x2), _p(d1), _p(d2), C.c_int(len(x1)), _p(m),
                          C.c_double(scale_reproj), C.c_double(weight_sampson), C.byref(bopt),
                          C.c_int(int(estimate_shift)), _p(w) if w is not None else None)
    return m, st


def ransac(kind, x1, x2, d1, d2, ropt, initial=None):
    x1, x2, d1, d2 = f64(x1), f64(x2), f64(d1), f64(d2)
    m = new_model() if initial is None else f64(initial).copy()
    mask = np.zeros(len(x1), dtype=np.uint8)
    st = lib().orc_ransac(C.c_int(kind), _p(x1), _p(x2), _p(d1), _p(d2), C.c_int(len(x1)), C.byref(ropt), _p(m),
                          mask.ctypes.data_as(C.c_void_p))
    return m, st, mask


def cam_flat(model_id, params):
    return f64([model_id, len(params)] + list(params))


def estimate(kind, x1, x2, d1, d2, ropt, bopt, cam1=None, cam2=None, initial=None):
    """initial: 12-wide model handed in (its pose is reset by the reference; scale / shifts survive when nothing is found)"""
    x1, x2, d1, d2 = f64(x1), f64(x2), f64(d1), f64(d2)
    m = new_model() if initial is None else f64(initial).copy()
    mask = np.zeros(len(x1), dtype=np.uint8)
    c1 = f64(cam1) if cam1 is not None else np.zeros(8)
    c2 = f64(cam2) if cam2 is not None else np.zeros(8)
    st = lib().orc_estimate(C.c_int(kind), _p(x1), _p(x2), _p(d1), _p(d2), C.c_int(len(x1)), _p(c1), _p(c2),
                            C.byref(ropt), C.byref(bopt), _p(m), mask.ctypes.data_as(C.c_void_p))
    return m, st, mask


# ---- non-monodepth baselines (orc_classic.c).  kind 3: 5-point relative pose — model = q, t of the 12-wide blob;
# ---- kind 5: 7-point fundamental — model = F row-major in the blob's first nine doubles.
RELPOSE, FUNDAMENTAL = 3, 5


def relpose_5pt(x1h, x2h):
    x1h, x2h = f64(x1h), f64(x2h)
    out = np.zeros((40, MODEL_W))
    n = lib().orc_relpose_5pt(_p(x1h), _p(x2h), _p(out))
    return out[:n]


def relpose_5pt_E(x1h, x2h):
    x1h, x2h = f64(x1h), f64(x2h)
    out = np.zeros((10, 9))
    n = lib().orc_relpose_5pt_E(_p(x1h), _p(x2h), _p(out))
    return out[:n].reshape(-1, 3, 3)


def relpose_7pt(x1h, x2h):
    x1h, x2h = f64(x1h), f64(x2h)
    out = np.zeros((3, 9))
    n = lib().orc_relpose_7pt(_p(x1h), _p(x2h), _p(out))
    return out[:n].reshape(-1, 3, 3)


def relpose_6pt(x1h, x2h):
    """6-point shared-focal solver: rows of the 12-wide model blob (q, t, ..., f1 = f2 = f), by ascending focal length"""
    x1h, x2h = f64(x1h), f64(x2h)
    out = np.zeros((60, MODEL_W))
    n = lib().orc_relpose_6pt(_p(x1h), _p(x2h), _p(out))
    return out[:n]


def eigenvalues(a):
    a = f64(a).copy()
    n = a.shape[0]
    wr, wi = np.zeros(n), np.zeros(n)
    rc = lib().orc_eigenvalues(_p(a), C.c_int(n), _p(wr), _p(wi))
    return rc, wr + 1j * wi


def classic_blob(kind, model=None):
    m = np.zeros(MODEL_W)
    if model is None:
        if kind == FUNDAMENTAL:
            m[[0, 4, 8]] = 1.0
        else:
            m[0] = 1.0; m[7] = 1.0; m[10] = m[11] = 1.0
        return m
    model = f64(model).reshape(-1)
    m[:len(model)] = model
    if kind != FUNDAMENTAL and len(model) <= 7:
        m[7] = 1.0; m[10] = m[11] = 1.0
    return m


def refine_classic(kind, x1, x2, model, bopt, weights=None):
    x1, x2 = f64(x1), f64(x2)
    m = classic_blob(kind, model)
    w = f64(weights) if weights is not None else None
    st = lib().orc_refine_classic(C.c_int(kind), _p(x1), _p(x2), C.c_int(len(x1)), _p(m), C.byref(bopt), _p(w) if w is not None else None)
    return m, st


def ransac_classic(kind, x1, x2, ropt):
    x1, x2 = f64(x1), f64(x2)
    m = classic_blob(kind)
    mask = np.zeros(len(x1), dtype=np.uint8)
    st = lib().orc_ransac_classic(C.c_int(kind), _p(x1), _p(x2), C.c_int(len(x1)), C.byref(ropt), _p(m), mask.ctypes.data_as(C.c_void_p))
    return m, st, mask


def estimate_classic(kind, x1, x2, ropt, bopt, cam1=None, cam2=None, pp=(0.0, 0.0)):
    x1, x2 = f64(x1), f64(x2)
    m = classic_blob(kind)
    mask = np.zeros(len(x1), dtype=np.uint8)
    c1 = f64(cam1) if cam1 is not None else np.zeros(8)
    c2 = f64(cam2) if cam2 is not None else np.zeros(8)
    ppv = f64(pp)
    st = lib().orc_estimate_classic(C.c_int(kind), _p(x1), _p(x2), C.c_int(len(x1)), _p(c1), _p(c2), _p(ppv), C.byref(ropt), C.byref(bopt),
                                    _p(m), mask.ctypes.data_as(C.c_void_p))
    return m, st, mask
