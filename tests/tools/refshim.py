"""ctypes front-end for oracle/_ref/librefshim.so (the reference's own PoseLib binary, dlopen()ed).

THIS CONTAINER ONLY — used by tests/tools/gen_golden.py and by ad-hoc parity probes while developing the
oracle.  No test (gpu or not), bench.py or mdrp_amd/ imports this module.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SHIM = os.path.join(_HERE, "..", "..", "oracle", "_ref", "librefshim.so")
_REF_SO = "/tmp/mdrp_ref_whl/poselib/_core.cpython-312-x86_64-linux-gnu.so"

_lib = None
_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def available():
    return os.path.exists(_SHIM) and os.path.exists(_REF_SO)


def lib():
    global _lib
    if _lib is None:
        # libpython must be global so the reference binary's Py* data symbols resolve
        C.CDLL("libpython3.10.so.1.0", mode=C.RTLD_GLOBAL)
        l = C.CDLL(_SHIM)
        l.ref_init.argtypes = [C.c_char_p]
        if l.ref_init(_REF_SO.encode()) != 0:
            raise RuntimeError("refshim init failed")
        l.ref_msac_pose.restype = C.c_double
        l.ref_msac_F.restype = C.c_double
        _lib = l
    return _lib


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def draw_samples(seed, N, count):
    out = np.zeros((count, 3), dtype=np.int64)
    lib().ref_draw_samples(C.c_ulong(seed), C.c_size_t(N), C.c_int(count), out.ctypes.data_as(C.c_void_p))
    return out


def p3p(x, X):
    x, X = f64(x), f64(X)
    out = np.zeros((4, 7))
    n = lib().ref_p3p(_p(x), _p(X), _p(out))
    return out[:n]


def _solver(fn, width, x1h, x2h, d1, d2):
    x1h, x2h, d1, d2 = f64(x1h), f64(x2h), f64(d1), f64(d2)
    out = np.zeros((4, width))
    n = fn(_p(x1h), _p(x2h), _p(d1), _p(d2), _p(out))
    return out[:n]


def solver_calib(x1h, x2h, d1, d2):
    return _solver(lib().ref_solver_calib, 10, x1h, x2h, d1, d2)


def solver_shared(x1h, x2h, d1, d2):
    return _solver(lib().ref_solver_shared, 12, x1h, x2h, d1, d2)


def solver_varying(x1h, x2h, d1, d2):
    return _solver(lib().ref_solver_varying, 12, x1h, x2h, d1, d2)


def msac_pose(pose7, x1, x2, sq_thr):
    pose7, x1, x2 = f64(pose7), f64(x1), f64(x2)
    cnt = C.c_longlong(0)
    s = lib().ref_msac_pose(_p(pose7), _p(x1), _p(x2), C.c_int(len(x1)), C.c_double(sq_thr), C.byref(cnt))
    return s, cnt.value


def msac_F(F, x1, x2, sq_thr):
    Fc = f64(np.asarray(F).T.reshape(-1))  # column-major
    x1, x2 = f64(x1), f64(x2)
    cnt = C.c_longlong(0)
    s = lib().ref_msac_F(_p(Fc), _p(x1), _p(x2), C.c_int(len(x1)), C.c_double(sq_thr), C.byref(cnt))
    return s, cnt.value


def inliers_pose(pose7, x1, x2, sq_thr):
    pose7, x1, x2 = f64(pose7), f64(x1), f64(x2)
    m = np.zeros(len(x1), dtype=np.uint8)
    lib().ref_inliers_pose(_p(pose7), _p(x1), _p(x2), C.c_int(len(x1)), C.c_double(sq_thr), m.ctypes.data_as(C.c_void_p))
    return m


def inliers_F(F, x1, x2, sq_thr):
    Fc = f64(np.asarray(F).T.reshape(-1))
    x1, x2 = f64(x1), f64(x2)
    m = np.zeros(len(x1), dtype=np.uint8)
    lib().ref_inliers_F(_p(Fc), _p(x1), _p(x2), C.c_int(len(x1)), C.c_double(sq_thr), m.ctypes.data_as(C.c_void_p))
    return m


def check_cheirality(pose7, x1, x2, min_depth=0.01):
    pose7, x1, x2 = f64(pose7), f64(x1), f64(x2)
    return bool(lib().ref_check_cheirality(_p(pose7), _p(x1), _p(x2), C.c_double(min_depth)))


def bopt(max_iterations=100, loss_type=3, loss_scale=1.0, gradient_tol=1e-8, step_tol=1e-8, initial_lambda=1e-3,
         min_lambda=1e-10, max_lambda=1e10):
    return f64([max_iterations, loss_type, loss_scale, gradient_tol, step_tol, initial_lambda, min_lambda, max_lambda])


def ropt(max_iterations=100000, min_iterations=1000, dyn_num_trials_mult=3.0, success_prob=0.9999,
         max_reproj_error=12.0, max_epipolar_error=1.0, seed=0, estimate_shift=False, weight_sampson=1.0):
    return f64([max_iterations, min_iterations, dyn_num_trials_mult, success_prob, max_reproj_error,
                max_epipolar_error, seed, 1.0 if estimate_shift else 0.0, weight_sampson])


def refine_calib(x1, x2, d1, d2, geom10, scale_reproj, weight_sampson, bo, estimate_shift, weights=None):
    x1, x2, d1, d2 = f64(x1), f64(x2), f64(d1), f64(d2)
    g = f64(geom10).copy()
    st = np.zeros(7)
    w = f64(weights) if weights is not None else np.zeros(1)
    nw = len(weights) if weights is not None else 0
    lib().ref_refine_calib(_p(x1), _p(x2), _p(d1), _p(d2), C.c_int(len(x1)), _p(g), C.c_double(scale_reproj),
                           C.c_double(weight_sampson), _p(bo), C.c_int(int(estimate_shift)), _p(w), C.c_int(nw), _p(st))
    return g, st


def refine_focal(varying, x1, x2, d1, d2, pair12, scale_reproj, weight_sampson, bo, weights=None):
    x1, x2, d1, d2 = f64(x1), f64(x2), f64(d1), f64(d2)
    g = f64(pair12).copy()
    st = np.zeros(7)
    w = f64(weights) if weights is not None else np.zeros(1)
    nw = len(weights) if weights is not None else 0
    lib().ref_refine_focal(C.c_int(int(varying)), _p(x1), _p(x2), _p(d1), _p(d2), C.c_int(len(x1)), _p(g),
                           C.c_double(scale_reproj), C.c_double(weight_sampson), _p(bo), _p(w), C.c_int(nw), _p(st))
    return g, st


def _init_model(kind):
    m = np.zeros(10 if kind == 0 else 12)
    m[0] = 1.0
    m[7] = 1.0
    if kind != 0:
        m[10] = m[11] = 1.0
    return m


def ransac(kind, x1, x2, d1, d2, ro):
    x1, x2, d1, d2 = f64(x1), f64(x2), f64(d1), f64(d2)
    model = _init_model(kind)
    st = np.zeros(5)
    mask = np.zeros(len(x1), dtype=np.uint8)
    lib().ref_ransac(C.c_int(kind), _p(x1), _p(x2), _p(d1), _p(d2), C.c_int(len(x1)), _p(ro), _p(model), _p(st),
                     mask.ctypes.data_as(C.c_void_p))
    return model, st, mask


def cam_flat(model_id, width, height, params):
    return f64([model_id, width, height, len(params)] + list(params))


def estimate(kind, x1, x2, d1, d2, ro, bo, cam1=None, cam2=None, initial=None, score_initial=False):
    """initial: model handed in (10 / 12 doubles; identity if None); score_initial: RansacOptions::score_initial_model"""
    x1, x2, d1, d2 = f64(x1), f64(x2), f64(d1), f64(d2)
    model = _init_model(kind) if initial is None else f64(initial).copy()
    lib().ref_set_score_initial(C.c_int(int(bool(score_initial))))
    st = np.zeros(5)
    mask = np.zeros(len(x1), dtype=np.uint8)
    c1 = cam1 if cam1 is not None else np.zeros(8)
    c2 = cam2 if cam2 is not None else np.zeros(8)
    lib().ref_estimate(C.c_int(kind), _p(x1), _p(x2), _p(d1), _p(d2), C.c_int(len(x1)), _p(c1), _p(c2), _p(ro), _p(bo),
                       _p(model), _p(st), mask.ctypes.data_as(C.c_void_p))
    lib().ref_set_score_initial(C.c_int(0))
    return model, st, mask


# ---- non-monodepth baselines of the same binary: kind 3 = 5-point relative pose (model: q, t),
# ---- 4 = 6-point shared focal (q, t, f1, f2), 5 = 7-point fundamental (F, row-major on this side)
def draw_samples_k(seed, N, k, count):
    out = np.zeros((count, k), dtype=np.int64)
    lib().ref_draw_samples_k(C.c_ulong(seed), C.c_size_t(N), C.c_int(k), C.c_int(count), out.ctypes.data_as(C.c_void_p))
    return out


def relpose_5pt_E(x1h, x2h):
    x1h, x2h = f64(x1h), f64(x2h)
    out = np.zeros((10, 9))
    n = lib().ref_relpose_5pt_E(_p(x1h), _p(x2h), _p(out))
    return out[:n].reshape(-1, 3, 3).transpose(0, 2, 1).copy()


def relpose_5pt(x1h, x2h):
    x1h, x2h = f64(x1h), f64(x2h)
    out = np.zeros((40, 7))
    n = lib().ref_relpose_5pt(_p(x1h), _p(x2h), _p(out))
    return out[:n]


def motion_from_essential(E, x1h, x2h):
    Ec = f64(np.asarray(E).T.reshape(-1))
    x1h, x2h = f64(x1h), f64(x2h)
    out = np.zeros((4, 7))
    n = lib().ref_motion_from_essential(_p(Ec), _p(x1h), _p(x2h), C.c_int(len(x1h)), _p(out))
    return out[:n]


def relpose_7pt(x1h, x2h):
    x1h, x2h = f64(x1h), f64(x2h)
    out = np.zeros((3, 9))
    n = lib().ref_relpose_7pt(_p(x1h), _p(x2h), _p(out))
    return out[:n].reshape(-1, 3, 3).transpose(0, 2, 1).copy()


def relpose_6pt(x1h, x2h):
    x1h, x2h = f64(x1h), f64(x2h)
    out = np.zeros((60, 8))
    n = lib().ref_relpose_6pt(_p(x1h), _p(x2h), _p(out))
    return out[:n]


def _classic_model(kind, initial=None):
    if initial is not None:
        m = f64(initial).copy()
        return f64(m.reshape(3, 3).T.reshape(-1)) if kind == 5 else m
    m = np.zeros(9)
    if kind != 5:
        m[0] = 1.0
        m[7] = m[8] = 1.0
    return m


def _classic_out(kind, m):
    return m.reshape(3, 3).T.reshape(-1).copy() if kind == 5 else (m[:7].copy() if kind == 3 else m.copy())


def refine_classic(kind, x1, x2, model, bo, weights=None):
    x1, x2 = f64(x1), f64(x2)
    m = np.zeros(9)
    mm = _classic_model(kind, model)
    m[:len(mm)] = mm
    st = np.zeros(7)
    w = f64(weights) if weights is not None else np.zeros(1)
    nw = len(weights) if weights is not None else 0
    lib().ref_refine_classic(C.c_int(kind), _p(x1), _p(x2), C.c_int(len(x1)), _p(m), _p(bo), _p(w), C.c_int(nw), _p(st))
    return _classic_out(kind, m), st


def ransac_classic(kind, x1, x2, ro):
    x1, x2 = f64(x1), f64(x2)
    m = _classic_model(kind)
    st = np.zeros(5)
    mask = np.zeros(len(x1), dtype=np.uint8)
    lib().ref_ransac_classic(C.c_int(kind), _p(x1), _p(x2), C.c_int(len(x1)), _p(ro), _p(m), _p(st), mask.ctypes.data_as(C.c_void_p))
    return _classic_out(kind, m), st, mask


def estimate_classic(kind, x1, x2, ro, bo, cam1=None, cam2=None, pp=(0.0, 0.0), score_initial=False, initial=None):
    x1, x2 = f64(x1), f64(x2)
    m = np.zeros(9)
    mm = _classic_model(kind, initial)
    m[:len(mm)] = mm
    lib().ref_set_score_initial(C.c_int(int(bool(score_initial))))
    st = np.zeros(5)
    mask = np.zeros(len(x1), dtype=np.uint8)
    c1 = cam1 if cam1 is not None else np.zeros(8)
    c2 = cam2 if cam2 is not None else np.zeros(8)
    ppv = f64(pp)
    lib().ref_estimate_classic(C.c_int(kind), _p(x1), _p(x2), C.c_int(len(x1)), _p(c1), _p(c2), _p(ppv), _p(ro), _p(bo), _p(m),
                               _p(st), mask.ctypes.data_as(C.c_void_p))
    lib().ref_set_score_initial(C.c_int(0))
    return _classic_out(kind, m), st, mask
