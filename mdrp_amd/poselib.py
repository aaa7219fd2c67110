"""Drop-in for the monodepth part of the reference's `poselib` module, backed by the HIP kernels.

    import mdrp_amd.poselib as poselib
    geometry, info = poselib.estimate_monodepth_relative_pose(x1, x2, d1, d2, cam1, cam2, ransac_opt, bundle_opt)

Signatures, option dictionaries, result classes and the `info` dictionary follow the reference binding
(wheel poselib/_core.pyi:446-501, 134-204; callers /root/reference/make_pair.py:111, make_video.py:284,
README.md:86-96).  Each single-pair call is a batch of one; `*_batch` variants take B pairs at once, which is how
the GPU is meant to be fed (the reference parallelises over pairs with a process pool, eval.py:355-359).
The fork names used by the paper scripts (eval.py:153, eval_shared_f.py:177, eval_varying_f.py:168) are provided as
adapters at the bottom.
"""
import numpy as np

from . import _capi, pipeline

__version__ = "2.0.5+mdrp_amd"

CAMERA_MODELS = {"SIMPLE_PINHOLE": 0, "PINHOLE": 1}
CAMERA_MODEL_NAMES = {v: k for k, v in CAMERA_MODELS.items()}


def _quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


class CameraPose:
    """q = (w,x,y,z), x_cam = R X + t  (_core.pyi:134-156)"""

    def __init__(self, q=None, t=None):
        self.q = np.array([1.0, 0.0, 0.0, 0.0]) if q is None else np.asarray(q, dtype=np.float64).reshape(4)
        self.t = np.zeros(3) if t is None else np.asarray(t, dtype=np.float64).reshape(3)

    @property
    def R(self):
        return _quat_to_R(self.q)

    @property
    def Rt(self):
        return np.hstack([self.R, self.t.reshape(3, 1)])

    def center(self):
        return -self.R.T @ self.t

    def __repr__(self):
        return f"[q: {self.q}, t: {self.t}]"


class Camera:
    """COLMAP-style pinhole camera (_core.pyi:76-132).  Only the models the reference callers use."""

    def __init__(self, model="SIMPLE_PINHOLE", params=None, width=-1, height=-1):
        if isinstance(model, str):
            if model not in CAMERA_MODELS:
                raise NotImplementedError(f"camera model {model!r}: only SIMPLE_PINHOLE and PINHOLE are on the monodepth path")
            model = CAMERA_MODELS[model]
        self.model_id = int(model)
        self.params = [1.0, 0.0, 0.0] if params is None else [float(p) for p in params]
        self.width, self.height = int(width), int(height)

    @classmethod
    def from_any(cls, c):
        if isinstance(c, Camera):
            return c
        if isinstance(c, dict):
            return cls(c["model"], c["params"], c.get("width", -1), c.get("height", -1))
        raise TypeError("camera must be a Camera or a dict {'model','width','height','params'}")

    def model_name(self):
        return CAMERA_MODEL_NAMES[self.model_id]

    def focal_x(self):
        return self.params[0]

    def focal_y(self):
        return self.params[1] if self.model_id == 1 else self.params[0]

    def focal(self):
        return 0.5 * (self.focal_x() + self.focal_y())

    def principal_point(self):
        return np.array(self.params[2:4] if self.model_id == 1 else self.params[1:3])

    def unproject(self, x):
        x = np.asarray(x, dtype=np.float64)
        return (x - self.principal_point()) / np.array([self.focal_x(), self.focal_y()])

    def project(self, x):
        x = np.asarray(x, dtype=np.float64)
        return x * np.array([self.focal_x(), self.focal_y()]) + self.principal_point()

    def _record(self):
        r = np.zeros((), dtype=_capi.CAMERA_DTYPE)
        r["model_id"] = self.model_id
        p = np.zeros(4)
        p[: len(self.params)] = self.params
        r["params"] = p
        return r

    def __repr__(self):
        return f"Camera({self.model_name()}, params={self.params}, {self.width}x{self.height})"


class MonoDepthTwoViewGeometry:
    """R (d1+shift1) K1^-1 x1 + t = scale (d2+shift2) K2^-1 x2  (_core.pyi:178-204)"""

    def __init__(self, pose=None, scale=1.0, shift1=0.0, shift2=0.0):
        self.pose = pose if pose is not None else CameraPose()
        self.scale, self.shift1, self.shift2 = float(scale), float(shift1), float(shift2)

    def __repr__(self):
        return f"[pose: {self.pose}, scale: {self.scale}, shift1: {self.shift1}, shift2: {self.shift2}]"


class MonoDepthImagePair:
    """(_core.pyi:171-176); `.pose` is kept for the older wheel's naming used by eval_shared_f.py:84-96"""

    def __init__(self, geometry=None, camera1=None, camera2=None):
        self.geometry = geometry if geometry is not None else MonoDepthTwoViewGeometry()
        self.camera1 = camera1 if camera1 is not None else Camera()
        self.camera2 = camera2 if camera2 is not None else Camera()

    @property
    def pose(self):
        return self.geometry.pose

    def __repr__(self):
        return f"[geometry: {self.geometry}, camera1: {self.camera1}, camera2: {self.camera2}]"


def _geometry_from_model(m):
    return MonoDepthTwoViewGeometry(CameraPose(m["q"].copy(), m["t"].copy()), m["scale"], m["shift1"], m["shift2"])


def _pair_from_model(m):
    return MonoDepthImagePair(_geometry_from_model(m), Camera("SIMPLE_PINHOLE", [float(m["f1"]), 0.0, 0.0]),
                              Camera("SIMPLE_PINHOLE", [float(m["f2"]), 0.0, 0.0]))


def _info(res, mask_row, n):
    return {"refinements": int(res["refinements"]), "iterations": int(res["iterations"]), "num_inliers": int(res["num_inliers"]),
            "inlier_ratio": float(res["inlier_ratio"]), "model_score": float(res["model_score"]),
            "inliers": np.asarray(mask_row[:n]).astype(bool).tolist()}  # list[bool] like the reference, built at C speed


def _as_points(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    if p.ndim != 2 or p.shape[1] != 2:
        raise ValueError("points must have shape (N, 2)")
    return p


def _with_initial(initial, ransac_opt):
    """The reference binding sets RansacOptions::score_initial_model whenever an initial pose / image pair is passed
    (_core.pyi:455; upstream pybind wrapper).  What the reference does with it, pinned in tests/golden/initial.npz from the
    binary: the POSE handed in is never read — ransac_*_relpose reset it to the identity — and the reset model (no inliers,
    score N eps^2) is scored and LO-refined first, to no effect except `refinements` + 1 and the records it sets."""
    ro = dict(ransac_opt or {})
    if initial is not None:
        ro["score_initial_model"] = True
    return ro


def _initial_fallback(initial, geometry):
    """When RANSAC adopts nothing the reference returns the caller's model with its pose reset: scale and shifts of the
    initial geometry survive (cameras of an initial image pair do not: they come back as the normalisation focal)."""
    if initial is None:
        return geometry
    g0 = getattr(initial, "geometry", initial)
    p = geometry.pose
    if tuple(p.q) == (1.0, 0.0, 0.0, 0.0) and not p.t.any() and geometry.scale == 1.0:
        geometry.scale, geometry.shift1, geometry.shift2 = float(g0.scale), float(g0.shift1), float(g0.shift2)
    return geometry


def _camera_records(c, B):
    """[B] mdrp_camera records: one Camera | dict for all pairs (one record, repeated — no per-pair Python objects), a list of B, or a
    ready CAMERA_DTYPE array"""
    if isinstance(c, np.ndarray) and c.dtype == _capi.CAMERA_DTYPE:
        if len(c) != B:
            raise ValueError(f"expected {B} camera records, got {len(c)}")
        return np.ascontiguousarray(c)
    if not isinstance(c, (list, tuple)):
        return np.repeat(np.asarray(Camera.from_any(c)._record()).reshape(1), B)
    if len(c) != B:
        raise ValueError(f"expected {B} cameras, got {len(c)}")
    return np.array([Camera.from_any(v)._record() for v in c], dtype=_capi.CAMERA_DTYPE)


# ------------------------------------------------------------------------------------------------ batch API
def _stack(points1, points2, depth1, depth2):
    """list of ragged pairs or already-stacked arrays -> padded (B,N,2),(B,N,2),(B,N),(B,N), n_per_pair"""
    if isinstance(points1, np.ndarray) and points1.ndim == 3:
        B, N = points1.shape[:2]
        return (np.ascontiguousarray(points1, np.float64), np.ascontiguousarray(points2, np.float64),
                np.ascontiguousarray(depth1, np.float64), np.ascontiguousarray(depth2, np.float64), np.full(B, N, np.int32))
    B = len(points1)
    ns = np.array([len(p) for p in points1], dtype=np.int32)
    N = int(ns.max()) if B else 0
    x1 = np.zeros((B, N, 2)); x2 = np.zeros((B, N, 2)); d1 = np.ones((B, N)); d2 = np.ones((B, N))
    for i in range(B):
        n = ns[i]
        x1[i, :n] = _as_points(points1[i]); x2[i, :n] = _as_points(points2[i])
        d1[i, :n] = np.asarray(depth1[i], np.float64).reshape(-1); d2[i, :n] = np.asarray(depth2[i], np.float64).reshape(-1)
    return x1, x2, d1, d2, ns


def estimate_monodepth_relative_pose_batch(points2D_1, points2D_2, depth_1, depth_2, cameras1, cameras2, ransac_opt=None,
                                           bundle_opt=None, device=0, as_arrays=False):
    """B calibrated pairs at once.  cameras1/2: one Camera|dict for all pairs, or a list of B.  Returns
    (list[MonoDepthTwoViewGeometry], list[info dict]) — or, with as_arrays=True, (records, inlier masks, n_per_pair) as numpy arrays
    (_capi.RESULT_DTYPE; (B, N) uint8): building B Python objects and B lists of N bools costs more than the estimate itself beyond
    a few thousand pairs.  A host batch is ONE call whatever its size: the C side copies the correspondences in 256-pair slices on a copy stream
    beside the first kernels of the slices before them (MDRP_PIPELINE_MIN=<pairs> brings back the chunked two-in-flight path of rounds 4-5,
    mdrp_amd.pipeline; results identical to sequential chunk calls)."""
    x1, x2, d1, d2, ns = _stack(points2D_1, points2D_2, depth_1, depth_2)
    B = len(ns)

    def cams(c):
        return _camera_records(c, B)

    res, mask = pipeline.estimate_host(_capi.CALIB, x1, x2, d1, d2, _capi.ransac_opt_from_dict(ransac_opt),
                                       _capi.bundle_opt_from_dict(bundle_opt), ns, cams(cameras1), cams(cameras2), device)
    if as_arrays:
        return res, mask, ns
    return [_geometry_from_model(r["model"]) for r in res], [_info(res[i], mask[i], ns[i]) for i in range(B)]


def _focal_batch(kind, points2D_1, points2D_2, depth_1, depth_2, ransac_opt, bundle_opt, device, as_arrays=False):
    x1, x2, d1, d2, ns = _stack(points2D_1, points2D_2, depth_1, depth_2)
    res, mask = pipeline.estimate_host(kind, x1, x2, d1, d2, _capi.ransac_opt_from_dict(ransac_opt), _capi.bundle_opt_from_dict(bundle_opt), ns, None, None, device)
    if as_arrays:
        return res, mask, ns
    return [_pair_from_model(r["model"]) for r in res], [_info(res[i], mask[i], ns[i]) for i in range(len(ns))]


def estimate_monodepth_shared_focal_relative_pose_batch(points2D_1, points2D_2, depth_1, depth_2, ransac_opt=None,
                                                        bundle_opt=None, device=0, as_arrays=False):
    return _focal_batch(_capi.SHARED_FOCAL, points2D_1, points2D_2, depth_1, depth_2, ransac_opt, bundle_opt, device, as_arrays)


def estimate_monodepth_varying_focal_relative_pose_batch(points2D_1, points2D_2, depth_1, depth_2, ransac_opt=None,
                                                         bundle_opt=None, device=0, as_arrays=False):
    return _focal_batch(_capi.VARYING_FOCAL, points2D_1, points2D_2, depth_1, depth_2, ransac_opt, bundle_opt, device, as_arrays)


# ------------------------------------------------------------------------------------------------ reference signatures
def estimate_monodepth_relative_pose(points2D_1, points2D_2, depth_1, depth_2, camera1, camera2, ransac_opt={},
                                     bundle_opt={}, initial_pose=None):
    """Pose estimation using depth estimates with non-linear refinement (_core.pyi:446-475)."""
    g, i = estimate_monodepth_relative_pose_batch([_as_points(points2D_1)], [_as_points(points2D_2)], [depth_1], [depth_2],
                                                  camera1, camera2, _with_initial(initial_pose, ransac_opt), bundle_opt)
    return _initial_fallback(initial_pose, g[0]), i[0]


def estimate_monodepth_shared_focal_relative_pose(points2D_1, points2D_2, depth_1, depth_2, ransac_opt={}, bundle_opt={},
                                                  initial_image_pair=None):
    """Unknown equal focal lengths; points principal-point-centred (_core.pyi:477-488, README.md:88-90)."""
    p, i = estimate_monodepth_shared_focal_relative_pose_batch([_as_points(points2D_1)], [_as_points(points2D_2)], [depth_1],
                                                               [depth_2], _with_initial(initial_image_pair, ransac_opt), bundle_opt)
    _initial_fallback(initial_image_pair, p[0].geometry)
    return p[0], i[0]


def estimate_monodepth_varying_focal_relative_pose(points2D_1, points2D_2, depth_1, depth_2, ransac_opt={}, bundle_opt={},
                                                   initial_image_pair=None):
    """Two unknown focal lengths (_core.pyi:490-501, README.md:94-96).  `monodepth_estimate_shift` is ignored here
    exactly like in the reference."""
    p, i = estimate_monodepth_varying_focal_relative_pose_batch([_as_points(points2D_1)], [_as_points(points2D_2)], [depth_1],
                                                                [depth_2], _with_initial(initial_image_pair, ransac_opt), bundle_opt)
    _initial_fallback(initial_image_pair, p[0].geometry)
    return p[0], i[0]


# ------------------------------------------------------------------------------------------------ the older wheel's names
# demo/poselib_old-2.0.5-cp312-*.whl (poselib/_core.pyi:171-199, 441-497) ships the same three estimators as estimate_monodepth_pose,
# estimate_monodepth_shared_focal_pose and estimate_monodepth_varying_focal_pose; its result type is MonoDepthCameraPose — a CameraPose
# that carries scale, shift_1 and shift_2 itself — and its MonoDepthImagePair holds that as `.pose`.
class MonoDepthCameraPose(CameraPose):
    def __init__(self, q=None, t=None, scale=1.0, shift_1=0.0, shift_2=0.0):
        if isinstance(q, CameraPose):
            q, t = q.q, q.t
        super().__init__(q, t)
        self.scale, self.shift_1, self.shift_2 = float(scale), float(shift_1), float(shift_2)

    def __repr__(self):
        return f"[q: {self.q}, t: {self.t}, scale: {self.scale}, shift_1: {self.shift_1}, shift_2: {self.shift_2}]"


def _old_pose(g):
    return MonoDepthCameraPose(g.pose.q, g.pose.t, g.scale, g.shift1, g.shift2)


def _old_initial(initial):
    """a MonoDepthCameraPose / old-style image pair handed in as the initial value -> today's types (the pose itself is never read:
    the reference resets it, include/mdrp.h score_initial_model)"""
    if initial is None:
        return None
    if isinstance(initial, MonoDepthCameraPose):
        return MonoDepthTwoViewGeometry(CameraPose(initial.q, initial.t), initial.scale, initial.shift_1, initial.shift_2)
    if isinstance(getattr(initial, "pose", None), MonoDepthCameraPose):  # an old-style image pair
        return MonoDepthImagePair(_old_initial(initial.pose), getattr(initial, "camera1", None), getattr(initial, "camera2", None))
    return initial


def estimate_monodepth_pose(points2D_1, points2D_2, depth_1, depth_2, camera1, camera2, ransac_opt={}, bundle_opt={}, initial_pose=None):
    """older wheel's name of estimate_monodepth_relative_pose (poselib_old _core.pyi:441-470); returns (MonoDepthCameraPose, info)"""
    g, info = estimate_monodepth_relative_pose(points2D_1, points2D_2, depth_1, depth_2, camera1, camera2, ransac_opt, bundle_opt, _old_initial(initial_pose))
    return _old_pose(g), info


class _OldImagePair:
    """MonoDepthImagePair of the older wheel: camera1, camera2, pose (a MonoDepthCameraPose)"""

    def __init__(self, pair):
        self.camera1, self.camera2, self.pose = pair.camera1, pair.camera2, _old_pose(pair.geometry)

    def __repr__(self):
        return f"[pose: {self.pose}, camera1: {self.camera1}, camera2: {self.camera2}]"


def estimate_monodepth_shared_focal_pose(points2D_1, points2D_2, depth_1, depth_2, ransac_opt={}, bundle_opt={}, initial_image_pair=None):
    """older wheel's name of estimate_monodepth_shared_focal_relative_pose (poselib_old _core.pyi:472-483)"""
    p, info = estimate_monodepth_shared_focal_relative_pose(points2D_1, points2D_2, depth_1, depth_2, ransac_opt, bundle_opt,
                                                            _old_initial(initial_image_pair))
    return _OldImagePair(p), info


def estimate_monodepth_varying_focal_pose(points2D_1, points2D_2, depth_1, depth_2, ransac_opt={}, bundle_opt={}, initial_image_pair=None):
    """older wheel's name of estimate_monodepth_varying_focal_relative_pose (poselib_old _core.pyi:485-497)"""
    p, info = estimate_monodepth_varying_focal_relative_pose(points2D_1, points2D_2, depth_1, depth_2, ransac_opt, bundle_opt,
                                                             _old_initial(initial_image_pair))
    return _OldImagePair(p), info


# ------------------------------------------------------------------------------------------------ minimal solvers
def _solver(solver, x1, x2, d1, d2, wrap):
    x1 = np.asarray(x1, dtype=np.float64).reshape(3, 3)
    x2 = np.asarray(x2, dtype=np.float64).reshape(3, 3)
    out, n = _capi.default_handle().solver_batch(solver, x1[None], x2[None], np.asarray(d1, np.float64)[None], np.asarray(d2, np.float64)[None])
    return [wrap(out[0, k]) for k in range(int(n[0]))]


def monodepth_pose_3pt(x1, x2, d1, d2):
    """relpose_monodepth_3pt: scale + two shifts, <= 4 solutions (_core.pyi:614-619); x = homogeneous points with z = 1"""
    return _solver(_capi.SOLVER_SHIFT, x1, x2, d1, d2, _geometry_from_model)


def shared_focal_monodepth_pose_3pt(x1, x2, d1, d2):
    """relpose_monodepth_3pt_shared_focal (_core.pyi:871-876)"""
    return _solver(_capi.SOLVER_SHARED, x1, x2, d1, d2, _pair_from_model)


def varying_focal_monodepth_pose_4pt(x1, x2, d1, d2):
    """relpose_monodepth_3pt_varying_focal — the reference's Python name says 4pt (_core.pyi:914-919)"""
    return _solver(_capi.SOLVER_VARYING, x1, x2, d1, d2, _pair_from_model)


# ------------------------------------------------------------------------------------------------ fork-name adapters
# The paper scripts were written against kocurvik/PoseLib-mdrp@iccv-eval, whose RansacOptions carry experiment switches the
# released PR-152 binary (the wheel in /root/reference/demo, the only behaviour that can be pinned here) does not have.
# Every switch eval.py:93-127 / eval_shared_f.py:111-152 / eval_varying_f.py:105-150 can emit is listed: either it selects
# exactly what the PR-152 estimators do (-> mapped), or it selects a fork-only variant (-> NotImplementedError; parity
# unpinned).  Nothing is mapped "to the nearest behaviour" silently.
#   PR-152 calibrated estimator:  monodepth_estimate_shift=False: P3P on the depth of image 1, hybrid LM (Sampson + both
#       reprojections) of R, t, scale, Sampson scoring;  =True: 3-point scale + two shifts solver, the same LM also over both shifts.
#   PR-152 shared / varying focal estimators: 3-point scale + focal(s) solver without shifts, hybrid LM, Sampson scoring.
FORK_FLAG_TABLE = {
    # flag:               (values PR-152 behaviour corresponds to,                                   what any other value selects)
    "use_madpose":        ((False,), "MADPose solvers / optimisation (mad_poselib_*)"),
    "use_reldepth":       ((False,), "3p_reldepth relative-depth solver"),
    "use_4p4d":           ((False,), "4p4d solver (varying focal)"),
    "use_fundamental":    ((False,), "7-point fundamental-matrix baseline inside the monodepth estimator"),
    "use_eigen":          ((False,), "eigen-decomposition variant of the varying-focal solver"),
    "use_reproj":         ((False,), "reprojection-error scoring instead of Sampson (…_reproj)"),
    "optimize_symmetric": ((False,), "symmetric reprojection cost (…_sym_reproj, calibrated)"),
    "sym_repro":          ((False,), "symmetric reprojection cost (…_sym_reproj, focal)"),
    "no_normalization":   ((False,), "no scale normalisation of the pixel coordinates (NN)"),
    "graduated_steps":    ((0,), "graduated LO (GLO)"),
    "optimize_hybrid":    ((True,), "Sampson-only (or reprojection-only) LM instead of the hybrid cost"),
    "lo_iterations":      ((25,), "LO iteration count other than the binary's fixed 25 (nLO)"),
    "progressive_sampling": ((False,), "PROSAC sampling"),
}
_FORK_KEYS = tuple(FORK_FLAG_TABLE) + ("use_p3p", "use_ours", "solver_shift", "solver_scale", "optimize_shift", "all_permutations",
                                       "use_madpose_shift_optim", "weight_sampson")


def _map_fork_options(ransac_opt, kind=_capi.CALIB):
    """Option dict of the paper scripts -> option dict of the PR-152 estimators, or NotImplementedError naming the
    fork-only variant.  Plain PR-152 dicts (none of the fork keys) pass through unchanged."""
    ro = dict(ransac_opt or {})
    if not any(k in ro for k in _FORK_KEYS):
        return ro

    def fork_only(what):
        raise NotImplementedError(f"ransac options select a fork-only variant — {what} — whose behaviour cannot be pinned against "
                                  "the released PoseLib binary (SURVEY.md 8b; mdrp_amd.poselib.FORK_FLAG_TABLE)")

    for flag, (ok, what) in FORK_FLAG_TABLE.items():
        if flag in ro and ro[flag] not in ok and not (isinstance(ro[flag], bool) and int(ro[flag]) in [int(v) for v in ok if isinstance(v, bool)]):
            fork_only(f"{flag}={ro[flag]!r}: {what}")
    p3p, ours = bool(ro.get("use_p3p")), bool(ro.get("use_ours"))
    shift, scale, opt_shift = bool(ro.get("solver_shift")), bool(ro.get("solver_scale")), bool(ro.get("optimize_shift"))
    if kind == _capi.CALIB:
        if p3p and not ours and not shift and not opt_shift:
            ro["monodepth_estimate_shift"] = False          # p3p_hybrid*: P3P, LM without shifts
        elif ours and not p3p and shift and scale and opt_shift:
            ro["monodepth_estimate_shift"] = True           # 3p_ours_shift_scale_hybrid-s*: shift solver, LM over the shifts too
        elif p3p or ours or shift or scale or opt_shift:
            fork_only(f"use_p3p={p3p}, use_ours={ours}, solver_shift={shift}, solver_scale={scale}, optimize_shift={opt_shift}: the "
                      "released estimator ties the shift solver and the shift optimisation together (monodepth_estimate_shift) and "
                      "has no scale-only / shift-only 3-point solver")
        # all_permutations: the calibrated 3-point solvers are symmetric in the sample, permutations change nothing -> accepted
    else:
        if ours and not p3p and scale and not shift and not opt_shift:
            pass                                            # 3p_ours_scale_hybrid*: the PR-152 focal estimators
        elif p3p or ours or shift or scale or opt_shift:
            fork_only(f"use_p3p={p3p}, use_ours={ours}, solver_shift={shift}, solver_scale={scale}, optimize_shift={opt_shift}: the "
                      "released focal estimators are the 3-point scale + focal solvers without shifts")
        if ro.get("all_permutations"):
            fork_only("all_permutations=True: the shared-focal solver treats the third correspondence differently; the released "
                      "estimator runs the sample order as drawn")
    if "weight_sampson" in ro:
        ro["monodepth_weight_sampson"] = ro["weight_sampson"]
    return ro


def estimate_relative_pose_w_mono_depth(kp1, kp2, d, camera1, camera2, ransac_opt={}, bundle_opt={}):
    """eval.py:153 — d is (N,2) with the two depth columns; returns (pose-like with .R/.t, info)"""
    d = np.asarray(d, dtype=np.float64)
    g, info = estimate_monodepth_relative_pose(kp1, kp2, d[:, 0], d[:, 1], camera1, camera2, _map_fork_options(ransac_opt), bundle_opt)
    pose = g.pose
    pose.scale, pose.shift1, pose.shift2 = g.scale, g.shift1, g.shift2
    return pose, info


def estimate_shared_focal_monodepth_relative_pose(kp1, kp2, d, ransac_opt={}, bundle_opt={}):
    """eval_shared_f.py:177"""
    d = np.asarray(d, dtype=np.float64)
    return estimate_monodepth_shared_focal_relative_pose(kp1, kp2, d[:, 0], d[:, 1], _map_fork_options(ransac_opt, _capi.SHARED_FOCAL), bundle_opt)


def estimate_varying_focal_monodepth_relative_pose(kp1, kp2, d, ransac_opt={}, bundle_opt={}):
    """eval_varying_f.py:168"""
    d = np.asarray(d, dtype=np.float64)
    return estimate_monodepth_varying_focal_relative_pose(kp1, kp2, d[:, 0], d[:, 1], _map_fork_options(ransac_opt, _capi.VARYING_FOCAL), bundle_opt)


def _not_on_path(name, where):
    def stub(*args, **kwargs):
        raise NotImplementedError(f"poselib.{name} ({where}) is a non-monodepth baseline outside the accelerated hot path "
                                  "(SURVEY.md §8f-4 / DESIGN.md §8); use upstream PoseLib for it")
    stub.__name__ = name
    stub.__doc__ = f"{where}: not on the monodepth RANSAC path rebuilt here — raises NotImplementedError."
    return stub


# ------------------------------------------------------------------------------------------------ non-monodepth baselines
# the comparison rows of the paper tables run through the same kernels (sampler, MFMA candidate counts,
# fp32 bound, exact sweep, scan, walk) with the upstream estimators' solvers and Sampson-only LM (mdrp_classic.h).
def _stack2(points1, points2):
    if isinstance(points1, np.ndarray) and points1.ndim == 3:
        B, N = points1.shape[:2]
        return np.ascontiguousarray(points1, np.float64), np.ascontiguousarray(points2, np.float64), np.full(B, N, np.int32)
    B = len(points1)
    ns = np.array([len(p) for p in points1], dtype=np.int32)
    N = int(ns.max()) if B else 0
    x1 = np.zeros((B, N, 2)); x2 = np.zeros((B, N, 2))
    for i in range(B):
        x1[i, :ns[i]] = _as_points(points1[i]); x2[i, :ns[i]] = _as_points(points2[i])
    return x1, x2, ns


def _check_baseline_options(ransac_opt):
    """options of upstream RansacOptions that change what the baseline estimators do and are not built here"""
    ro = ransac_opt or {}
    if ro.get("progressive_sampling"):
        raise NotImplementedError("progressive_sampling (PROSAC) is not built (DESIGN.md 9)")
    if ro.get("real_focal_check"):
        raise NotImplementedError("real_focal_check (FundamentalEstimator drops models without real focal lengths) is not built")


def estimate_relative_pose_batch(points2D_1, points2D_2, cameras1, cameras2, ransac_opt=None, bundle_opt=None, device=0):
    """B calibrated pairs through the 5-point estimator.  Returns (list[CameraPose], list[info dict])."""
    _check_baseline_options(ransac_opt)
    x1, x2, ns = _stack2(points2D_1, points2D_2)
    B = len(ns)

    def cams(c):
        return _camera_records(c, B)

    res, mask = pipeline.estimate_host(_capi.RELPOSE_5PT, x1, x2, None, None, _capi.ransac_opt_from_dict(ransac_opt),
                                       _capi.bundle_opt_from_dict(bundle_opt), ns, cams(cameras1), cams(cameras2), device)
    return [CameraPose(r["model"]["q"].copy(), r["model"]["t"].copy()) for r in res], [_info(res[i], mask[i], ns[i]) for i in range(B)]


def estimate_fundamental_batch(points2D_1, points2D_2, ransac_opt=None, bundle_opt=None, device=0):
    """B pairs through the 7-point estimator.  Returns (list[3 x 3 ndarray], list[info dict])."""
    _check_baseline_options(ransac_opt)
    x1, x2, ns = _stack2(points2D_1, points2D_2)
    res, mask = pipeline.estimate_host(_capi.FUNDAMENTAL_7PT, x1, x2, None, None, _capi.ransac_opt_from_dict(ransac_opt),
                                       _capi.bundle_opt_from_dict(bundle_opt), ns, None, None, device)
    return [_capi.model_to_fundamental(r["model"]) for r in res], [_info(res[i], mask[i], ns[i]) for i in range(len(ns))]


def estimate_relative_pose(points2D_1, points2D_2, camera1, camera2, ransac_opt={}, bundle_opt={}, initial_pose=None):
    """Relative pose estimation with non-linear refinement (_core.pyi:504-529; eval.py:136 — the 5-point baseline).  As in the
    monodepth estimators, an initial pose only sets score_initial_model: ransac_relpose resets the pose itself."""
    poses, infos = estimate_relative_pose_batch([_as_points(points2D_1)], [_as_points(points2D_2)], camera1, camera2,
                                                _with_initial(initial_pose, ransac_opt), bundle_opt)
    return poses[0], infos[0]


def estimate_fundamental(points2D_1, points2D_2, ransac_opt={}, bundle_opt={}, initial_F=None):
    """Fundamental matrix estimation with non-linear refinement (_core.pyi:309-323; the 7-point baseline)."""
    if initial_F is not None:
        # unlike ransac_*relpose, which reset the model they are handed (an initial pose only sets score_initial_model), the reference's
        # ransac_fundamental scores the caller's F itself: not expressible through the flag, and no caller in the reference uses it
        raise NotImplementedError("estimate_fundamental with initial_F (score_initial_model on a caller's F) is not built")
    Fs, infos = estimate_fundamental_batch([_as_points(points2D_1)], [_as_points(points2D_2)], ransac_opt, bundle_opt)
    return Fs[0], infos[0]


def _bearings(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.ndim != 2 or x.shape[1] != 3:
        raise ValueError("bearings must have shape (K, 3)")
    return x


def relpose_5pt(x1, x2):
    """_core.pyi:851-854: five unit bearings per view -> list[CameraPose] in the reference's order"""
    out, n = _capi.default_handle(0).classic_solver_batch(_capi.RELPOSE_5PT, _bearings(x1)[None], _bearings(x2)[None])
    return [CameraPose(m["q"].copy(), m["t"].copy()) for m in out[0][:n[0]]]


def relpose_7pt(x1, x2):
    """seven unit bearings per view -> list of 3 x 3 fundamental matrices (unit Frobenius norm) in the reference's order"""
    out, n = _capi.default_handle(0).classic_solver_batch(_capi.FUNDAMENTAL_7PT, _bearings(x1)[None], _bearings(x2)[None])
    return [_capi.model_to_fundamental(m) for m in out[0][:n[0]]]


class ImagePair:
    """(_core.pyi:164-169): pose and the two cameras of the 6-point shared-focal estimator"""

    def __init__(self, pose=None, camera1=None, camera2=None):
        self.pose = pose if pose is not None else CameraPose()
        self.camera1 = camera1 if camera1 is not None else Camera()
        self.camera2 = camera2 if camera2 is not None else Camera()

    def __repr__(self):
        return f"ImagePair(pose={self.pose!r}, f1={self.camera1.focal():.6g}, f2={self.camera2.focal():.6g})"


def _pp_records(pp, B):
    """principal point(s) as the camera records the C ABI carries them in (MDRP_SHARED_6PT: cam1[i].params[0..1])"""
    pp = np.zeros(2) if pp is None else np.asarray(pp, dtype=np.float64)
    pp = np.broadcast_to(pp.reshape(-1, 2) if pp.size != 2 else pp.reshape(1, 2), (B, 2))
    rec = np.zeros(B, dtype=_capi.CAMERA_DTYPE)
    rec["params"][:, 0] = pp[:, 0]; rec["params"][:, 1] = pp[:, 1]
    return rec, pp


def estimate_shared_focal_relative_pose_batch(points2D_1, points2D_2, pp=None, ransac_opt=None, bundle_opt=None, device=0):
    """B pairs through the 6-point shared-focal estimator.  pp: one principal point for all pairs or (B, 2).
    Returns (list[ImagePair], list[info dict])."""
    _check_baseline_options(ransac_opt)
    x1, x2, ns = _stack2(points2D_1, points2D_2)
    B = len(ns)
    rec, ppb = _pp_records(pp, B)
    res, mask = pipeline.estimate_host(_capi.SHARED_6PT, x1, x2, None, None, _capi.ransac_opt_from_dict(ransac_opt),
                                       _capi.bundle_opt_from_dict(bundle_opt), ns, rec, rec, device)
    out = []
    for i, r in enumerate(res):
        f = float(r["model"]["f1"])
        cam = Camera("SIMPLE_PINHOLE", [f, float(ppb[i, 0]), float(ppb[i, 1])])
        out.append(ImagePair(CameraPose(r["model"]["q"].copy(), r["model"]["t"].copy()), cam, Camera("SIMPLE_PINHOLE", [f, float(ppb[i, 0]), float(ppb[i, 1])])))
    return out, [_info(res[i], mask[i], ns[i]) for i in range(B)]


def estimate_shared_focal_relative_pose(points2D_1, points2D_2, pp=None, ransac_opt={}, bundle_opt={}, initial_image_pair=None):
    """Relative pose with one unknown focal length shared by both images, 6-point solver + non-linear refinement
    (_core.pyi:531-543; /root/reference/eval_shared_f.py:161, where the fork's signature has no `pp` and the points are
    already centred: a dict in the third position is taken as ransac_opt).  An initial image pair only sets
    score_initial_model, like the initial pose of the other estimators."""
    if isinstance(pp, dict):  # fork call shape: (kp1, kp2, ransac_dict, bundle_dict)
        pp, ransac_opt, bundle_opt = None, pp, (ransac_opt if ransac_opt else bundle_opt)
    pairs, infos = estimate_shared_focal_relative_pose_batch([_as_points(points2D_1)], [_as_points(points2D_2)], pp,
                                                             _with_initial(initial_image_pair, ransac_opt), bundle_opt)
    return pairs[0], infos[0]


def relpose_6pt_shared_focal(x1, x2):
    """six homogeneous image points per view -> list[ImagePair], by ascending focal length (the reference's order is the
    order of its eigenvalue solver and is not reproduced)"""
    out, n = _capi.default_handle(0).classic_solver_batch(_capi.SHARED_6PT, _bearings(x1)[None], _bearings(x2)[None])
    return [ImagePair(CameraPose(m["q"].copy(), m["t"].copy()), Camera("SIMPLE_PINHOLE", [float(m["f1"]), 0.0, 0.0]),
                      Camera("SIMPLE_PINHOLE", [float(m["f2"]), 0.0, 0.0])) for m in out[0][:n[0]]]


# names the reference scripts also reach for: present so that a swapped import fails with a clear message at the call
estimate_relative_pose_w_relative_depth = _not_on_path("estimate_relative_pose_w_relative_depth", "fork-only variant, eval.py:140 (commented out upstream)")


# ------------------------------------------------------------------------------------------------ device-resident batches
_TORCH_HANDLE_CACHE = 4  # handles kept per thread (each owns its scratch buffers); least recently used is closed
_torch_tls = None


def _torch_handle(dev, stream_ptr):
    """the calling thread's handle bound to (device, stream), LRU-bounded; stream_ptr 0 = the legacy default stream"""
    global _torch_tls
    import collections
    import threading
    if _torch_tls is None:
        _torch_tls = threading.local()
    cache = getattr(_torch_tls, "cache", None)
    if cache is None:
        cache = _torch_tls.cache = collections.OrderedDict()
    key = (dev, stream_ptr)
    h = cache.pop(key, None)
    if h is None:
        h = _capi.Handle(dev, stream_ptr)
        while len(cache) >= _TORCH_HANDLE_CACHE:
            cache.popitem(last=False)[1].close()
    cache[key] = h
    return h


def estimate_batch_torch(kind, points2D_1, points2D_2, depth_1, depth_2, cameras1=None, cameras2=None, ransac_opt=None,
                         bundle_opt=None, n_per_pair=None):
    """Batch that already lives on the GPU (e.g. matcher output): `points2D_*` (B, N, 2) and `depth_*` (B, N) float64 torch
    tensors on a ROCm device; the work is queued on that device's CURRENT torch stream — including torch's default
    (null) stream — so it is ordered after whatever produced the inputs there and before later consumers of the mask;
    nothing crosses PCIe except the 136-byte result records.  kind: "calibrated" | "shared_focal" | "varying_focal".
    Returns (records: numpy structured array with `model`, `refinements`, `iterations`, `num_inliers`, `inlier_ratio`,
    `model_score`; inlier mask: (B, N) uint8 tensor on the device).  Ragged batches: pad and pass `n_per_pair`
    (sequence, numpy array or tensor on any device)."""
    import torch
    kinds = {"calibrated": _capi.CALIB, "shared_focal": _capi.SHARED_FOCAL, "varying_focal": _capi.VARYING_FOCAL}
    k = kinds[kind] if isinstance(kind, str) else int(kind)
    x1, x2, d1, d2 = (t.contiguous() for t in (points2D_1, points2D_2, depth_1, depth_2))
    for t in (x1, x2, d1, d2):
        if not (t.is_cuda and t.dtype == torch.float64):
            raise ValueError("estimate_batch_torch needs float64 tensors on the GPU")
        if t.device != x1.device:
            raise ValueError("all tensors must live on the same device")
    B, N = d1.shape
    if x1.shape != (B, N, 2) or x2.shape != (B, N, 2) or d2.shape != (B, N):
        raise ValueError("shapes must be (B, N, 2), (B, N, 2), (B, N), (B, N)")
    dev = x1.device.index if x1.device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(x1.device)
    h = _torch_handle(dev, int(stream.cuda_stream))
    if n_per_pair is not None and isinstance(n_per_pair, torch.Tensor):
        n_per_pair = n_per_pair.detach().cpu().numpy()
    cams1 = cams2 = None
    if k == _capi.CALIB:
        cams1, cams2 = _camera_records(cameras1, B), _camera_records(cameras2, B)
    with torch.cuda.device(x1.device):
        mask = torch.zeros((B, N), dtype=torch.uint8, device=x1.device)
        h.estimate_batch_device(k, x1.data_ptr(), x2.data_ptr(), d1.data_ptr(), d2.data_ptr(), B, N,
                                _capi.ransac_opt_from_dict(ransac_opt), _capi.bundle_opt_from_dict(bundle_opt), n_per_pair, cams1, cams2,
                                mask.data_ptr())
        res = h.fetch_results(B)
    return res, mask
