"""ctypes binding of libmdrp_hip.so (include/mdrp.h) — the only route from Python to the HIP kernels.

There is no CPU fallback: if the shared library is missing or no gfx950 device is usable, every entry point
raises.  Build with `python -c "import __graft_entry__ as g; g.build()"` (or mdrp_amd/build.py).
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDRP_LIB") or os.path.join(_HERE, "libmdrp_hip.so")  # MDRP_LIB: experiment builds (tools/)

CALIB, SHARED_FOCAL, VARYING_FOCAL = 0, 1, 2
RELPOSE_5PT, SHARED_6PT, FUNDAMENTAL_7PT = 3, 4, 5  # non-monodepth baselines (d1 = d2 = None)
MEM_HOST, MEM_DEVICE = 0, 1
SOLVER_P3P, SOLVER_SHIFT, SOLVER_SHARED, SOLVER_VARYING = 0, 1, 2, 3

EXPORTS = (
    "mdrp_create_", "mdrp_create_on_stream_", "mdrp_destroy", "mdrp_last_error", "mdrp_version", "mdrp_abi_version", "mdrp_hip_build_version", "mdrp_synchronize", "mdrp_estimate_batch",
    "mdrp_estimate_batch_async", "mdrp_fetch_results", "mdrp_copy_results_device", "mdrp_solver_batch", "mdrp_score_models", "mdrp_count_candidates", "mdrp_bound_models", "mdrp_refine_models",
    "mdrp_last_sweep_stats", "mdrp_last_stats", "mdrp_last_stats_sized", "mdrp_classic_solver_batch",
)


class Model(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3), ("scale", C.c_double), ("shift1", C.c_double),
                ("shift2", C.c_double), ("f1", C.c_double), ("f2", C.c_double)]


class RansacOpt(C.Structure):
    _fields_ = [("max_iterations", C.c_uint64), ("min_iterations", C.c_uint64), ("dyn_num_trials_mult", C.c_double),
                ("success_prob", C.c_double), ("max_reproj_error", C.c_double), ("max_epipolar_error", C.c_double),
                ("seed", C.c_uint64), ("monodepth_estimate_shift", C.c_int32), ("monodepth_weight_sampson", C.c_float),
                ("score_initial_model", C.c_int32), ("progressive_sampling", C.c_int32), ("max_prosac_iterations", C.c_uint64),
                ("real_focal_check", C.c_int32), ("reserved_", C.c_int32)]


class BundleOpt(C.Structure):
    _fields_ = [("max_iterations", C.c_uint64), ("loss_type", C.c_int32), ("loss_scale", C.c_double),
                ("gradient_tol", C.c_double), ("step_tol", C.c_double), ("initial_lambda", C.c_double),
                ("min_lambda", C.c_double), ("max_lambda", C.c_double)]


class Camera(C.Structure):
    _fields_ = [("model_id", C.c_int32), ("pad_", C.c_int32), ("params", C.c_double * 4)]


class Stats(C.Structure):
    _fields_ = [("count_ms", C.c_double), ("count_launches", C.c_int64), ("sweep_ms", C.c_double), ("sweep_launches", C.c_int64),
                ("evals_algorithmic", C.c_int64), ("evals_mfma", C.c_int64), ("evals_fp64", C.c_int64), ("evals_bound", C.c_int64),
                ("lo_ms", C.c_double), ("lo_launches", C.c_int64), ("final_ms", C.c_double), ("final_launches", C.c_int64),
                ("bound_ms", C.c_double), ("bound_launches", C.c_int64), ("solve_ms", C.c_double), ("solve_launches", C.c_int64),
                ("lm_cost_evals", C.c_int64), ("lm_accum_evals", C.c_int64), ("final_cost_evals", C.c_int64), ("final_accum_evals", C.c_int64),
                ("fuse_gate_timeouts", C.c_int64), ("fuse_wait_timeouts", C.c_int64),  # ABI 0.3 (mdrp_last_stats_sized)
                ("first_chunk", C.c_int64)]  # ABI 0.5


class Result(C.Structure):
    _fields_ = [("model", Model), ("refinements", C.c_uint64), ("iterations", C.c_uint64), ("num_inliers", C.c_uint64),
                ("inlier_ratio", C.c_double), ("model_score", C.c_double)]


MODEL_DTYPE = np.dtype([("q", "f8", 4), ("t", "f8", 3), ("scale", "f8"), ("shift1", "f8"), ("shift2", "f8"), ("f1", "f8"), ("f2", "f8")])
RESULT_DTYPE = np.dtype([("model", MODEL_DTYPE), ("refinements", "u8"), ("iterations", "u8"), ("num_inliers", "u8"),
                         ("inlier_ratio", "f8"), ("model_score", "f8")])
CAMERA_DTYPE = np.dtype([("model_id", "i4"), ("pad_", "i4"), ("params", "f8", 4)])
assert MODEL_DTYPE.itemsize == C.sizeof(Model) and RESULT_DTYPE.itemsize == C.sizeof(Result) and CAMERA_DTYPE.itemsize == C.sizeof(Camera)

_lib = None
_lib_lock = threading.Lock()


class MdrpError(RuntimeError):
    pass


def load_library():
    """dlopen libmdrp_hip.so and declare prototypes.  Raises MdrpError if it is not built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MdrpError(f"{LIB_PATH} is not built (run __graft_entry__.build()); mdrp_amd has no CPU fallback")
        _ensure_hip_runtime()  # the library binds its hip* symbols to the process's one runtime when it is loaded
        lib = C.CDLL(LIB_PATH)
        _check_hip_runtime_version(lib)
        vp, dp, ip = C.c_void_p, C.c_void_p, C.c_void_p
        lib.mdrp_last_error.restype = C.c_char_p
        lib.mdrp_version.restype = C.c_char_p
        abi = getattr(lib, "mdrp_abi_version", None)  # (a library from before ABI 0.4, e.g. through MDRP_LIB, has no such symbol)
        if abi is None or getattr(lib, "mdrp_create_", None) is None:
            raise MdrpError(f"{LIB_PATH} predates ABI {ABI_VERSION:#x} (no mdrp_abi_version / mdrp_create_): rebuild (mdrp_amd/build.py)")
        abi.restype = C.c_int
        if abi() != ABI_VERSION:
            raise MdrpError(f"{LIB_PATH} speaks ABI {abi():#x}, this binding {ABI_VERSION:#x}: rebuild (mdrp_amd/build.py)")
        lib.mdrp_create_.argtypes = [C.c_int, vp, C.POINTER(vp), C.c_int, C.c_int]
        lib.mdrp_create_on_stream_.argtypes = [C.c_int, vp, C.POINTER(vp), C.c_int, C.c_int]
        lib.mdrp_destroy.argtypes = [vp]
        lib.mdrp_destroy.restype = None
        lib.mdrp_synchronize.argtypes = [vp]
        lib.mdrp_estimate_batch.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, ip, vp, vp,
                                            C.POINTER(RansacOpt), C.POINTER(BundleOpt), vp, vp]
        lib.mdrp_estimate_batch_async.argtypes = [vp, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, ip, vp, vp,
                                                  C.POINTER(RansacOpt), C.POINTER(BundleOpt), vp]
        lib.mdrp_fetch_results.argtypes = [vp, vp, C.c_int]
        lib.mdrp_copy_results_device.argtypes = [vp, vp, C.c_int]
        lib.mdrp_solver_batch.argtypes = [vp, C.c_int, dp, dp, dp, dp, C.c_int, vp, vp]
        lib.mdrp_score_models.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, dp, dp, C.c_int, C.c_double, vp, vp]
        lib.mdrp_count_candidates.argtypes = [vp, C.c_int, vp, C.c_int, dp, dp, C.c_int, C.c_double, vp]
        lib.mdrp_bound_models.argtypes = [vp, C.c_int, vp, C.c_int, dp, dp, C.c_int, C.c_double, vp, vp]
        lib.mdrp_refine_models.argtypes = [vp, C.c_int, vp, C.c_int, dp, dp, dp, dp, C.c_int, C.c_double, C.c_double,
                                           C.POINTER(BundleOpt), C.c_int, vp]
        lib.mdrp_last_sweep_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.mdrp_last_stats.argtypes = [vp, C.POINTER(Stats)]
        lib.mdrp_last_stats_sized.argtypes = [vp, C.POINTER(Stats), C.c_size_t]
        lib.mdrp_classic_solver_batch.argtypes = [vp, C.c_int, dp, dp, C.c_int, vp, vp]
        _lib = lib
        return lib


ABI_VERSION = 0x00000005  # include/mdrp.h MDRP_ABI_VERSION
ERR_UNSUPPORTED = 4  # include/mdrp.h MDRP_ERR_UNSUPPORTED: a reference option that selects behaviour the library does not build


def _check(lib, rc):
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(f"mdrp: {lib.mdrp_last_error().decode(errors='replace')} (DESIGN.md 9)")
    if rc != 0:
        raise MdrpError(f"mdrp error {rc}: {lib.mdrp_last_error().decode(errors='replace')}")


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def ransac_opt_from_dict(d=None):
    """RansacOptions from a poselib-style dict; defaults as the reference's pybind wrapper;
    unknown keys are ignored like the reference does.  progressive_sampling / max_prosac_iterations / real_focal_check travel to the
    library, which refuses what it does not build (MDRP_ERR_UNSUPPORTED -> NotImplementedError) instead of ignoring it."""
    d = d or {}
    return RansacOpt(int(d.get("max_iterations", 100000)), int(d.get("min_iterations", 1000)),
                     float(d.get("dyn_num_trials_mult", 3.0)), float(d.get("success_prob", 0.9999)),
                     float(d.get("max_reproj_error", 12.0)), float(d.get("max_epipolar_error", 1.0)),
                     int(d.get("seed", 0)), int(bool(d.get("monodepth_estimate_shift", False))),
                     float(d.get("monodepth_weight_sampson", 1.0)), int(bool(d.get("score_initial_model", False))),
                     int(bool(d.get("progressive_sampling", False))), int(d.get("max_prosac_iterations", 100000)),
                     int(bool(d.get("real_focal_check", False))), 0)


LOSS_TYPES = {"TRIVIAL": 0, "TRUNCATED": 1, "HUBER": 2, "CAUCHY": 3, "TRUNCATED_CAUCHY": 4, "TRUNCATED_LE_ZACH": 5}


def bundle_opt_from_dict(d=None):
    d = d or {}
    lt = d.get("loss_type", "CAUCHY")
    if isinstance(lt, str):
        if lt.upper() not in LOSS_TYPES:
            raise ValueError(f"unknown loss_type {lt!r}")
        lt = LOSS_TYPES[lt.upper()]
    return BundleOpt(int(d.get("max_iterations", 100)), int(lt), float(d.get("loss_scale", 1.0)),
                     float(d.get("gradient_tol", 1e-10)), float(d.get("step_tol", 1e-8)), float(d.get("initial_lambda", 1e-3)),
                     float(d.get("min_lambda", 1e-10)), float(d.get("max_lambda", 1e10)))


def library_version():
    return load_library().mdrp_version().decode()


def library_source_hash():
    """the source hash the loaded library was built from (mdrp_version(); compare with build.source_hash())"""
    v = library_version()
    return v.split("MDRP_SRC_HASH=", 1)[1][:16] if "MDRP_SRC_HASH=" in v else None


_hip_runtime = None
hip_versions = None  # {"built": HIP_VERSION of the toolchain, "runtime": hipRuntimeGetVersion()} once the library is loaded


def _check_hip_runtime_version(lib):
    """The library binds to whatever HIP runtime the process holds (-no-hip-rt).  Compare that runtime's version
    (hipRuntimeGetVersion: major * 10^7 + minor * 10^5 + patch) with the HIP_VERSION hipcc compiled the library against
    (mdrp_hip_build_version()): a different MAJOR is refused (fat-binary registration and struct layouts may differ), a different
    minor is reported once."""
    import warnings
    try:
        lib.mdrp_hip_build_version.restype = C.c_int
        built = int(lib.mdrp_hip_build_version())
        v = C.c_int(0)
        rt = _hip_runtime.hipRuntimeGetVersion
        rt.argtypes = [C.POINTER(C.c_int)]
        if rt(C.byref(v)) != 0:
            return
    except (AttributeError, OSError):
        return
    have = int(v.value)
    if have // 10_000_000 != built // 10_000_000:
        raise MdrpError(f"libmdrp_hip.so was built against HIP {built // 10_000_000}.{built // 100_000 % 100} but the process's HIP runtime "
                        f"({getattr(_hip_runtime, '_name', '?')}) is {have // 10_000_000}.{have // 100_000 % 100}: set MDRP_HIP_RUNTIME to a "
                        "matching libamdhip64.so or rebuild (mdrp_amd/build.py)")
    global hip_versions
    hip_versions = {"built": built, "runtime": have}  # e.g. PyTorch-ROCm 2.10 wheels bundle HIP 7.0, this image's hipcc is 7.2: same major
    if have // 100_000 != built // 100_000 and os.environ.get("MDRP_DEBUG"):
        warnings.warn(f"mdrp: HIP runtime {have // 10_000_000}.{have // 100_000 % 100} in this process, library built against "
                      f"{built // 10_000_000}.{built // 100_000 % 100} (same major: continuing)", RuntimeWarning, stacklevel=3)


def _ensure_hip_runtime():
    """libmdrp_hip.so is linked WITHOUT a HIP runtime of its own (build.py: -no-hip-rt): its hip* symbols bind to whatever runtime
    the process has, so that there is exactly one — streams, events and device pointers of the host framework are then valid
    inside the library by construction.  PyTorch-ROCm wheels bundle a runtime (torch/lib/libamdhip64.so, no SONAME): if torch is
    imported or merely installed, THAT file is (re)opened with RTLD_GLOBAL (dlopen of an already loaded library only widens its
    scope); otherwise the system runtime is.  MDRP_HIP_RUNTIME=/path/to/libamdhip64.so overrides the choice."""
    global _hip_runtime
    if _hip_runtime is not None:
        return _hip_runtime
    import sys
    cands = []
    if os.environ.get("MDRP_HIP_RUNTIME"):
        cands.append(os.environ["MDRP_HIP_RUNTIME"])
    try:
        torch = sys.modules.get("torch")
        if torch is not None:
            tdir = os.path.dirname(torch.__file__)
        else:
            import importlib.util
            spec = importlib.util.find_spec("torch")
            tdir = list(spec.submodule_search_locations)[0] if spec is not None and spec.submodule_search_locations else None
        if tdir and os.path.exists(os.path.join(tdir, "lib", "libamdhip64.so")):
            cands.append(os.path.join(tdir, "lib", "libamdhip64.so"))
    except Exception:
        pass
    import glob
    # the SONAME major follows the ROCm release: whatever /opt/rocm (or ROCM_PATH) ships, newest first, then the linker's search
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    def _soname_version(path):
        tail = os.path.basename(path).split(".so.", 1)[-1]
        return tuple(int(x) for x in tail.split(".") if x.isdigit())
    cands += sorted(glob.glob(os.path.join(rocm, "lib", "libamdhip64.so.*")), key=_soname_version, reverse=True) + ["libamdhip64.so", os.path.join(rocm, "lib", "libamdhip64.so")]
    errs = []
    for c in cands:
        try:
            _hip_runtime = C.CDLL(c, mode=C.RTLD_GLOBAL)
            return _hip_runtime
        except OSError as e:
            errs.append(f"{c}: {e}")
    raise MdrpError("no HIP runtime (libamdhip64) could be loaded: " + "; ".join(errs))


class Handle:
    """One handle = one HIP device + one stream + its scratch buffers.  Calls on one handle are serialised inside the
    library; use one handle per host thread for concurrency (default_handle() does).

    stream=None: the handle creates its own stream.  stream=<int>: a hipStream_t of the caller, used as given —
    0 is the device's legacy default stream (torch.cuda.current_stream().cuda_stream is 0 on torch's default stream)."""

    def __init__(self, device=0, stream=None):
        self._lib = load_library()
        h = C.c_void_p()
        if stream is None:
            _check(self._lib, self._lib.mdrp_create_(int(device), None, C.byref(h), ABI_VERSION, C.sizeof(RansacOpt)))
        else:
            _check(self._lib, self._lib.mdrp_create_on_stream_(int(device), C.c_void_p(int(stream)) if stream else None, C.byref(h), ABI_VERSION, C.sizeof(RansacOpt)))
        self._h = h
        self.device = int(device)
        self.stream = stream

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mdrp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self._lib, self._lib.mdrp_synchronize(self._h))

    # ---- batched estimators, host (numpy) buffers
    def estimate_batch(self, kind, x1, x2, d1, d2, ropt, bopt, n_per_pair=None, cam1=None, cam2=None, want_mask=True):
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        x2 = np.ascontiguousarray(x2, dtype=np.float64)
        if x1.ndim != 3 or x1.shape[2] != 2 or x2.shape != x1.shape:
            raise ValueError("expected x1,x2 (B,N,2)")
        if kind >= RELPOSE_5PT:
            d1 = d2 = None  # the non-monodepth baselines take no depths
        else:
            d1 = np.ascontiguousarray(d1, dtype=np.float64)
            d2 = np.ascontiguousarray(d2, dtype=np.float64)
            if d1.shape != x1.shape[:2] or d2.shape != d1.shape:
                raise ValueError("expected d1,d2 (B,N)")
        B, N = x1.shape[:2]
        npp = None if n_per_pair is None else np.ascontiguousarray(n_per_pair, dtype=np.int32)
        out = np.zeros(B, dtype=RESULT_DTYPE)
        mask = np.zeros((B, N), dtype=np.uint8) if want_mask else None
        c1 = None if cam1 is None else np.ascontiguousarray(cam1, dtype=CAMERA_DTYPE)
        c2 = None if cam2 is None else np.ascontiguousarray(cam2, dtype=CAMERA_DTYPE)
        _check(self._lib, self._lib.mdrp_estimate_batch(self._h, kind, MEM_HOST, _ptr(x1), _ptr(x2), _ptr(d1), _ptr(d2), B, N,
                                                        _ptr(npp), _ptr(c1), _ptr(c2), C.byref(ropt), C.byref(bopt),
                                                        _ptr(out), _ptr(mask)))
        return out, mask

    # ---- batched estimators, device pointers (ints), results fetched separately
    def estimate_batch_device(self, kind, x1_ptr, x2_ptr, d1_ptr, d2_ptr, batch, n_max, ropt, bopt, n_per_pair=None,
                              cam1=None, cam2=None, mask_ptr=None):
        npp = None if n_per_pair is None else np.ascontiguousarray(n_per_pair, dtype=np.int32)
        c1 = None if cam1 is None else np.ascontiguousarray(cam1, dtype=CAMERA_DTYPE)
        c2 = None if cam2 is None else np.ascontiguousarray(cam2, dtype=CAMERA_DTYPE)
        _check(self._lib, self._lib.mdrp_estimate_batch_async(self._h, kind, C.c_void_p(x1_ptr), C.c_void_p(x2_ptr),
                                                              C.c_void_p(d1_ptr) if d1_ptr else None, C.c_void_p(d2_ptr) if d2_ptr else None, int(batch), int(n_max),
                                                              _ptr(npp), _ptr(c1), _ptr(c2), C.byref(ropt), C.byref(bopt),
                                                              C.c_void_p(mask_ptr) if mask_ptr else None))

    def fetch_results(self, batch):
        out = np.zeros(batch, dtype=RESULT_DTYPE)
        _check(self._lib, self._lib.mdrp_fetch_results(self._h, _ptr(out), int(batch)))
        return out

    def copy_results_device(self, dst_ptr, batch):
        """the result records of the last device-resident estimate into device memory at dst_ptr (batch x 136 bytes)"""
        _check(self._lib, self._lib.mdrp_copy_results_device(self._h, C.c_void_p(dst_ptr), int(batch)))

    def last_sweep_stats(self):
        ms, launches, evals = C.c_double(0), C.c_int64(0), C.c_int64(0)
        _check(self._lib, self._lib.mdrp_last_sweep_stats(self._h, C.byref(ms), C.byref(launches), C.byref(evals)))
        return ms.value, launches.value, evals.value

    def last_stats(self):
        """dict: time and launches of k_count (MFMA) and k_score (fp64), and the evaluation counters of the last estimate call"""
        st = Stats()
        _check(self._lib, self._lib.mdrp_last_stats_sized(self._h, C.byref(st), C.sizeof(Stats)))
        d = {k: getattr(st, k) for k, _ in Stats._fields_}
        d["fuse_timeouts"] = d["fuse_gate_timeouts"] + d["fuse_wait_timeouts"]
        return d

    # ---- unit-parity entry points
    def solver_batch(self, solver, x1h, x2h, d1, d2):
        x1h = np.ascontiguousarray(x1h, dtype=np.float64).reshape(-1, 3, 3)
        x2h = np.ascontiguousarray(x2h, dtype=np.float64).reshape(-1, 3, 3)
        d1 = np.ascontiguousarray(d1, dtype=np.float64).reshape(-1, 3)
        d2 = np.ascontiguousarray(d2, dtype=np.float64).reshape(-1, 3)
        count = len(d1)
        out = np.zeros((count, 4), dtype=MODEL_DTYPE)
        n_out = np.zeros(count, dtype=np.int32)
        _check(self._lib, self._lib.mdrp_solver_batch(self._h, int(solver), _ptr(x1h), _ptr(x2h), _ptr(d1), _ptr(d2), count,
                                                      _ptr(out), _ptr(n_out)))
        return out, n_out

    def classic_solver_batch(self, kind, x1h, x2h):
        """relpose_5pt (kind 3: x1h, x2h (count, 5, 3) unit bearings -> up to 10 poses) / relpose_6pt_shared_focal (kind 4:
        (count, 6, 3) -> up to 15 poses with f1 = f2 = f, by ascending f) / relpose_7pt (kind 5: (count, 7, 3) -> up to 3
        fundamental matrices in the models' first nine doubles)"""
        K, M = {RELPOSE_5PT: (5, 10), SHARED_6PT: (6, 15), FUNDAMENTAL_7PT: (7, 3)}[kind]
        x1h = np.ascontiguousarray(x1h, dtype=np.float64).reshape(-1, K, 3)
        x2h = np.ascontiguousarray(x2h, dtype=np.float64).reshape(-1, K, 3)
        count = len(x1h)
        out = np.zeros((count, M), dtype=MODEL_DTYPE)
        n_out = np.zeros(count, dtype=np.int32)
        _check(self._lib, self._lib.mdrp_classic_solver_batch(self._h, int(kind), _ptr(x1h), _ptr(x2h), count, _ptr(out), _ptr(n_out)))
        return out, n_out

    def score_models(self, kind, models, x1, x2, sq_threshold):
        models = np.ascontiguousarray(models, dtype=MODEL_DTYPE).reshape(-1)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        x2 = np.ascontiguousarray(x2, dtype=np.float64)
        scores = np.zeros(len(models))
        counts = np.zeros(len(models), dtype=np.int32)
        _check(self._lib, self._lib.mdrp_score_models(self._h, int(kind), MEM_HOST, _ptr(models), len(models), _ptr(x1), _ptr(x2),
                                                      len(x1), float(sq_threshold), _ptr(scores), _ptr(counts)))
        return scores, counts

    def count_candidates(self, kind, models, x1, x2, sq_threshold):
        """k_count alone: per model an upper bound on its inlier count (MFMA pre-pass)"""
        models = np.ascontiguousarray(models, dtype=MODEL_DTYPE).reshape(-1)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        x2 = np.ascontiguousarray(x2, dtype=np.float64)
        cand = np.zeros(len(models), dtype=np.int32)
        _check(self._lib, self._lib.mdrp_count_candidates(self._h, int(kind), _ptr(models), len(models), _ptr(x1), _ptr(x2), len(x1),
                                                          float(sq_threshold), _ptr(cand)))
        return cand

    def bound_models(self, kind, models, x1, x2, sq_threshold):
        """k_bound alone: per model (lower bound of the MSAC score, upper bound of the inlier count) from the fp32 stage"""
        models = np.ascontiguousarray(models, dtype=MODEL_DTYPE).reshape(-1)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        x2 = np.ascontiguousarray(x2, dtype=np.float64)
        lb = np.zeros(len(models))
        ub = np.zeros(len(models), dtype=np.int32)
        _check(self._lib, self._lib.mdrp_bound_models(self._h, int(kind), _ptr(models), len(models), _ptr(x1), _ptr(x2), len(x1),
                                                      float(sq_threshold), _ptr(lb), _ptr(ub)))
        return lb, ub

    def score_models_device(self, kind, models_ptr, num_models, x1_ptr, x2_ptr, n, sq_threshold, scores_ptr, counts_ptr):
        _check(self._lib, self._lib.mdrp_score_models(self._h, int(kind), MEM_DEVICE, C.c_void_p(models_ptr), int(num_models),
                                                      C.c_void_p(x1_ptr), C.c_void_p(x2_ptr), int(n), float(sq_threshold),
                                                      C.c_void_p(scores_ptr), C.c_void_p(counts_ptr)))

    def refine_models(self, kind, models, x1, x2, d1, d2, scale_reproj, weight_sampson, bopt, estimate_shift=False):
        models = np.ascontiguousarray(models, dtype=MODEL_DTYPE).reshape(-1).copy()
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        x2 = np.ascontiguousarray(x2, dtype=np.float64)
        d1 = None if d1 is None else np.ascontiguousarray(d1, dtype=np.float64)
        d2 = None if d2 is None else np.ascontiguousarray(d2, dtype=np.float64)
        cost = np.zeros(len(models))
        _check(self._lib, self._lib.mdrp_refine_models(self._h, int(kind), _ptr(models), len(models), _ptr(x1), _ptr(x2), _ptr(d1),
                                                       _ptr(d2), len(x1), float(scale_reproj), float(weight_sampson), C.byref(bopt),
                                                       int(bool(estimate_shift)), _ptr(cost)))
        return models, cost


_default_tls = threading.local()


def default_handle(device=0):
    """The calling THREAD's handle for `device` (created on first use): the drop-in entry points release the GIL inside
    the C call like the reference does, so two Python threads must not share scratch buffers — each gets its own handle
    and stream, and their batches run concurrently on the GPU."""
    handles = getattr(_default_tls, "handles", None)
    if handles is None:
        handles = _default_tls.handles = {}
    h = handles.get(device)
    if h is None:
        h = handles[device] = Handle(device)
    return h


def fundamental_to_model(F):
    """a 3 x 3 fundamental matrix as an MDRP_FUNDAMENTAL_7PT model record (row-major in the first nine doubles)"""
    m = np.zeros((), dtype=MODEL_DTYPE)
    flat = np.asarray(F, dtype=np.float64).reshape(9)
    m["q"] = flat[:4]; m["t"] = flat[4:7]; m["scale"] = flat[7]; m["shift1"] = flat[8]
    m["f1"] = m["f2"] = 1.0
    return m


def model_to_fundamental(m):
    return np.r_[m["q"], m["t"], m["scale"], m["shift1"]].reshape(3, 3).copy()


def model_to_array(m):
    """structured MODEL_DTYPE scalar -> flat (12,) float64 [q t scale shift1 shift2 f1 f2]"""
    return np.concatenate([m["q"], m["t"], [m["scale"], m["shift1"], m["shift2"], m["f1"], m["f2"]]]).astype(np.float64)


def array_to_models(a):
    a = np.asarray(a, dtype=np.float64).reshape(-1, 12)
    out = np.zeros(len(a), dtype=MODEL_DTYPE)
    out["q"] = a[:, :4]; out["t"] = a[:, 4:7]; out["scale"] = a[:, 7]; out["shift1"] = a[:, 8]; out["shift2"] = a[:, 9]
    out["f1"] = a[:, 10]; out["f2"] = a[:, 11]
    return out
