"""Shared comparison helpers for the parity tests."""
import numpy as np


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def model_diff(a, b):
    """max of |dR| and relative difference of (t, scale, shifts, focals); a, b are 12-wide models."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    with np.errstate(invalid="ignore"):  # (inf - inf in models of corrupted inputs: the NaN travels to the caller's comparison)
        dR = np.abs(quat_to_R(a[:4]) - quat_to_R(b[:4])).max()
        rest = np.abs((a[4:] - b[4:]) / (1.0 + np.abs(b[4:]))).max()
    # np.max propagates NaN where Python's max(0.1, nan) is 0.1: a NaN translation / scale / shift / focal against a finite reference must not pass
    # a `model_diff(...) < tol` assertion just because the rotation is finite
    return float(np.max([dR, rest]))


def match_solution_sets(A, B, tol=1e-6):
    """Greedy one-to-one matching of two solution lists; True iff same size and every pair within tol."""
    if len(A) != len(B):
        return False
    used = set()
    for a in A:
        best, bj = np.inf, -1
        for j, b in enumerate(B):
            if j in used:
                continue
            e = model_diff(a, b)
            if e < best:
                best, bj = e, j
        if best > tol:
            return False
        used.add(bj)
    return True


def widen(sol, width=12):
    out = np.ones(width)
    out[: len(sol)] = sol
    return out


# ---- enumerated deviations from the reference binary, by fixture index -------------------------------
# tests/golden/solvers.npz: problems on which the reference binary itself returns NaN models (nothing to compare)
REFERENCE_NAN_SOLUTIONS = {"p3p": (), "calib_shift": (21,), "shared": (), "varying": ()}

# tests/golden/estimate_full.npz: our LO count minus the reference's.  Found with the per-iteration replay of both
# sides' minimal models (tests/tools/diag_lo_count.py):
#   case 4 (calibrated + shift, pair 1): at iteration 2827 the reference's relpose_monodepth_3pt misses a true root
#           that ours finds; that model has 993 inliers, sets a record and costs us one LO.
# Iterations, inliers, score, mask and model are identical.  (Case 1 — the reference's P3P returns four NaN poses for the sample of
# iteration 0, one LO that cannot change anything — was a deviation until the second half of round 5: the NaN poses are reproduced now.)
KNOWN_LO_COUNT_DEVIATIONS = {4: +1}

# tests/golden/initial.npz: case 11 (varying focal, ALL correspondences identical, score_initial_model) was an oracle-only deviation up to round 5: the LO
# that starts from the reset identity pose (E = 0: every Sampson residual 0/0) moved in our LM and stays put in the reference's.  Cause: the reference's
# truncated losses are std::min(r2, t2), which hands a NaN residual on — the cost is NaN, no step is ever accepted.  orc_refine.c does the same now.
INITIAL_ORACLE_DEVIATIONS = ()


# ---- the randomised OPTIONS campaign against the reference binary (tests/golden/options_ref.npz, tests/tools/gen_golden_options_ref.py) -----------------
OPTIONS_NAMES = ("calib_p3p", "calib_shift", "shared", "varying")
OPTIONS_KINDS = {"calib_p3p": (0, False, None), "calib_shift": (0, True, None), "shared": (1, False, "shared"), "varying": (2, False, "varying")}
OPTIONS_COLS = ("n", "outlier_frac", "noise_px", "max_epipolar_error", "max_reproj_error", "weight_sampson", "seed", "max_iterations", "min_iterations",
                "loss_type", "loss_scale", "bundle_max_iterations", "success_prob", "dyn_num_trials_mult", "gradient_tol", "step_tol", "initial_lambda",
                "min_lambda", "max_lambda", "f1", "f2", "ppx", "ppy", "pinhole2")
OPTIONS_FIRST = 30000
# oracle - reference in the LO count on the 384 cases (everything else identical there): the solver classes of DESIGN.md §5
# (shared 16: the reference's action-matrix solver returns the real part of a complex root pair 0.4326 +- 0.0321i as a double root at iteration 1)
OPTIONS_LO_DEVIATIONS = {"calib_p3p": {}, "calib_shift": {67: +1}, "shared": {16: +3}, "varying": {58: +1, 69: -1, 87: +1}}
# HIP path - oracle in the LO count (score ties decided by the last bits)
OPTIONS_GPU_MINUS_ORACLE_LO = {"varying": {58: -1, 69: +1}}  # (= the reference on both)
# another RANSAC winner than the reference's (230 against 227 inliers, 7 mask bits, model 2.7e-3; the shift solver's missed roots): the HIP path must
# equal the ORACLE there
OPTIONS_OTHER_WINNER = {"calib_shift": (23,)}
OPTIONS_MODEL_DEVIATIONS = {}


def options_pair(name, j, row):
    """inputs of case j of the options campaign (row = its line of the case table)"""
    from mdrp_amd import synth
    kind, es, rf = OPTIONS_KINDS[name]
    cams = dict(f1=float(row[19]), f2=float(row[20]), pp=(float(row[21]), float(row[22]))) if kind == 0 else {}  # calibrated: cameras of the case
    return synth.make_pair(OPTIONS_FIRST + j, int(row[0]), noise_px=float(row[2]), depth_noise=0.02 if row[2] > 0 else 0.0, outlier_frac=float(row[1]),
                           random_focal=rf, shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0, **cams)


def options_cameras(row):
    """(model_id, params) of the two cameras of a calibrated case: SIMPLE_PINHOLE, and PINHOLE with fx != fy for the second one when row[23]"""
    c1 = (0, [float(row[19]), float(row[21]), float(row[22])])
    c2 = (1, [float(row[20]) * 1.01, float(row[20]) * 0.99, float(row[21]), float(row[22])]) if row[23] else (0, [float(row[20]), float(row[21]), float(row[22])])
    return c1, c2


def options_dicts(row, es):
    """(RansacOptions, BundleOptions) of a case as keyword dicts (loss_type numeric)"""
    ro = dict(max_iterations=int(row[7]), min_iterations=int(row[8]), dyn_num_trials_mult=float(row[13]), success_prob=float(row[12]),
              max_reproj_error=float(row[4]), max_epipolar_error=float(row[3]), seed=int(row[6]), estimate_shift=es, weight_sampson=float(row[5]))
    bo = dict(max_iterations=int(row[11]), loss_type=int(row[9]), loss_scale=float(row[10]), gradient_tol=float(row[14]), step_tol=float(row[15]),
              initial_lambda=float(row[16]), min_lambda=float(row[17]), max_lambda=float(row[18]))
    return ro, bo


def input_digest(p):
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    for key in ("x1", "x2", "d1", "d2"):
        h.update(np.ascontiguousarray(p[key], dtype=np.float64).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def refine_ws_weights(i, n):
    """per-correspondence weights of the weighted cases of tests/golden/refine_ws.npz (problem i of refine.npz)"""
    import numpy as np
    return np.random.default_rng(4200 + i).uniform(0.2, 1.5, n)


# ---- the same campaign for the comparison rows (tests/golden/options_ref_classic.npz, tests/tools/gen_golden_options_ref_classic.py) ---------------------
CLASSIC_OPTIONS_KINDS = {"relpose_5pt": 3, "shared_6pt": 4, "fundamental_7pt": 5}
CLASSIC_OPTIONS_COLS = ("n", "outlier_frac", "noise_px", "max_epipolar_error", "seed", "max_iterations", "min_iterations", "loss_type", "loss_scale",
                        "bundle_max_iterations", "f1", "f2", "ppx", "ppy", "pinhole2")
CLASSIC_OPTIONS_FIRST = 40000


def classic_options_pair(name, j, row):
    """inputs of case j: 5-point: two cameras of their own focal length (the second one PINHOLE with fx != fy when row[14]) and a principal point;
    6-point: one shared focal length drawn by synth and the principal point handed to the estimator; 7-point: pixels as they are"""
    from mdrp_amd import synth
    kind = CLASSIC_OPTIONS_KINDS[name]
    kw = dict(noise_px=float(row[2]), depth_noise=0.0, outlier_frac=float(row[1]), pp=(float(row[12]), float(row[13])))
    if kind == 4:
        return synth.make_pair(CLASSIC_OPTIONS_FIRST + j, int(row[0]), random_focal="shared", **kw)
    return synth.make_pair(CLASSIC_OPTIONS_FIRST + j, int(row[0]), f1=float(row[10]), f2=float(row[11]), **kw)


def classic_options_cameras(row):
    """(model_id, params) of the two cameras of a 5-point case"""
    c1 = (0, [float(row[10]), float(row[12]), float(row[13])])
    c2 = (1, [float(row[11]) * 1.01, float(row[11]) * 0.99, float(row[12]), float(row[13])]) if row[14] else (0, [float(row[11]), float(row[12]), float(row[13])])
    return c1, c2


# oracle vs reference on the 3 x 64 cases: identical (iterations, inliers, mask, model 1e-6, LO count) on all 5- and 7-point cases; 6-point: case 6 ends
# on another RANSAC winner (620 vs 619 inliers, 35 mask bits: a 6-point solution-set difference), cases 19 and 51 differ by one LO (oracle - reference)
CLASSIC_OPTIONS_OTHER_WINNER = {"shared_6pt": (6,)}
CLASSIC_OPTIONS_LO_DEVIATIONS = {"shared_6pt": {19: -1, 51: -1}}


# ---- options at the edges of their ranges against the reference binary (tests/golden/edge_options_ref.npz, tests/tools/gen_golden_edge_options_ref.py) -----
EDGE_RANSAC = [dict(max_reproj_error=0.0), dict(weight_sampson=0.0), dict(weight_sampson=-1.0), dict(max_iterations=100, min_iterations=1000),
               dict(max_iterations=0, min_iterations=0), dict(max_iterations=1, min_iterations=0), dict(max_iterations=2, min_iterations=5),
               dict(max_iterations=5000, min_iterations=0), dict(success_prob=1.0, max_iterations=3000, min_iterations=100),
               dict(success_prob=0.0, max_iterations=3000, min_iterations=100), dict(dyn_num_trials_mult=0.0, max_iterations=3000, min_iterations=100),
               dict(max_epipolar_error=0.0), dict(max_epipolar_error=1e-3), dict(max_epipolar_error=100.0), dict(seed=2 ** 40 + 7),
               # calibrated P3P: the only sample(s) of the run are ones on which the reference's p3p() returns NaN poses -> the answer is a NaN pose
               dict(max_iterations=1, min_iterations=0, seed=12), dict(max_iterations=2, min_iterations=0, seed=171)]
EDGE_BUNDLE = [dict(loss_scale=0.0), dict(max_iterations=1), dict(initial_lambda=0.0), dict(min_lambda=1.0, max_lambda=1.0), dict(gradient_tol=1.0), dict(step_tol=1.0),
               dict(loss_type=0, max_iterations=3), dict(loss_type=5, max_iterations=200)]


def edge_cases():
    """[(RansacOptions kwargs, BundleOptions kwargs)] — one setting moved to an edge per case, the rest at the reference's own values"""
    base_r = dict(max_iterations=1000, min_iterations=1000, max_epipolar_error=2.0, max_reproj_error=16.0, seed=3, weight_sampson=1.0)
    base_b = dict(max_iterations=100, loss_type=4, loss_scale=1.0, gradient_tol=1e-10)
    return [(dict(base_r, **e), dict(base_b)) for e in EDGE_RANSAC] + [(dict(base_r), dict(base_b, **e)) for e in EDGE_BUNDLE]


def edge_pair(name):
    from mdrp_amd import synth
    kind, es, rf = OPTIONS_KINDS[name]
    return synth.make_pair(51000, 300, noise_px=0.5, depth_noise=0.02, outlier_frac=0.3, random_focal=rf, shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0)


def classic_edge_cases():
    """edge_cases() without the monodepth-only options"""
    out = []
    for rod, bod in edge_cases():
        if rod.get("max_reproj_error") != 16.0 or rod.get("weight_sampson") != 1.0:
            continue
        out.append(({k: v for k, v in rod.items() if k not in ("max_reproj_error", "weight_sampson")}, bod))
    return out


def classic_edge_pair(name):
    from mdrp_amd import synth
    return synth.make_pair(52000, 300, noise_px=0.5, depth_noise=0.0, outlier_frac=0.3, random_focal="shared" if CLASSIC_OPTIONS_KINDS[name] == 4 else None)


# edge cases whose winner is a tie: threshold 0 (every model scores 0: the first one scored stays, i.e. the solvers' solution ORDER decides; 6-point) and
# oracle - reference in the LO count where only that differs (5-point at a threshold of 1e-3 px: 7 inliers)
CLASSIC_EDGE_TIES = {"shared_6pt": ({"max_epipolar_error": 0.0},)}
CLASSIC_EDGE_LO_DEVIATIONS = {"relpose_5pt": {"max_epipolar_error=0.001": -1}}


def same_model(m, ref, tol=1e-6):
    """model_diff < tol, or — where the reference's answer has NaN / inf components — the same non-finite pattern and the finite rest equal"""
    import numpy as np
    m, ref = np.asarray(m, float), np.asarray(ref, float)
    if not (np.isfinite(ref).all() and np.isfinite(m).all()):
        # NaN in exactly the same components, infinities equal with their sign (models of corrupted inputs carry inf focals on both sides:
        # inf - inf is NaN in model_diff, which propagates it since round 6), the finite rest equal
        fin = np.isfinite(ref)
        return bool(np.array_equal(np.isnan(m), np.isnan(ref)) and np.array_equal(np.isfinite(m), fin) and np.array_equal(m[np.isinf(ref)], ref[np.isinf(ref)])
                    and np.allclose(m[fin], ref[fin], rtol=1e-9, atol=1e-12))
    return bool(model_diff(m, ref) < tol)


# ---- corrupted inputs against the reference binary (tests/golden/bad_inputs_ref.npz, tests/tools/gen_golden_bad_inputs_ref.py) ----------------------------
BAD_INPUT_MODES = ("zero_depth", "neg_depth", "nan_depth", "nan_depth2", "nan_point", "inf_point", "dup_points")


def bad_input_pair(name, mode):
    """N = 300 pair of the estimator with a tenth of its correspondences corrupted: depths 0 / negated / NaN (image 1, image 2), NaN / inf coordinates
    (5 points), 30 identical correspondences"""
    import numpy as np
    from mdrp_amd import synth
    kind, es, rf = OPTIONS_KINDS[name]
    p0 = synth.make_pair(53000, 300, noise_px=0.5, depth_noise=0.02, outlier_frac=0.3, random_focal=rf, shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0)
    p = {k: np.array(p0[k], copy=True) for k in ("x1", "x2", "d1", "d2")}
    idx = np.random.default_rng(5).choice(300, 30, replace=False)
    if mode == "zero_depth": p["d1"][idx] = 0.0
    elif mode == "neg_depth": p["d2"][idx] = -p["d2"][idx]
    elif mode == "nan_depth": p["d1"][idx[:5]] = np.nan
    elif mode == "nan_depth2": p["d2"][idx[:5]] = np.nan
    elif mode == "nan_point": p["x2"][idx[:5], 0] = np.nan
    elif mode == "inf_point": p["x1"][idx[:5], 1] = np.inf
    elif mode == "dup_points":
        for k in ("x1", "x2", "d1", "d2"): p[k][idx] = p[k][idx[0]]
    return p


# not compared: shared focal with inf coordinates — the normalisation scale is inf, every point becomes 0 (or NaN), the threshold 0; the reference's
# generated solver returns a NaN model for the all-zero samples (which then is the answer: any score is 0), ours returns none (identity, f = inf)
BAD_INPUT_SKIP = {("shared", "inf_point")}
BAD_INPUT_LO_DEVIATIONS = {}


# ---- degenerate geometry against the reference binary (tests/golden/degenerate_ref.npz, tests/tools/gen_golden_degenerate_ref.py) ---------------------------
DEGENERATE_MODES = ("pure_rotation", "planar", "tiny_baseline", "forward")
DEGENERATE_SEEDS = 6
# shared focal under pure rotation: the focal length is unobservable, the refinements stop 5e-6 apart (iterations, inliers, mask, LO count identical)
DEGENERATE_MODEL_TOL = {("shared", "pure_rotation", 4): 1e-5}
# HIP path only: with the shift solver under pure rotation the two depth shifts are unobservable (they end at 16.6 and 10.4 here) and the last LM steps
# amplify the summation order (iterations, inliers, mask identical)
DEGENERATE_GPU_MODEL_TOL = {("calib_shift", "pure_rotation", 5): 1e-3}


def degenerate_pair(name, mode, seed):
    """N = 400 correspondences, 25 % outliers, 0.5 px noise, of a scene that is degenerate for epipolar geometry: pure rotation (t = 0), a plane,
    a baseline of 1e-4 scene units, motion along the optical axis"""
    import numpy as np
    from mdrp_amd import synth
    kind, es, rf = OPTIONS_KINDS[name]
    rng = np.random.default_rng(seed)
    n = 400
    f1 = f2 = 800.0
    if rf == "shared": f1 = f2 = rng.uniform(400, 1500)
    if rf == "varying": f1, f2 = rng.uniform(400, 1500, 2)
    R = synth.rodrigues(rng.normal(0, 0.2, 3)); t = rng.normal(0, 0.5, 3); scale = rng.uniform(0.5, 2)
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(3, 8, n)], 1)
    if mode == "pure_rotation": t = np.zeros(3)
    if mode == "planar": X[:, 2] = 5.0 + 0.3 * X[:, 0]
    if mode == "tiny_baseline": t = t * 1e-4
    if mode == "forward": t = np.array([0, 0, 0.7])
    X2 = X @ R.T + t
    x1 = f1 * X[:, :2] / X[:, 2:3] + rng.normal(0, 0.5, (n, 2)); x2 = f2 * X2[:, :2] / X2[:, 2:3] + rng.normal(0, 0.5, (n, 2))
    d1 = X[:, 2] * (1 + rng.normal(0, 0.02, n)) - (0.2 if es else 0); d2 = X2[:, 2] / scale * (1 + rng.normal(0, 0.02, n)) - (-0.1 if es else 0)
    o = rng.choice(n, n // 4, replace=False); x2[o] = rng.uniform(-600, 600, (len(o), 2))
    return {"x1": x1, "x2": x2, "d1": d1, "d2": d2}
