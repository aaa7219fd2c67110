"""Parity of the HIP path (through the C-ABI, mdrp_amd/_capi.py) against the CPU oracle and the golden vectors
captured from the reference binary.  Needs an MI355X:  pytest -m gpu."""
import numpy as np
import pytest

from helpers import (KNOWN_LO_COUNT_DEVIATIONS, OPTIONS_GPU_MINUS_ORACLE_LO, OPTIONS_KINDS, OPTIONS_LO_DEVIATIONS, OPTIONS_MODEL_DEVIATIONS, OPTIONS_NAMES,
                     OPTIONS_OTHER_WINNER, REFERENCE_NAN_SOLUTIONS, match_solution_sets, model_diff, options_cameras, options_dicts, options_pair, widen)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle():
    from mdrp_amd import _capi
    return _capi.default_handle(0)


@pytest.fixture(scope="module")
def capi():
    from mdrp_amd import _capi
    return _capi


@pytest.fixture(scope="module")
def po():
    from oracle import pyorc
    return pyorc


def _flat(capi, models):
    return [capi.model_to_array(m) for m in models]


# ---------------------------------------------------------------------------------------------- solvers
@pytest.mark.parametrize("kind", ["p3p", "calib_shift", "shared", "varying"])
def test_solvers_vs_oracle_and_reference(handle, capi, po, golden, kind):
    g = golden("solvers")
    src = "calib_shift" if kind == "p3p" else kind
    solver = {"p3p": 0, "calib_shift": 1, "shared": 2, "varying": 3}[kind]
    ofn = {"p3p": po.solver_calib_p3p, "calib_shift": po.solver_calib_shift, "shared": po.solver_shared, "varying": po.solver_varying}[kind]
    x1, x2, d1, d2 = g[f"{src}_x1"], g[f"{src}_x2"], g[f"{src}_d1"], g[f"{src}_d2"]
    out, n = handle.solver_batch(solver, x1, x2, d1, d2)
    agree_ref = checked_ref = 0
    for i in range(len(n)):
        mine = _flat(capi, out[i, : n[i]])
        ref = list(ofn(x1[i], x2[i], d1[i], d2[i]))
        assert match_solution_sets(ref, mine, 1e-7), (kind, i)
        if kind != "p3p":  # golden reference solution sets exist for the three monodepth solvers
            nref = int(g[f"{kind}_n"][i])
            r = g[f"{kind}_sols"][i][:nref]
            if not np.isnan(r).any():
                checked_ref += 1
                agree_ref += match_solution_sets([widen(v) for v in r], mine, 1e-6)
            else:
                assert i in REFERENCE_NAN_SOLUTIONS[kind], (kind, i)
    if kind != "p3p":  # every golden problem on which the reference itself returns finite models
        assert checked_ref == len(n) - len(REFERENCE_NAN_SOLUTIONS[kind]) and agree_ref == checked_ref, (agree_ref, checked_ref)


def test_reference_nan_p3p_samples_never_have_a_real_pose_on_the_device(handle, capi, po):
    """k_solve evaluates the reference's NaN-pose predicate (p3p_reference_nan) only on lanes whose own P3P found no pose; the oracle — like the
    reference — evaluates it first and lets it override real roots.  The two agree as long as `predicate true` implies `the device solver finds nothing`.
    Checked here on 60 000 minimal samples drawn the way the estimator draws them (headline generator, outliers and noise included; degenerate and
    near-degenerate triples from tiny N): the oracle's predicate (orc_p3p_reference_nan, sqrt-and-divide arithmetic) on every sample against the device
    solver's solution count through mdrp_solver_batch — no sample may have both, and the predicate fires often enough for the check to mean something."""
    import ctypes as C
    from mdrp_amd import synth
    lib = po.lib()
    lib.orc_p3p_reference_nan.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.orc_p3p_reference_nan.restype = C.c_int
    rng = np.random.default_rng(2024)
    X1, X2, D1, D2 = [], [], [], []
    for k, n in enumerate((2000, 2000, 400, 60, 12, 5)):
        b = synth.make_batch(31000 + 100 * k, 20, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
        for j in range(20):
            idx = np.stack([rng.choice(n, 3, replace=False) for _ in range(500)])
            X1.append(b["x1"][j][idx]); X2.append(b["x2"][j][idx]); D1.append(b["d1"][j][idx]); D2.append(b["d2"][j][idx])
    x1 = np.concatenate(X1) / 800.0; x2 = np.concatenate(X2) / 800.0; d1 = np.concatenate(D1); d2 = np.concatenate(D2)
    S = len(d1)
    x1h = np.concatenate([x1, np.ones((S, 3, 1))], axis=2); x2h = np.concatenate([x2, np.ones((S, 3, 1))], axis=2)
    out, n_dev = handle.solver_batch(0, x1h, x2h, d1, d2)
    xb = np.ascontiguousarray(x2h / np.linalg.norm(x2h, axis=2, keepdims=True))  # unit bearings of image 2 (sqrt and divide, as the oracle's caller forms them)
    Xp = np.ascontiguousarray(x1h * d1[:, :, None])                                # back-projected points of image 1
    dp = C.POINTER(C.c_double)
    pred = np.array([lib.orc_p3p_reference_nan(xb[i].ctypes.data_as(dp), Xp[i].ctypes.data_as(dp)) for i in range(S)], dtype=bool)
    both = np.nonzero(pred & (n_dev > 0))[0]
    assert S == 60000 and pred.sum() > 300, (S, int(pred.sum()))
    assert len(both) == 0, (len(both), both[:10], n_dev[both[:10]])
    print(f"reference-NaN P3P samples: {int(pred.sum())} of {S}; none of them has a real pose on the device (solutions on the others: mean {n_dev[~pred].mean():.2f})")


# ---------------------------------------------------------------------------------------------- scoring sweep
def test_score_sweep_vs_reference_golden(handle, capi, po, golden):
    g = golden("scoring")
    for i in range(6):
        x1, x2, m, thr = g[f"x1_{i}"], g[f"x2_{i}"], g[f"model_{i}"], float(g[f"thr_{i}"])
        models = capi.array_to_models(np.stack([m, m]))
        s, c = handle.score_models(capi.CALIB, models, x1, x2, thr)
        assert c[0] == int(g[f"pose_cnt_{i}"]) and c[1] == c[0]
        assert s[0] == pytest.approx(float(g[f"pose_score_{i}"]), rel=1e-12)
        s, c = handle.score_models(capi.VARYING_FOCAL, models, x1, x2, thr)
        assert c[0] == int(g[f"F_cnt_{i}"])
        assert s[0] == pytest.approx(float(g[f"F_score_{i}"]), rel=1e-12)


@pytest.mark.parametrize("n", [3, 64, 2000, 2049, 5000])
def test_score_sweep_many_models_vs_oracle(handle, capi, po, n):
    """ragged sizes around the LDS tile (2048) and the largest BASELINE size; 700 hypotheses (> one workgroup)"""
    from mdrp_amd import synth
    rng = np.random.default_rng(n)
    p = synth.make_pair(n, n, noise_px=1.0, outlier_frac=0.4)
    x1, x2 = p["x1"] / 800.0, p["x2"] / 800.0
    ms = []
    for k in range(700):
        m = po.new_model()
        if k % 3 == 0:
            R = p["R"] @ synth.rodrigues(rng.normal(0, 0.01, 3)); t = p["t"] + rng.normal(0, 0.01, 3)
        else:
            R = synth.rodrigues(rng.normal(0, 1.0, 3)); t = rng.normal(size=3)
        q = np.zeros(4); po.lib().orc_rotmat_to_quat(np.ascontiguousarray(R.reshape(-1)).ctypes.data_as(po._dp), q.ctypes.data_as(po._dp))
        m[:4] = q; m[4:7] = t; m[10] = 1.0 + 0.3 * (k % 5); m[11] = 0.8 + 0.1 * (k % 7)
        ms.append(m)
    thr = (2.0 / 800.0) ** 2
    models = capi.array_to_models(np.stack(ms))
    for kind in (capi.CALIB, capi.VARYING_FOCAL):
        s, c = handle.score_models(kind, models, x1, x2, thr)
        for k in range(0, 700, 7):
            if kind == capi.CALIB:
                so, co = po.msac_pose(ms[k], x1, x2, thr)
            else:
                so, co = po.msac_F(po.fundamental(ms[k]), x1, x2, thr)
            assert c[k] == co, (n, kind, k)
            assert s[k] == pytest.approx(so, rel=1e-10)


# ---------------------------------------------------------------------------------------------- MFMA candidate count
@pytest.mark.parametrize("n", [3, 15, 16, 17, 2000, 2049, 5000])
def test_count_candidates_is_conservative(handle, capi, po, n):
    """k_count (bf16-split MFMA filter) must never undercount: for every model, candidates >= correspondences that pass the
    exact fp64 Sampson test (a superset of the inliers, cheirality aside), at group-ragged sizes; garbage models must stay
    sparse (that is what retires them), NaN / inf models must keep everything."""
    from mdrp_amd import synth
    rng = np.random.default_rng(1000 + n)
    p = synth.make_pair(n, n, noise_px=1.0, outlier_frac=0.4 if n > 20 else 0.0)
    x1, x2 = p["x1"] / 800.0, p["x2"] / 800.0
    ms = []
    for k in range(600):
        m = po.new_model()
        if k % 3 == 0:
            R = p["R"] @ synth.rodrigues(rng.normal(0, 0.003 * (k % 7), 3)); t = p["t"] + rng.normal(0, 0.003 * (k % 5), 3)
        else:
            R = synth.rodrigues(rng.normal(0, 1.0, 3)); t = rng.normal(size=3) * 10.0 ** rng.integers(-3, 3)
        q = np.zeros(4); po.lib().orc_rotmat_to_quat(np.ascontiguousarray(R.reshape(-1)).ctypes.data_as(po._dp), q.ctypes.data_as(po._dp))
        m[:4] = q; m[4:7] = t; m[10] = 1.0 + 0.3 * (k % 5); m[11] = 0.8 + 0.1 * (k % 7)
        ms.append(m)
    ms[7][4] = np.nan; ms[8][5] = np.inf; ms[9][:4] = 0.0
    thr = (2.0 / 800.0) ** 2
    models = capi.array_to_models(np.stack(ms))
    h1 = np.c_[x1, np.ones(n)]; h2 = np.c_[x2, np.ones(n)]
    for kind in (capi.CALIB, capi.VARYING_FOCAL):
        cand = handle.count_candidates(kind, models, x1, x2, thr)
        assert cand[7] == n and cand[8] == n
        sparse = []
        for k in range(600):
            if k in (7, 8, 9):
                continue
            E = po.essential(ms[k]) if kind == capi.CALIB else po.fundamental(ms[k])
            Ex1 = h1 @ E.T; Etx2 = h2 @ E
            C = np.sum(h2 * Ex1, axis=1)
            den = Ex1[:, 0] ** 2 + Ex1[:, 1] ** 2 + Etx2[:, 0] ** 2 + Etx2[:, 1] ** 2
            exact = int(np.sum(C * C < thr * den))
            assert cand[k] >= exact, (n, kind, k, cand[k], exact)
            assert cand[k] <= n
            if k % 3:
                sparse.append(cand[k] / n)
        if n >= 2000:
            assert np.median(sparse) < 0.15, np.median(sparse)


def test_count_filter_error_bound_on_exact_correspondences(handle, capi, po):
    """The analytical error bound of the bf16-split contraction, measured on the real instruction: with a ZERO threshold the
    filter keeps a correspondence only if |C_mfma| <= KAPPA * M.  Noise-free correspondences of the true model have C = 0
    in exact arithmetic, so every one of them must be kept: 48 pairs x 2000 correspondences x 2 estimators."""
    from mdrp_amd import synth
    for i in range(48):
        rf = "varying" if i % 2 else None
        p = synth.make_pair(7000 + i, 2000, noise_px=0.0, depth_noise=0.0, random_focal=rf)
        f1, f2 = (p["f1"], p["f2"]) if rf else (800.0, 800.0)
        sc = 800.0
        x1, x2 = p["x1"] / (f1 if not rf else sc), p["x2"] / (f2 if not rf else sc)
        m = po.new_model()
        q = np.zeros(4); po.lib().orc_rotmat_to_quat(np.ascontiguousarray(p["R"].reshape(-1)).ctypes.data_as(po._dp), q.ctypes.data_as(po._dp))
        m[:4] = q; m[4:7] = p["t"] * (1.0 + i)   # the scale of t does not matter to E x = 0
        if rf:
            m[10], m[11] = f1 / sc, f2 / sc  # F = diag(1,1,f2') E diag(1,1,f1') with x in pixels / sc
        cand = handle.count_candidates(capi.VARYING_FOCAL if rf else capi.CALIB, capi.array_to_models(m[None]), x1, x2, 0.0)
        assert cand[0] == 2000, (i, cand[0])


# ---------------------------------------------------------------------------------------------- refinement
def test_refine_vs_reference_golden(handle, capi, golden):
    g = golden("refine")
    for ci, case in enumerate(g["cases"]):
        i, kind, es, lt, its, thr = int(case[0]), int(case[1]), int(case[2]), int(case[3]), int(case[4]), case[5]
        if its == 0:
            continue
        bo = capi.bundle_opt_from_dict({"max_iterations": its, "loss_type": lt, "loss_scale": thr, "gradient_tol": 1e-10})
        m, cost = handle.refine_models(kind, capi.array_to_models(g[f"model_{i}"]), g[f"x1_{i}"], g[f"x2_{i}"], g[f"d1_{i}"],
                                       g[f"d2_{i}"], 1 / 64.0, 1.0, bo, es)
        ref = g[f"out_{ci}"]
        assert model_diff(capi.model_to_array(m[0]), ref[:12]) < 1e-6, case
        assert cost[0] == pytest.approx(ref[14], rel=1e-8, abs=1e-18)


def test_refine_weight_sampson_vs_reference_golden(handle, capi, golden):
    """tests/golden/refine_ws.npz: the reference's three refiners at ws in {0.3 ... 3} — cost ws rho(r^2), normal equations ws^2 w(.), the loss weight at
    r^2 (calibrated) or ws r^2 (focal) — on the 648 unweighted cases (the C ABI's refine entry point has no per-correspondence weights, like the
    estimators' calls of it), model to 1e-6 and final cost to 1e-8."""
    g, inp = golden("refine_ws"), golden("refine")
    ran = 0
    for case, ref in zip(g["cases"], g["out"]):
        i, kind, es, ws, lt, its, weighted, thr = int(case[0]), int(case[1]), int(case[2]), case[3], int(case[4]), int(case[5]), int(case[6]), case[7]
        if weighted:
            continue
        bo = capi.bundle_opt_from_dict({"max_iterations": its, "loss_type": lt, "loss_scale": thr, "gradient_tol": 1e-10})
        m, cost = handle.refine_models(kind, capi.array_to_models(inp[f"model_{i}"]), inp[f"x1_{i}"], inp[f"x2_{i}"], inp[f"d1_{i}"], inp[f"d2_{i}"],
                                       1 / 64.0, float(ws), bo, es)
        assert model_diff(capi.model_to_array(m[0]), ref[:12]) < 1e-6, case
        assert cost[0] == pytest.approx(ref[14], rel=1e-8, abs=1e-18), case
        ran += 1
    assert ran == 648


# ---------------------------------------------------------------------------------------------- full estimators
@pytest.mark.parametrize("name", list(OPTIONS_NAMES))
def test_randomised_options_vs_reference_fixture(golden, po, name):
    """tests/golden/options_ref.npz through the drop-in module's single-pair entry points (option DICTS and Camera dicts as the reference's scripts pass
    them): 96 cases per estimator with size, outlier share, noise, both thresholds, the Sampson weight, seed, fixed / dynamic budget, stopping rule, every
    BundleOptions field and (calibrated) both cameras drawn at random.  Iterations, inlier count and mask identical to the REFERENCE BINARY, model within
    1e-6; LO count = the oracle's (exact lists: helpers.OPTIONS_LO_DEVIATIONS, 8 cases oracle != reference; OPTIONS_GPU_MINUS_ORACLE_LO); on the one
    case where the oracle ends on another winner than the reference the HIP path must equal the oracle, run here as the checker."""
    import mdrp_amd.poselib as poselib
    from mdrp_amd import _capi
    g = golden("options_ref")
    kind, es, rf = OPTIONS_KINDS[name]
    loss_name = {v: k for k, v in _capi.LOSS_TYPES.items()}
    lo_dev, model_dev = OPTIONS_LO_DEVIATIONS.get(name, {}), OPTIONS_MODEL_DEVIATIONS.get(name, {})
    for j, row in enumerate(g["cases"]):
        n = int(row[0])
        p = options_pair(name, j, row)
        rod, bod = options_dicts(row, es)
        ro = {("monodepth_" + k if k in ("estimate_shift", "weight_sampson") else k): v for k, v in rod.items()}
        bo = dict(bod, loss_type=loss_name[bod["loss_type"]])
        if kind == 0:
            c1, c2 = options_cameras(row)
            cams = [{"model": "PINHOLE" if c[0] else "SIMPLE_PINHOLE", "width": 1600, "height": 1200, "params": c[1]} for c in (c1, c2)]
            geom, info = poselib.estimate_monodepth_relative_pose(p["x1"], p["x2"], p["d1"], p["d2"], cams[0], cams[1], ro, bo)
            m = np.r_[geom.pose.q, geom.pose.t, geom.scale, geom.shift1, geom.shift2, 1.0, 1.0]
        else:
            fn = poselib.estimate_monodepth_shared_focal_relative_pose if kind == 1 else poselib.estimate_monodepth_varying_focal_relative_pose
            pair, info = fn(p["x1"], p["x2"], p["d1"], p["d2"], ro, bo)
            geom = pair.geometry
            m = np.r_[geom.pose.q, geom.pose.t, geom.scale, geom.shift1, geom.shift2, pair.camera1.focal(), pair.camera2.focal()]
        ist, ref_mask, ref_model = g[f"{name}_istats"][j], np.unpackbits(g[f"{name}_mask"][j])[:n], g[f"{name}_model"][j]
        ref_lo = int(ist[0]) + lo_dev.get(j, 0)
        if j in OPTIONS_OTHER_WINNER.get(name, ()):
            c1, c2 = options_cameras(row)
            ref_model, st, ref_mask = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], po.ransac_opt(**rod), po.bundle_opt(**bod),
                                                  po.cam_flat(*c1) if kind == 0 else None, po.cam_flat(*c2) if kind == 0 else None)
            ist, ref_lo = (st.refinements, st.iterations, st.num_inliers), st.refinements
        assert (info["iterations"], info["num_inliers"]) == (int(ist[1]), int(ist[2])), (name, j, info["iterations"], info["num_inliers"], ist)
        assert (np.asarray(info["inliers"], dtype=np.uint8) == ref_mask).all(), (name, j)
        assert model_diff(m, ref_model) < model_dev.get(j, 1e-6), (name, j, model_diff(m, ref_model))
        assert info["refinements"] == ref_lo + OPTIONS_GPU_MINUS_ORACLE_LO.get(name, {}).get(j, 0), (name, j, info["refinements"], int(ist[0]))


@pytest.mark.parametrize("name", list(OPTIONS_NAMES))
def test_edge_options_vs_reference_fixture(golden, name):
    """tests/golden/edge_options_ref.npz through the drop-in module: one option at an edge of its range per case — max_iterations 0 / 1 / below
    min_iterations, success_prob 0 / 1, dyn_num_trials_mult 0, thresholds 0 / 1e-3 / 100 px, reprojection off, weight_sampson 0 / negative, a 41-bit
    seed, loss_scale 0, pinned damping, tolerances of 1.  Stats, mask and model identical to the REFERENCE BINARY on all 4 x 25 cases (two of them, for the calibrated P3P estimator, runs whose only samples make the reference's p3p() return NaN poses: the answer is that NaN pose)."""
    import mdrp_amd.poselib as poselib
    from mdrp_amd import _capi
    from helpers import edge_cases, edge_pair, same_model
    g = golden("edge_options_ref")
    kind, es, rf = OPTIONS_KINDS[name]
    p = edge_pair(name)
    cam = {"model": "SIMPLE_PINHOLE", "width": 1600, "height": 1200, "params": [800.0, 0.0, 0.0]}
    loss_name = {v: k for k, v in _capi.LOSS_TYPES.items()}
    for j, (rod, bod) in enumerate(edge_cases()):
        ro = {("monodepth_" + k if k == "weight_sampson" else k): v for k, v in rod.items()}
        ro["monodepth_estimate_shift"] = es
        bo = dict(bod, loss_type=loss_name[bod["loss_type"]])
        if kind == 0:
            geom, info = poselib.estimate_monodepth_relative_pose(p["x1"], p["x2"], p["d1"], p["d2"], cam, cam, ro, bo)
            m = np.r_[geom.pose.q, geom.pose.t, geom.scale, geom.shift1, geom.shift2, 1.0, 1.0]
        else:
            fn = poselib.estimate_monodepth_shared_focal_relative_pose if kind == 1 else poselib.estimate_monodepth_varying_focal_relative_pose
            pair, info = fn(p["x1"], p["x2"], p["d1"], p["d2"], ro, bo)
            geom = pair.geometry
            m = np.r_[geom.pose.q, geom.pose.t, geom.scale, geom.shift1, geom.shift2, pair.camera1.focal(), pair.camera2.focal()]
        ref = g[f"{name}_stats"][j]
        assert (info["refinements"], info["iterations"], info["num_inliers"]) == tuple(int(v) for v in ref[:3]), (name, j, rod, bod, info["refinements"], info["iterations"], info["num_inliers"], ref)
        assert (np.asarray(info["inliers"], dtype=np.uint8) == np.unpackbits(g[f"{name}_mask"][j])[:300]).all(), (name, j)
        assert same_model(m, g[f"{name}_model"][j]), (name, j, rod, bod, m, g[f"{name}_model"][j])
        assert info["model_score"] == ref[4] or abs(info["model_score"] - ref[4]) <= 1e-9 * abs(ref[4]), (name, j)


@pytest.mark.parametrize("name", list(OPTIONS_NAMES))
def test_corrupted_inputs_vs_reference_fixture(handle, capi, golden, name):
    """tests/golden/bad_inputs_ref.npz: a tenth of the correspondences with zero / negative / NaN depths (either image), NaN / inf coordinates, or all
    identical — stats, mask and model identical to the REFERENCE BINARY on 27 of the 4 x 7 cases (the other one, garbage in both, is not compared: helpers.BAD_INPUT_SKIP): non-positive and NaN depths drop the reprojection terms, NaN / inf correspondences are never inliers, a NaN coordinate
    poisons the focal estimators' normalisation (0 inliers, NaN model)."""
    from helpers import BAD_INPUT_LO_DEVIATIONS, BAD_INPUT_MODES, BAD_INPUT_SKIP, bad_input_pair, same_model
    g = golden("bad_inputs_ref")
    kind, es, rf = OPTIONS_KINDS[name]
    cams = np.zeros(1, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    ro = capi.ransac_opt_from_dict({"max_iterations": 500, "min_iterations": 500, "max_epipolar_error": 2.0, "max_reproj_error": 16.0, "seed": 2, "monodepth_estimate_shift": es})
    bo = capi.bundle_opt_from_dict({"max_iterations": 100, "loss_type": "TRUNCATED_CAUCHY", "loss_scale": 1.0, "gradient_tol": 1e-10})
    for j, mode in enumerate(BAD_INPUT_MODES):
        if (name, mode) in BAD_INPUT_SKIP:
            continue
        p = bad_input_pair(name, mode)
        res, mask = handle.estimate_batch(kind, p["x1"][None], p["x2"][None], p["d1"][None], p["d2"][None], ro, bo, None, cams if kind == 0 else None, cams if kind == 0 else None)
        r, ref = res[0], g[f"{name}_stats"][j]
        assert (int(r["iterations"]), int(r["num_inliers"])) == (int(ref[1]), int(ref[2])), (name, mode, int(r["iterations"]), int(r["num_inliers"]), ref)
        assert int(r["refinements"]) - int(ref[0]) in (0, BAD_INPUT_LO_DEVIATIONS.get((name, mode), 0)), (name, mode, int(r["refinements"]), ref[0])
        assert (mask[0] == np.unpackbits(g[f"{name}_mask"][j])[:300]).all(), (name, mode)
        assert same_model(capi.model_to_array(r["model"]), g[f"{name}_model"][j]), (name, mode, capi.model_to_array(r["model"]), g[f"{name}_model"][j])


@pytest.mark.parametrize("name", list(OPTIONS_NAMES))
def test_degenerate_geometry_vs_reference_fixture(handle, capi, golden, name):
    """tests/golden/degenerate_ref.npz as ONE batched call per estimator: pure rotation, a planar scene, a baseline of 1e-4, motion along the optical axis
    (6 seeds each — the seed is an option of the call, so one call per seed over the four scene types): iterations, inlier count and mask identical to
    the REFERENCE BINARY on all 24 cases, model within 1e-6 (one enumerated: 1e-5), LO count equal or one apart (rounding ties on degenerate data)."""
    from helpers import DEGENERATE_GPU_MODEL_TOL, DEGENERATE_MODEL_TOL, DEGENERATE_MODES, DEGENERATE_SEEDS, degenerate_pair, same_model
    g = golden("degenerate_ref")
    kind, es, rf = OPTIONS_KINDS[name]
    B = len(DEGENERATE_MODES)
    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    bo = capi.bundle_opt_from_dict({"max_iterations": 100, "loss_type": "TRUNCATED_CAUCHY", "loss_scale": 1.0, "gradient_tol": 1e-10})
    lo_off = 0
    for seed in range(DEGENERATE_SEEDS):
        pairs = [degenerate_pair(name, mode, seed) for mode in DEGENERATE_MODES]
        ro = capi.ransac_opt_from_dict({"max_iterations": 1000, "min_iterations": 1000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0, "seed": seed, "monodepth_estimate_shift": es})
        res, mask = handle.estimate_batch(kind, np.stack([p["x1"] for p in pairs]), np.stack([p["x2"] for p in pairs]), np.stack([p["d1"] for p in pairs]),
                                          np.stack([p["d2"] for p in pairs]), ro, bo, None, cams if kind == 0 else None, cams if kind == 0 else None)
        for i, mode in enumerate(DEGENERATE_MODES):
            k = i * DEGENERATE_SEEDS + seed
            r, ref = res[i], g[f"{name}_stats"][k]
            assert (int(r["iterations"]), int(r["num_inliers"])) == (int(ref[1]), int(ref[2])), (name, mode, seed, int(r["num_inliers"]), ref)
            assert (mask[i] == np.unpackbits(g[f"{name}_mask"][k])[:400]).all(), (name, mode, seed)
            tol = DEGENERATE_GPU_MODEL_TOL.get((name, mode, seed), DEGENERATE_MODEL_TOL.get((name, mode, seed), 1e-6))
            assert same_model(capi.model_to_array(r["model"]), g[f"{name}_model"][k], tol), (name, mode, seed, model_diff(capi.model_to_array(r["model"]), g[f"{name}_model"][k]))
            assert abs(int(r["refinements"]) - int(ref[0])) <= 1, (name, mode, seed, int(r["refinements"]), ref[0])
            lo_off += int(r["refinements"]) != int(ref[0])
    assert lo_off <= 2, lo_off


def _run_estimate(capi, handle, kind, x1, x2, d1, d2, ro, bo, cam1=None, cam2=None):
    def camrec(c):
        r = np.zeros(1, dtype=capi.CAMERA_DTYPE)
        r["model_id"] = int(c[0]); p = np.zeros(4); p[: int(c[1])] = c[2:2 + int(c[1])]; r["params"] = p
        return r
    res, mask = handle.estimate_batch(kind, x1[None], x2[None], d1[None], d2[None], capi.ransac_opt_from_dict(ro),
                                      capi.bundle_opt_from_dict(bo), None, camrec(cam1) if cam1 is not None else None,
                                      camrec(cam2) if cam2 is not None else None)
    return res[0], mask[0]


def test_estimate_vs_reference_golden(handle, capi, golden):
    """all 18 golden cases: calibrated (P3P and shift), shared, varying; noise-free ones to 1e-6 (north_star),
    noisy ones must land on the same RANSAC trajectory up to rounding-level ties."""
    g = golden("estimate")
    same_traj = noisy = 0
    for case in g["cases"]:
        i, kind, es, noise, of, max_it, min_it, seed, lt = case
        i, kind, es = int(i), int(kind), int(es)
        ro = {"max_iterations": int(max_it), "min_iterations": int(min_it), "max_epipolar_error": 2.0, "max_reproj_error": 16.0,
              "seed": int(seed), "monodepth_estimate_shift": bool(es)}
        bo = {"loss_type": int(lt)}
        res, mask = _run_estimate(capi, handle, kind, g[f"x1_{i}"], g[f"x2_{i}"], g[f"d1_{i}"], g[f"d2_{i}"], ro, bo,
                                  g[f"cam1_{i}"] if kind == 0 else None, g[f"cam2_{i}"] if kind == 0 else None)
        ref_m, ref_st, ref_mask = g[f"model_{i}"], g[f"stats_{i}"], g[f"mask_{i}"]
        m = capi.model_to_array(res["model"])
        assert int(res["iterations"]) == int(ref_st[1]), case
        if noise == 0:
            assert model_diff(m, ref_m) < 1e-6, (case, model_diff(m, ref_m))
            assert int(res["num_inliers"]) == int(ref_st[2]) and (mask == ref_mask).all()
        else:
            noisy += 1
            ok = (int(res["num_inliers"]) == int(ref_st[2]) and int(res["refinements"]) == int(ref_st[0])
                  and (mask == ref_mask).all() and model_diff(m, ref_m) < 1e-6)
            same_traj += ok
    assert same_traj == noisy, (same_traj, noisy)  # every noisy golden case lands on the reference's exact trajectory


@pytest.mark.parametrize("kind,es,rf", [(0, False, None), (0, True, None), (1, False, "shared"), (2, False, "varying")])
def test_statistical_equivalence_noisy(handle, capi, po, kind, es, rf):
    """north_star: 'statistically equivalent inlier counts on noisy data'.  48 noisy pairs per estimator (N = 500, 35 %
    outliers, default dynamic stopping) against the CPU oracle: identical iteration counts, mean inlier count within
    0.5 %, and every pair on exactly the same trajectory (same LO count, inliers and mask)."""
    from mdrp_amd import synth
    B, N = 48, 500
    b = synth.make_batch(2000 + 100 * kind + int(es), B, N, noise_px=0.5, depth_noise=0.02, outlier_frac=0.35, random_focal=rf,
                         shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0)
    ro = {"max_epipolar_error": 2.0, "max_reproj_error": 16.0, "min_iterations": 300, "monodepth_estimate_shift": es}
    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    res, mask = handle.estimate_batch(kind, b["x1"], b["x2"], b["d1"], b["d2"], capi.ransac_opt_from_dict(ro),
                                      capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), None,
                                      cams if kind == 0 else None, cams if kind == 0 else None)
    oro = po.ransac_opt(max_epipolar_error=2.0, max_reproj_error=16.0, min_iterations=300, estimate_shift=es)
    cam = po.cam_flat(0, [800.0, 0, 0])
    same = 0
    inl_gpu, inl_cpu = [], []
    for i in range(B):
        m, st, mk = po.estimate(kind, b["x1"][i], b["x2"][i], b["d1"][i], b["d2"][i], oro, po.bundle_opt(loss_type=4),
                                cam if kind == 0 else None, cam if kind == 0 else None)
        inl_gpu.append(int(res[i]["num_inliers"])); inl_cpu.append(st.num_inliers)
        same += (int(res[i]["iterations"]) == st.iterations and int(res[i]["refinements"]) == st.refinements
                 and int(res[i]["num_inliers"]) == st.num_inliers and (mask[i] == mk).all())
    assert abs(np.mean(inl_gpu) - np.mean(inl_cpu)) <= 0.005 * np.mean(inl_cpu), (np.mean(inl_gpu), np.mean(inl_cpu))
    assert same == B, (same, B)


def test_full_size_vs_reference_golden(handle, capi, golden):
    """Every BASELINE.json shape at FULL size against the reference binary's own output (tests/golden/estimate_full.npz,
    generated by tests/tools/gen_golden.py): calibrated P3P (50 % and 0 % outliers) and shift solver, shared focal at
    N = 2000, varying focal with monodepth_estimate_shift=True at N = 5000, all at 10^4 iterations.  One batch per
    estimator so the chunked three-stream schedule with bail-out scoring runs as in bench.py.  Asserted per pair:
    iterations, inlier count, inlier mask, model <= 1e-6, score; LO count exact up to the two enumerated deviations."""
    from test_oracle_golden import full_size_cases
    g = golden("estimate_full")
    groups = {}
    for c in full_size_cases(g):
        groups.setdefault((c[1], c[2], c[3]), []).append(c)
    for (kind, es, n), cases in groups.items():
        B = len(cases)
        x1 = np.stack([g[f"x1_{c[0]}"] for c in cases]); x2 = np.stack([g[f"x2_{c[0]}"] for c in cases])
        d1 = np.stack([g[f"d1_{c[0]}"] for c in cases]); d2 = np.stack([g[f"d2_{c[0]}"] for c in cases])
        cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
        ro = {"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0,
              "monodepth_estimate_shift": bool(es)}
        res, mask = handle.estimate_batch(kind, x1, x2, d1, d2, capi.ransac_opt_from_dict(ro), capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}),
                                          None, cams if kind == 0 else None, cams if kind == 0 else None)
        for j, (i, _, _, _, ref_m, ref_st, ref_mask) in enumerate(cases):
            assert int(res[j]["iterations"]) == int(ref_st[1]) == 10000
            assert int(res[j]["num_inliers"]) == int(ref_st[2]), (i, int(res[j]["num_inliers"]), ref_st[2])
            assert (mask[j] == ref_mask).all(), i
            assert model_diff(capi.model_to_array(res[j]["model"]), ref_m) < 1e-6, (i, model_diff(capi.model_to_array(res[j]["model"]), ref_m))
            assert res[j]["model_score"] == pytest.approx(ref_st[4], rel=1e-9)
            assert res[j]["inlier_ratio"] == pytest.approx(ref_st[3], rel=1e-12)
            assert int(res[j]["refinements"]) == int(ref_st[0]) + KNOWN_LO_COUNT_DEVIATIONS.get(i, 0), (i, int(res[j]["refinements"]), ref_st[0])
            if kind != 0 or not es:
                assert res[j]["model"]["shift1"] == 0.0 and res[j]["model"]["shift2"] == 0.0  # the flag is ignored off the calibrated path


@pytest.mark.parametrize("kind,es,rf", [(0, False, None), (0, True, None), (1, False, "shared"), (2, False, "varying")])
def test_stress_grid_vs_oracle(handle, capi, po, kind, es, rf):
    """tests/tools/stress_parity.py's grid (sizes 64 ... 2500, 0-60 % outliers, dynamic and fixed stopping, three seeds),
    8 pairs per cell: every pair on exactly the oracle's trajectory, models within 2e-6."""
    from mdrp_amd import synth
    B = 8
    cam = po.cam_flat(0, [800.0, 0, 0])
    for N, of, opts in ((150, 0.2, {}), (400, 0.5, {"min_iterations": 500}), (1000, 0.6, {"max_iterations": 3000, "min_iterations": 3000}),
                        (64, 0.0, {"min_iterations": 200, "seed": 7}), (2500, 0.35, {"max_iterations": 1500, "min_iterations": 1500, "seed": 3})):
        b = synth.make_batch(9000 + 37 * N + 11 * kind + int(es), B, N, noise_px=0.7, depth_noise=0.03, outlier_frac=of, random_focal=rf,
                             shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0)
        ro = {"max_epipolar_error": 2.0, "max_reproj_error": 16.0, "monodepth_estimate_shift": es, **opts}
        cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
        res, mask = handle.estimate_batch(kind, b["x1"], b["x2"], b["d1"], b["d2"], capi.ransac_opt_from_dict(ro),
                                          capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), None,
                                          cams if kind == 0 else None, cams if kind == 0 else None)
        oro = po.ransac_opt(max_epipolar_error=2.0, max_reproj_error=16.0, estimate_shift=es, **opts)
        for i in range(B):
            m, st, mk = po.estimate(kind, b["x1"][i], b["x2"][i], b["d1"][i], b["d2"][i], oro, po.bundle_opt(loss_type=4),
                                    cam if kind == 0 else None, cam if kind == 0 else None)
            where = (kind, es, N, i)
            assert int(res[i]["iterations"]) == st.iterations and int(res[i]["refinements"]) == st.refinements, where
            assert int(res[i]["num_inliers"]) == st.num_inliers and (mask[i] == mk).all(), where
            assert model_diff(capi.model_to_array(res[i]["model"]), m) < 2e-6, where


@pytest.mark.parametrize("name", list(OPTIONS_NAMES))
def test_batched_ragged_calls_under_random_options_vs_oracle(handle, capi, po, golden, name):
    """The options campaign above runs one pair per call; the timed path is the BATCHED call.  Here the first 12 option sets of
    tests/golden/options_ref.npz (every loss type twice, thresholds, Sampson weight, budgets, stopping rule, LM tolerances and damping as drawn there)
    each drive ONE call over 20 ragged pairs (N = 40 ... 1500 in one buffer, per-pair cameras for the calibrated estimators): every pair on the
    oracle's trajectory — iterations, inlier count, mask, LO count (+-1 below N = 100, where scores tie) — and its model within 2e-6."""
    from mdrp_amd import synth
    g = golden("options_ref")
    kind, es, rf = OPTIONS_KINDS[name]
    B, n_max = 20, 1500
    sizes = [40, 41, 63, 64, 65, 100, 150, 255, 256, 257, 400, 511, 512, 700, 900, 1023, 1024, 1200, 1499, 1500]
    focals = [500.0, 800.0, 1400.0, 650.0]
    for j in range(12):
        row = g["cases"][j]
        rod, bod = options_dicts(row, es)
        x1, x2 = np.zeros((B, n_max, 2)), np.zeros((B, n_max, 2))
        d1, d2 = np.ones((B, n_max)), np.ones((B, n_max))
        pairs = []
        cams = np.zeros(B, dtype=capi.CAMERA_DTYPE)
        for i, n in enumerate(sizes):
            f = focals[i % 4]
            p = synth.make_pair(97000 + 100 * j + i, n, noise_px=float(row[2]), depth_noise=0.02, outlier_frac=[0.0, 0.3, 0.5, 0.6][(i + j) % 4], random_focal=rf,
                                shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0, **(dict(f1=f, f2=f, pp=(3.0, -2.0)) if kind == 0 else {}))
            x1[i, :n], x2[i, :n], d1[i, :n], d2[i, :n] = p["x1"], p["x2"], p["d1"], p["d2"]
            cams["params"][i, :3] = (f, 3.0, -2.0)
            pairs.append(p)
        ro = capi.ransac_opt_from_dict({("monodepth_" + k if k in ("estimate_shift", "weight_sampson") else k): v for k, v in rod.items()})
        res, mask = handle.estimate_batch(kind, x1, x2, d1, d2, ro, capi.bundle_opt_from_dict(bod), np.array(sizes, dtype=np.int32),
                                          cams if kind == 0 else None, cams if kind == 0 else None)
        for i, n in enumerate(sizes):
            p = pairs[i]
            cam = po.cam_flat(0, [focals[i % 4], 3.0, -2.0]) if kind == 0 else None
            m, st, mk = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], po.ransac_opt(**rod), po.bundle_opt(**bod), cam, cam)
            where = (name, j, i, n)
            assert (int(res[i]["iterations"]), int(res[i]["num_inliers"])) == (st.iterations, st.num_inliers), where
            assert (mask[i][:n] == mk).all() and not mask[i][n:].any(), where
            assert model_diff(capi.model_to_array(res[i]["model"]), m) < 2e-6, where
            # a few dozen correspondences repeat samples and tie scores in the last bits (DESIGN.md §5 class v): the LO count may differ by one there
            assert abs(int(res[i]["refinements"]) - st.refinements) <= (1 if n < 100 else 0), (where, int(res[i]["refinements"]), st.refinements)


def test_estimate_shift_flag_ignored_by_focal_estimators(handle, capi):
    """BASELINE configs[3] combines varying focal with monodepth_estimate_shift=True; the reference reads that flag only in
    the calibrated estimator -> identical results with and without it, shifts stay 0"""
    from mdrp_amd import synth
    b = synth.make_batch(3000, 4, 400, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5, random_focal="varying")
    out = []
    for flag in (False, True):
        ro = {"max_iterations": 2000, "min_iterations": 2000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0, "monodepth_estimate_shift": flag}
        res, mask = handle.estimate_batch(capi.VARYING_FOCAL, b["x1"], b["x2"], b["d1"], b["d2"], capi.ransac_opt_from_dict(ro),
                                          capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}))
        out.append((res.tobytes(), mask.tobytes()))
        assert (res["model"]["shift1"] == 0).all() and (res["model"]["shift2"] == 0).all()
    assert out[0] == out[1]


def test_batch_ragged_and_degenerate(handle, capi, po):
    """B pairs with different N (one sample table per N), including N<3 and N=0, equal the oracle pair by pair"""
    from mdrp_amd import synth
    ns = [200, 0, 2, 3, 150, 200, 777]
    B, N = len(ns), max(ns)
    x1 = np.zeros((B, N, 2)); x2 = np.zeros((B, N, 2)); d1 = np.ones((B, N)); d2 = np.ones((B, N))
    pairs = []
    for i, n in enumerate(ns):
        p = synth.make_pair(40 + i, max(n, 3), noise_px=0.0, depth_noise=0.0, random_focal="shared")
        x1[i, :n] = p["x1"][:n]; x2[i, :n] = p["x2"][:n]; d1[i, :n] = p["d1"][:n]; d2[i, :n] = p["d2"][:n]
        pairs.append(p)
    ro = {"max_iterations": 500, "min_iterations": 500, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}
    res, mask = handle.estimate_batch(capi.SHARED_FOCAL, x1, x2, d1, d2, capi.ransac_opt_from_dict(ro),
                                      capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), np.array(ns, np.int32))
    for i, n in enumerate(ns):
        if n < 3:
            assert int(res[i]["iterations"]) == 0 and int(res[i]["num_inliers"]) == 0 and res[i]["model_score"] > 1e300
            assert mask[i].sum() == 0
            continue
        m, st, mk = po.estimate(po.SHARED, x1[i, :n], x2[i, :n], d1[i, :n], d2[i, :n],
                                po.ransac_opt(max_iterations=500, min_iterations=500, max_epipolar_error=2.0, max_reproj_error=16.0),
                                po.bundle_opt(loss_type=4))
        assert int(res[i]["iterations"]) == st.iterations
        assert int(res[i]["num_inliers"]) == st.num_inliers
        if n > 3:
            assert model_diff(capi.model_to_array(res[i]["model"]), m) < 1e-6, (i, n)
        assert (mask[i, :n] == mk).all() and mask[i, n:].sum() == 0


def test_all_degenerate_and_empty_batches(handle, capi):
    """ransac<> early-out (@0x22f087): fewer than 3 correspondences -> zeroed stats, model_score = DBL_MAX, identity model"""
    ro, bo = capi.ransac_opt_from_dict({}), capi.bundle_opt_from_dict({})
    x = np.zeros((3, 2, 2)); d = np.ones((3, 2))
    res, mask = handle.estimate_batch(capi.VARYING_FOCAL, x, x, d, d, ro, bo, np.array([0, 1, 2], np.int32))
    assert (res["iterations"] == 0).all() and (res["num_inliers"] == 0).all() and (res["model_score"] > 1e300).all()
    assert np.allclose(res["model"]["q"], [[1, 0, 0, 0]] * 3) and (res["model"]["scale"] == 1).all() and mask.sum() == 0
    res, mask = handle.estimate_batch(capi.SHARED_FOCAL, np.zeros((2, 0, 2)), np.zeros((2, 0, 2)), np.zeros((2, 0)), np.zeros((2, 0)), ro, bo)
    assert len(res) == 2 and (res["iterations"] == 0).all()
    res, mask = handle.estimate_batch(capi.SHARED_FOCAL, np.zeros((0, 5, 2)), np.zeros((0, 5, 2)), np.zeros((0, 5)), np.zeros((0, 5)), ro, bo)
    assert len(res) == 0


@pytest.mark.timeout(120)
def test_garbage_inputs_terminate(handle, capi):
    """NaN / inf / negative depths, all-identical correspondences and huge coordinates must neither hang nor crash the
    kernels (the reference callers sanitise depths themselves, eval.py:344-346; the binding must still be safe)"""
    from mdrp_amd import synth
    b = synth.make_batch(4000, 5, 300, noise_px=0.5, outlier_frac=0.3)
    x1, x2, d1, d2 = b["x1"].copy(), b["x2"].copy(), b["d1"].copy(), b["d2"].copy()
    d1[0, ::7] = np.nan; d2[0, ::5] = np.inf; d1[0, ::11] = -1.0
    x1[1] = x1[1, :1]; x2[1] = x2[1, :1]; d1[1] = 1.0; d2[1] = 1.0          # one correspondence repeated 300 times
    x1[2] *= 1e12; x2[2] *= 1e-12                                           # absurd scales
    d1[3] = 0.0; d2[3] = 0.0                                                # zero depths
    cams = np.zeros(5, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    for kind in (capi.CALIB, capi.SHARED_FOCAL, capi.VARYING_FOCAL):
        for es in (False, True):
            ro = capi.ransac_opt_from_dict({"max_iterations": 600, "min_iterations": 300, "max_epipolar_error": 2.0,
                                            "max_reproj_error": 16.0, "monodepth_estimate_shift": es})
            res, mask = handle.estimate_batch(kind, x1, x2, d1, d2, ro, capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), None,
                                              cams if kind == 0 else None, cams if kind == 0 else None)
            assert (res["iterations"] >= 301).all() and (res["iterations"] <= 600).all()
            assert (res["num_inliers"] <= 300).all()
            assert int(res[4]["num_inliers"]) > 150  # the clean pair next to them is unaffected


def test_dynamic_stopping_chunks(handle, capi, po):
    """default options (max 100000 / min 1000): the chunked driver must stop at the reference's iteration"""
    from mdrp_amd import synth
    for idx, of in ((1, 0.0), (2, 0.5), (3, 0.7)):
        p = synth.make_pair(70 + idx, 400, noise_px=0.5, depth_noise=0.02, outlier_frac=of)
        cam = np.array([0, 3, 800.0, 0.0, 0.0, 0.0])
        ro = {"max_epipolar_error": 2.0, "max_reproj_error": 16.0, "min_iterations": 100 if of else 1000}
        res, mask = _run_estimate(capi, handle, 0, p["x1"], p["x2"], p["d1"], p["d2"], ro, {"loss_type": "TRUNCATED_CAUCHY"}, cam, cam)
        m, st, mk = po.estimate(po.CALIB, p["x1"], p["x2"], p["d1"], p["d2"],
                                po.ransac_opt(max_epipolar_error=2.0, max_reproj_error=16.0, min_iterations=100 if of else 1000),
                                po.bundle_opt(loss_type=4), po.cam_flat(0, [800.0, 0, 0]), po.cam_flat(0, [800.0, 0, 0]))
        assert int(res["iterations"]) == st.iterations, (of, int(res["iterations"]), st.iterations)
        assert int(res["num_inliers"]) == st.num_inliers and int(res["refinements"]) == st.refinements and (mask == mk).all(), of


def test_poselib_signatures(po):
    """the drop-in module: same call shape and info keys as the reference demo (make_pair.py:111, notebook cell 16)"""
    import mdrp_amd.poselib as poselib
    from mdrp_amd import synth
    p = synth.make_pair(5, 300, noise_px=0.0, depth_noise=0.0)
    cam = {"model": "SIMPLE_PINHOLE", "width": 1600, "height": 1200, "params": [800.0, 0.0, 0.0]}
    ro = {"max_epipolar_error": 2.0, "max_reproj_error": 16.0, "lo_iterations": 25, "progressive_sampling": False}
    geom, info = poselib.estimate_monodepth_relative_pose(p["x1"].astype(np.float32), p["x2"].astype(np.float32), p["d1"], list(p["d2"]),
                                                          cam, cam, ro, {"loss_type": "TRUNCATED_CAUCHY"})
    assert set(info) == {"refinements", "iterations", "num_inliers", "inlier_ratio", "model_score", "inliers"}
    assert len(info["inliers"]) == 300 and isinstance(info["inliers"][0], bool)
    assert synth.rotation_error_deg(p["R"], geom.pose.R) < 1e-3 and abs(geom.scale - p["scale"]) < 1e-3 * p["scale"]
    info_calib_refinements = info["refinements"]
    p = synth.make_pair(6, 300, noise_px=0.0, depth_noise=0.0, random_focal="varying")
    pair, info = poselib.estimate_monodepth_varying_focal_relative_pose(p["x1"], p["x2"], p["d1"], p["d2"], ro, {"loss_type": "TRUNCATED_CAUCHY"})
    assert abs(pair.camera1.focal() - p["f1"]) < 1e-6 * p["f1"] and abs(pair.camera2.focal() - p["f2"]) < 1e-6 * p["f2"]
    assert pair.geometry.shift1 == 0.0 and np.allclose(pair.pose.R, pair.geometry.pose.R)
    sols = poselib.varying_focal_monodepth_pose_4pt(np.c_[p["x1"][:3] / 500, np.ones(3)], np.c_[p["x2"][:3] / 500, np.ones(3)], p["d1"][:3], p["d2"][:3])
    assert len(sols) == 1 and abs(sols[0].camera1.focal() * 500 - p["f1"]) < 1e-6 * p["f1"]
    # the older wheel's names of the same estimators (demo/poselib_old-*.whl _core.pyi:441-497): MonoDepthCameraPose carries scale / shifts itself
    old_pair, old_info = poselib.estimate_monodepth_varying_focal_pose(p["x1"], p["x2"], p["d1"], p["d2"], ro, {"loss_type": "TRUNCATED_CAUCHY"})
    assert old_info == info and isinstance(old_pair.pose, poselib.MonoDepthCameraPose)
    assert np.array_equal(old_pair.pose.q, pair.geometry.pose.q) and old_pair.pose.scale == pair.geometry.scale and old_pair.camera1.focal() == pair.camera1.focal()
    p = synth.make_pair(5, 300, noise_px=0.0, depth_noise=0.0)
    old_pose, old_info = poselib.estimate_monodepth_pose(p["x1"], p["x2"], p["d1"], p["d2"], cam, cam, ro, {"loss_type": "TRUNCATED_CAUCHY"}, initial_pose=poselib.MonoDepthCameraPose())
    new_geom, new_info = poselib.estimate_monodepth_relative_pose(p["x1"], p["x2"], p["d1"], p["d2"], cam, cam, ro, {"loss_type": "TRUNCATED_CAUCHY"},
                                                                  initial_pose=poselib.MonoDepthTwoViewGeometry())  # (an initial pose sets score_initial_model: one more LO)
    assert np.array_equal(old_pose.q, new_geom.pose.q) and old_pose.scale == new_geom.scale and (old_pose.shift_1, old_pose.shift_2) == (0.0, 0.0)
    assert old_info == new_info and old_info["refinements"] != info_calib_refinements  # (the records start from the reset model: another trajectory)
    ps = synth.make_pair(7, 300, noise_px=0.0, depth_noise=0.0, random_focal="shared")
    old_pair, _ = poselib.estimate_monodepth_shared_focal_pose(ps["x1"], ps["x2"], ps["d1"], ps["d2"], ro, {"loss_type": "TRUNCATED_CAUCHY"})
    assert abs(old_pair.camera1.focal() - ps["f1"]) < 1e-6 * ps["f1"] and synth.rotation_error_deg(ps["R"], old_pair.pose.R) < 1e-3


def test_full_size_noisy_vs_oracle(handle, capi, po):
    """BASELINE configs[1] shape — N = 2000, 10k iterations, 50 % outliers, noisy — on 6 pairs against the CPU oracle.
    At this size the driver runs chunks with bail-out scoring; the trajectory must still be the sequential one:
    same iterations / LO count / inliers / mask, model within 1e-6, on every pair."""
    from mdrp_amd import synth
    B = 6
    b = synth.make_batch(500, B, 2000, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    ro = {"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}
    res, mask = handle.estimate_batch(capi.CALIB, b["x1"], b["x2"], b["d1"], b["d2"], capi.ransac_opt_from_dict(ro),
                                      capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), None, cams, cams)
    for i in range(B):
        m, st, mk = po.estimate(po.CALIB, b["x1"][i], b["x2"][i], b["d1"][i], b["d2"][i],
                                po.ransac_opt(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, max_reproj_error=16.0),
                                po.bundle_opt(loss_type=4), po.cam_flat(0, [800.0, 0, 0]), po.cam_flat(0, [800.0, 0, 0]))
        assert int(res[i]["iterations"]) == st.iterations == 10000
        assert int(res[i]["num_inliers"]) == st.num_inliers and int(res[i]["refinements"]) == st.refinements and (mask[i] == mk).all(), i
        assert model_diff(capi.model_to_array(res[i]["model"]), m) < 1e-6, i


@pytest.mark.parametrize("kind,es,rf", [(0, False, None), (0, True, None), (1, False, "shared"), (2, False, "varying")])
def test_sizes_beyond_the_sixteen_bit_lists_vs_oracle(handle, capi, po, kind, es, rf):
    """N = 70 001 correspondences in one pair (past every u16 index and LDS work list of the LM sweeps, and not a multiple of any tile): stats and
    mask identical to the oracle, model to 1e-6."""
    from mdrp_amd import synth
    n = 70001
    p = synth.make_pair(88000 + kind + int(es), n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.4, random_focal=rf, shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0)
    ro = dict(max_iterations=300, min_iterations=300, max_epipolar_error=2.0, max_reproj_error=16.0, seed=1)
    cams = np.zeros(1, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    res, mask = handle.estimate_batch(kind, p["x1"][None], p["x2"][None], p["d1"][None], p["d2"][None], capi.ransac_opt_from_dict(dict(ro, monodepth_estimate_shift=es)),
                                      capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), None, cams if kind == 0 else None, cams if kind == 0 else None)
    cam = po.cam_flat(0, [800.0, 0.0, 0.0]) if kind == 0 else None
    m, st, mk = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], po.ransac_opt(estimate_shift=es, **ro), po.bundle_opt(loss_type=4), cam, cam)
    r = res[0]
    assert (int(r["refinements"]), int(r["iterations"]), int(r["num_inliers"])) == (st.refinements, st.iterations, st.num_inliers)
    assert (mask[0][:n] == mk).all() and model_diff(capi.model_to_array(r["model"]), m) < 1e-6


def test_full_size_properties(handle, capi):
    """BASELINE config 2 shape (N=2000, 10k iterations) on a few pairs: size-independent properties —
    noise-free pairs recover ground truth to 1e-6; every true inlier is flagged; duplicating a pair gives
    bit-identical results (batch independence)."""
    from mdrp_amd import synth
    b = synth.make_batch(100, 3, 2000, noise_px=0.0, depth_noise=0.0)
    x1 = np.concatenate([b["x1"], b["x1"][:1]]); x2 = np.concatenate([b["x2"], b["x2"][:1]])
    d1 = np.concatenate([b["d1"], b["d1"][:1]]); d2 = np.concatenate([b["d2"], b["d2"][:1]])
    cams = np.zeros(4, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    ro = capi.ransac_opt_from_dict({"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
    res, mask = handle.estimate_batch(capi.CALIB, x1, x2, d1, d2, ro, capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), None, cams, cams)
    from helpers import quat_to_R
    for i in range(3):
        gt = b["gt"][i]
        assert np.abs(quat_to_R(res[i]["model"]["q"]) - gt["R"]).max() < 1e-6
        assert np.abs(res[i]["model"]["t"] - gt["t"]).max() < 1e-6 and abs(res[i]["model"]["scale"] - gt["scale"]) < 1e-6 * gt["scale"]
        assert int(res[i]["iterations"]) == 10000 and int(res[i]["num_inliers"]) == 2000 and mask[i].all()
    assert res[3].tobytes() == res[0].tobytes() and (mask[3] == mask[0]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(12, 700), (12, 2300), (136, 2300)])
def test_schedule_does_not_change_results(handle, capi, monkeypatch, B, N):
    """The chunk split (and with it which hypotheses k_count / k_bound retire against which records), the three-stream
    pipeline, the fp32 bound stage and the LO grid are scheduling only: every setting must reproduce the same records bit
    for bit (single chunk on one stream, where nothing is ever retired, is the plain sequential schedule).
    N = 2300 (nine 256-record tiles): k_count's TWO-PHASE retirement is active in every chunked schedule (phase A over the leading tiles,
    phase B for the undecided hypotheses) and absent from the single-chunk reference run; N = 700 (three tiles) never splits.
    B = 12: the small-call path (k_score_w, no fp32 stage); B = 136: k_bound and the lane-per-hypothesis k_score."""
    from mdrp_amd import synth
    b = synth.make_batch(4000, B, N, noise_px=0.5, depth_noise=0.02, outlier_frac=0.4)
    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE)
    cams["params"][:, 0] = 800.0
    ro = capi.ransac_opt_from_dict({"max_iterations": 4000, "min_iterations": 4000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
    bo = capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})

    def run():
        res, mask = handle.estimate_batch(0, b["x1"], b["x2"], b["d1"], b["d2"], ro, bo, None, cams, cams)
        return res.copy(), mask.copy()

    monkeypatch.setenv("MDRP_CHUNKS", "0")          # no leading chunk: one chunk, no bail-out bar, no overlap
    monkeypatch.setenv("MDRP_LO_OVERLAP", "0")
    ref, ref_mask = run()
    assert int(ref["iterations"].min()) == 4000 and int(ref["num_inliers"].min()) > 300
    for env in ({"MDRP_CHUNKS": "512"}, {"MDRP_CHUNKS": "256,1024"}, {"MDRP_CHUNKS": "512", "MDRP_LO_OVERLAP": "0"},
                {"MDRP_CHUNKS": "128,256,512"}, {"MDRP_CHUNKS": "128", "MDRP_BOUND": "0"}, {"MDRP_CHUNKS": "64,128", "MDRP_LO_OVERLAP": "0"},
                {"MDRP_CHUNKS": "512", "MDRP_LO_THREADS": "256", "MDRP_FINAL_THREADS": "64"},
                {"MDRP_CHUNKS": "128", "MDRP_LO_THREADS": "64"}, {"MDRP_CHUNKS": "128", "MDRP_FINAL_THREADS": "64"},
                {"MDRP_CHUNKS": "128", "MDRP_FUSE_TAIL": "0"}, {"MDRP_CHUNKS": "128,512", "MDRP_FUSE_TAIL": "1"},
                {"MDRP_CHUNKS": "64,256", "MDRP_FUSE_TAIL": "0", "MDRP_BOUND": "0"}):
        for k in ("MDRP_CHUNKS", "MDRP_LO_OVERLAP", "MDRP_BOUND", "MDRP_LO_THREADS", "MDRP_FINAL_THREADS", "MDRP_FUSE_TAIL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        res, mask = run()
        for f in ("refinements", "iterations", "num_inliers"):
            assert np.array_equal(res[f], ref[f]), (env, f)
        assert np.array_equal(mask, ref_mask), env
        # thread-count variants reduce in a different order: models agree to rounding, not bitwise
        tol = 1e-9 if ("MDRP_LO_THREADS" in env or "MDRP_FINAL_THREADS" in env) else 0.0
        def flat(m):
            return np.c_[m["q"], m["t"], m["scale"], m["shift1"], m["shift2"], m["f1"], m["f2"]]
        assert np.allclose(flat(res["model"]), flat(ref["model"]), rtol=tol, atol=tol), env
        assert np.allclose(res["model_score"], ref["model_score"], rtol=max(tol, 0.0), atol=0.0), env


@pytest.mark.gpu
@pytest.mark.parametrize("n_max", [900, 5440, 5441, 6000])
def test_final_refinement_over_the_inlier_index_at_and_beyond_the_list_limit(capi, monkeypatch, n_max):
    """The inlier-only final refinement (estimate_* tail @0x2247c3 / @0x223815) walks a compacted index of the inliers (lm_mask_index: a third LDS list,
    N <= 5440); beyond that size it evaluates every record under the mask.  Both paths, at and around the limit, with 64 and 256 lanes per pair, on ragged
    pairs (short ones, few inliers) for the calibrated and the varying-focal estimator: integer statistics and masks identical between the two widths, models
    to 1e-9 (another summation tree)."""
    from mdrp_amd import synth
    ns = [n_max, n_max - 1, 700, 257, 64, 5, 3, 0]
    B = len(ns)
    for kind, rf in ((0, None), (2, "varying")):
        x1, x2 = np.zeros((B, n_max, 2)), np.zeros((B, n_max, 2))
        d1, d2 = np.ones((B, n_max)), np.ones((B, n_max))
        for i, n in enumerate(ns):
            if n:
                b = synth.make_batch(7100 + 13 * i + kind, 1, n, noise_px=0.6, depth_noise=0.02, outlier_frac=0.7 if i == 2 else 0.4, random_focal=rf)
                x1[i, :n], x2[i, :n], d1[i, :n], d2[i, :n] = b["x1"][0], b["x2"][0], b["d1"][0], b["d2"][0]
        cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
        ro = capi.ransac_opt_from_dict({"max_iterations": 400, "min_iterations": 400, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
        bo = capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
        out = {}
        for threads in ("256", "64"):
            monkeypatch.setenv("MDRP_FINAL_THREADS", threads)
            h = capi.Handle(0)
            res, mask = h.estimate_batch(kind, x1, x2, d1, d2, ro, bo, np.array(ns, dtype=np.int32), cams if kind == 0 else None, cams if kind == 0 else None)
            out[threads] = (res.copy(), mask.copy())
            h.close()
        def flat(m):
            return np.c_[m["q"], m["t"], m["scale"], m["shift1"], m["shift2"], m["f1"], m["f2"]]
        (r0, m0), (r1, m1) = out["256"], out["64"]
        for f in ("refinements", "iterations", "num_inliers"):
            assert np.array_equal(r0[f], r1[f]), (kind, f)
        assert np.array_equal(m0, m1), kind
        assert np.allclose(flat(r0["model"]), flat(r1["model"]), rtol=1e-9, atol=1e-9), kind
        assert int(r0["num_inliers"][0]) > n_max // 3


@pytest.mark.gpu
@pytest.mark.parametrize("kind,rf", [(0, None), (2, "varying"), (5, None)])
def test_wave_per_hypothesis_scoring_equals_lane_per_hypothesis(capi, monkeypatch, kind, rf):
    """Calls of at most 128 pairs score their hypotheses with one WAVEFRONT each (k_score_w: 64 records per trip, the inliers' r^2 added in record
    order through LDS); larger calls with one LANE each (k_score).  Same arithmetic, same order of the sum, equivalent bail-outs: 100 ragged pairs
    estimated as a call of their own and as the first half of a 200-pair call must give the same records and masks BIT FOR BIT (the lanes per LO
    problem are pinned: they too depend on the size of the call)."""
    from mdrp_amd import synth
    monkeypatch.setenv("MDRP_LO_THREADS", "64")
    B, N = 200, 900
    ns = [N, 700, 257, 64, 63, 5, 3, 2, 0, 450] * (B // 10)
    x1, x2 = np.zeros((B, N, 2)), np.zeros((B, N, 2))
    d1, d2 = np.ones((B, N)), np.ones((B, N))
    for i, n in enumerate(ns):
        if n:
            p = synth.make_pair(12000 + i, n, noise_px=0.5, depth_noise=0.02, outlier_frac=[0.5, 0.0, 0.3][i % 3], random_focal=rf)
            x1[i, :n], x2[i, :n], d1[i, :n], d2[i, :n] = p["x1"], p["x2"], p["d1"], p["d2"]
    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    ro = capi.ransac_opt_from_dict({"max_iterations": 3000, "min_iterations": 3000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
    bo = capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
    c = cams if kind == 0 else None
    npp = np.array(ns, dtype=np.int32)
    h = capi.Handle(0)
    try:
        big, big_mask = h.estimate_batch(kind, x1, x2, d1, d2, ro, bo, npp, c, c)
        small, small_mask = h.estimate_batch(kind, x1[:100], x2[:100], d1[:100], d2[:100], ro, bo, npp[:100], None if c is None else c[:100], None if c is None else c[:100])
    finally:
        h.close()
    assert int(small["num_inliers"].max()) > 400 and int(small["refinements"].max()) > 3
    assert small.tobytes() == big[:100].tobytes(), np.nonzero(small["model_score"] != big[:100]["model_score"])[0][:10]
    assert np.array_equal(small_mask, big_mask[:100])


@pytest.mark.gpu
def test_device_resident_batch_matches_host_batch(capi):
    """poselib.estimate_batch_torch (tensors already on the GPU, current torch stream) == the host-buffer batch API"""
    import torch
    import mdrp_amd.poselib as poselib
    from mdrp_amd import synth
    b = synth.make_batch(6100, 10, 350, noise_px=0.5, depth_noise=0.02, outlier_frac=0.3)
    cam = {"model": "SIMPLE_PINHOLE", "width": 1600, "height": 1200, "params": [800.0, 0.0, 0.0]}
    ro = {"max_iterations": 1500, "min_iterations": 1500, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}
    bo = {"loss_type": "TRUNCATED_CAUCHY"}
    geoms, infos = poselib.estimate_monodepth_relative_pose_batch(b["x1"], b["x2"], b["d1"], b["d2"], cam, cam, ro, bo)
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(b[k]).to(dev) for k in ("x1", "x2", "d1", "d2")]
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        res, mask = poselib.estimate_batch_torch("calibrated", *t, cam, cam, ro, bo)
    assert mask.is_cuda and mask.shape == (10, 350)
    for i in range(10):
        assert int(res[i]["num_inliers"]) == infos[i]["num_inliers"] and int(res[i]["refinements"]) == infos[i]["refinements"]
        assert np.array_equal(mask[i].cpu().numpy().astype(bool), np.array(infos[i]["inliers"]))
        assert np.allclose(res[i]["model"]["q"], geoms[i].pose.q, atol=0, rtol=0)
    with pytest.raises(ValueError):
        poselib.estimate_batch_torch("calibrated", t[0].float(), t[1], t[2], t[3], cam, cam, ro, bo)


@pytest.mark.gpu
def test_sampler_workgroup_size_does_not_change_results(capi, monkeypatch):
    """The sample tables are drawn speculatively by one workgroup per table (thread i assumes that no sample before its own met a
    rejection; the threads up to the first rejection are right): the sequence must not depend on how many threads speculate.  Tiny
    N, where almost every step ends at a rejection, for the 3-, 5- and 7-point samplers: 64 threads (one wavefront, the round-2
    scheme) against 256 and 1024, records and masks bit for bit."""
    from mdrp_amd import synth
    h = capi.Handle(0)
    try:
        out = {}
        for thr in ("64", "256", "1024"):
            monkeypatch.setenv("MDRP_SAMPLE_THREADS", thr)
            rows = []
            for N in (3, 4, 5, 7, 9, 20):
                for kind in (0, 3, 5):
                    if N < {0: 3, 3: 5, 5: 7}[kind]:
                        continue
                    B = 6
                    b = synth.make_batch(7000 + N, B, N, noise_px=0.5, depth_noise=0.02, outlier_frac=0.0)
                    ro = capi.ransac_opt_from_dict({"max_iterations": 2000, "min_iterations": 2000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
                    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE)
                    cams["params"][:, 0] = 800.0
                    res, mask = h.estimate_batch(kind, b["x1"], b["x2"], b["d1"], b["d2"], ro, capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}),
                                                 None, cams if kind in (0, 3) else None, cams if kind in (0, 3) else None)
                    rows.append(res.tobytes() + mask.tobytes())
            out[thr] = rows
        assert out["64"] == out["256"] == out["1024"]
    finally:
        h.close()
