"""The CPU oracle (oracle/*.c) against golden vectors produced by the reference's own PoseLib binary
(tests/tools/gen_golden.py).  No GPU."""
import numpy as np
import pytest

from oracle import pyorc as po
from helpers import (INITIAL_ORACLE_DEVIATIONS, KNOWN_LO_COUNT_DEVIATIONS, OPTIONS_KINDS, OPTIONS_LO_DEVIATIONS, OPTIONS_MODEL_DEVIATIONS, OPTIONS_NAMES, OPTIONS_OTHER_WINNER,
                     REFERENCE_NAN_SOLUTIONS, input_digest, match_solution_sets, model_diff, options_cameras, options_dicts, options_pair, refine_ws_weights, widen)


def test_sampler_known_answers(golden):
    g = golden("sampler")
    for key in g.files:
        n, seed = int(key.split("_")[0][1:]), int(key.split("_")[1][1:])
        assert (po.draw_samples(seed, n, 64) == g[key]).all(), key
    # SURVEY.md §8a-3 known answers (seed 0)
    assert po.draw_samples(0, 2000, 2).tolist() == [[767, 356, 1535], [620, 395, 298]]
    assert po.draw_samples(0, 7, 2).tolist() == [[0, 1, 2], [3, 0, 4]]


def test_scoring_bit_exact(golden):
    g = golden("scoring")
    for i in range(6):
        x1, x2, m, thr = g[f"x1_{i}"], g[f"x2_{i}"], g[f"model_{i}"], float(g[f"thr_{i}"])
        s, c = po.msac_pose(m, x1, x2, thr)
        assert s == float(g[f"pose_score_{i}"]) and c == int(g[f"pose_cnt_{i}"])
        assert (po.inliers_pose(m, x1, x2, thr) == g[f"pose_mask_{i}"]).all()
        F = po.fundamental(m)
        assert np.allclose(F, g[f"F_{i}"], rtol=1e-14, atol=1e-15)
        s, c = po.msac_F(g[f"F_{i}"], x1, x2, thr)
        assert s == float(g[f"F_score_{i}"]) and c == int(g[f"F_cnt_{i}"])
        assert (po.inliers_F(g[f"F_{i}"], x1, x2, thr) == g[f"F_mask_{i}"]).all()


@pytest.mark.parametrize("kind", ["p3p", "calib_shift", "shared", "varying"])
def test_solver_solution_sets(golden, kind):
    """Solution sets equal the reference's on every golden problem.  The only exclusion is enumerated in
    helpers.REFERENCE_NAN_SOLUTIONS: problems on which the reference binary itself returns NaN models."""
    g = golden("solvers")
    n = len(g[f"{kind}_n"])
    agree = checked = 0
    for i in range(n):
        nref = int(g[f"{kind}_n"][i])
        ref = g[f"{kind}_sols"][i][:nref]
        if np.isnan(ref).any():
            assert i in REFERENCE_NAN_SOLUTIONS[kind], (kind, i)
            continue
        if kind == "p3p":
            mine = po.p3p(g["p3p_x"][i], g["p3p_X"][i])
        else:
            fn = {"calib_shift": po.solver_calib_shift, "shared": po.solver_shared, "varying": po.solver_varying}[kind]
            mine = fn(g[f"{kind}_x1"][i], g[f"{kind}_x2"][i], g[f"{kind}_d1"][i], g[f"{kind}_d2"][i])
        ref = [widen(r) for r in ref]
        if kind == "p3p":
            mine = [np.r_[m[:7], np.ones(5)] for m in mine]
        checked += 1
        agree += match_solution_sets(ref, list(mine), 1e-6)
    assert checked == n - len(REFERENCE_NAN_SOLUTIONS[kind])
    assert agree == checked, (agree, checked)


def test_refine_matches_reference(golden):
    g = golden("refine")
    worst = 0.0
    for ci, case in enumerate(g["cases"]):
        i, kind, es, lt, its, thr = int(case[0]), int(case[1]), int(case[2]), int(case[3]), int(case[4]), case[5]
        bo = po.bundle_opt(max_iterations=its, loss_type=lt, loss_scale=thr, gradient_tol=1e-10)
        m, st = po.refine(kind, g[f"x1_{i}"], g[f"x2_{i}"], g[f"d1_{i}"], g[f"d2_{i}"], g[f"model_{i}"], 1 / 64.0, 1.0, bo, es)
        ref = g[f"out_{ci}"]
        assert abs(st.initial_cost - ref[13]) <= 1e-12 * abs(ref[13]), (case, st.initial_cost, ref[13])
        d = model_diff(m, ref[:12])
        worst = max(worst, d)
        assert d < 1e-6, (case, d)
        assert abs(st.cost - ref[14]) <= 1e-9 * abs(ref[14]) + 1e-18
    assert worst < 1e-6


def test_weight_sampson_in_the_refiners(golden):
    """tests/golden/refine_ws.npz (tests/tools/gen_golden_refine_ws.py): the reference binary's three refiners at ws != 1 — cost ws rho(r^2), normal
    equations ws^2 w(.), loss weight at r^2 (calibrated) or ws r^2 (focal) — on the nine problems of refine.npz, all six losses, with and without
    per-correspondence weights: initial cost to 1e-12, model to 1e-6, final cost to 1e-9 on all 1296 cases."""
    g, inp = golden("refine_ws"), golden("refine")
    assert len(g["cases"]) == 1296
    worst = 0.0
    for case, ref in zip(g["cases"], g["out"]):
        i, kind, es, ws, lt, its, weighted, thr = int(case[0]), int(case[1]), int(case[2]), case[3], int(case[4]), int(case[5]), int(case[6]), case[7]
        bo = po.bundle_opt(max_iterations=its, loss_type=lt, loss_scale=thr, gradient_tol=1e-10)
        w = refine_ws_weights(i, len(inp[f"x1_{i}"])) if weighted else None
        m, st = po.refine(kind, inp[f"x1_{i}"], inp[f"x2_{i}"], inp[f"d1_{i}"], inp[f"d2_{i}"], inp[f"model_{i}"], 1 / 64.0, ws, bo, es, w)
        assert abs(st.initial_cost - ref[13]) <= 1e-12 * abs(ref[13]), (case, st.initial_cost, ref[13])
        d = model_diff(m, ref[:12])
        worst = max(worst, d)
        assert d < 1e-6, (case, d)
        assert abs(st.cost - ref[14]) <= 1e-9 * abs(ref[14]) + 1e-18, case
        assert st.iterations == int(ref[12]), case
    assert worst < 1e-6


@pytest.mark.parametrize("name", list(OPTIONS_NAMES))
def test_randomised_options_vs_reference_fixture(golden, name):
    """tests/golden/options_ref.npz: the reference binary on 4 x 96 cases whose problem size, outlier share, noise, thresholds, Sampson weight, seed,
    iteration budget (fixed and dynamic), stopping rule (success_prob, dyn_num_trials_mult), BundleOptions (loss type and scale, iteration cap, both
    tolerances, damping and its bounds) and cameras (calibrated: two focal lengths, principal point, SIMPLE_PINHOLE / PINHOLE) are all drawn at random —
    what the boundary hands through, varied together (the reference's own scripts only ever use one setting).  Oracle == reference in iterations, inlier
    count, mask and model (1e-6) on 383 of 384 cases (one enumerated shift case ends on another winner); the LO count differs on 8 (an exact list: the
    solver classes of DESIGN.md §5)."""
    g = golden("options_ref")
    kind, es, rf = OPTIONS_KINDS[name]
    lo_dev, model_dev = OPTIONS_LO_DEVIATIONS.get(name, {}), OPTIONS_MODEL_DEVIATIONS.get(name, {})
    for j, row in enumerate(g["cases"]):
        n = int(row[0])
        p = options_pair(name, j, row)
        assert input_digest(p) == g[f"{name}_digest"][j]
        if j in OPTIONS_OTHER_WINNER.get(name, ()):
            continue
        rod, bod = options_dicts(row, es)
        c1, c2 = options_cameras(row)
        cam1, cam2 = (po.cam_flat(*c1), po.cam_flat(*c2)) if kind == 0 else (None, None)
        m, st, mask = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], po.ransac_opt(**rod), po.bundle_opt(**bod), cam1, cam2)
        ist = g[f"{name}_istats"][j]
        assert (st.iterations, st.num_inliers) == (int(ist[1]), int(ist[2])), (name, j, st.iterations, st.num_inliers, ist)
        assert (mask == np.unpackbits(g[f"{name}_mask"][j])[:n]).all(), (name, j)
        assert model_diff(m, g[f"{name}_model"][j]) < model_dev.get(j, 1e-6), (name, j, model_diff(m, g[f"{name}_model"][j]))
        assert st.refinements - int(ist[0]) == lo_dev.get(j, 0), (name, j, st.refinements, int(ist[0]))


@pytest.mark.parametrize("name", list(OPTIONS_NAMES))
def test_edge_options_vs_reference_fixture(golden, name):
    """tests/golden/edge_options_ref.npz: one option at an edge of its range per case (max_iterations 0 / 1 / below min_iterations — the loop head checks
    before a sample is drawn and the closing LO then runs on the reset identity model, whose NaN cost must leave it untouched —, success_prob 0 / 1,
    dyn_num_trials_mult 0, thresholds 0 / 1e-3 / 100 px, reprojection off, weight_sampson 0 / negative, a 41-bit seed, loss_scale 0, pinned damping,
    tolerances of 1): stats, mask and model identical to the reference binary on all 4 x 25 cases (two of them, for the calibrated P3P estimator, runs whose only samples make the reference's p3p() return NaN poses: the answer is that NaN pose)."""
    from helpers import edge_cases, edge_pair, same_model
    g = golden("edge_options_ref")
    kind, es, rf = OPTIONS_KINDS[name]
    p = edge_pair(name)
    assert input_digest(p) == g[f"{name}_digest"]
    cam = po.cam_flat(0, [800.0, 0.0, 0.0]) if kind == 0 else None
    for j, (rod, bod) in enumerate(edge_cases()):
        m, st, mask = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], po.ransac_opt(estimate_shift=es, **rod), po.bundle_opt(**bod), cam, cam)
        ref = g[f"{name}_stats"][j]
        assert (st.refinements, st.iterations, st.num_inliers) == tuple(int(v) for v in ref[:3]), (name, j, rod, bod, st.refinements, st.iterations, st.num_inliers, ref)
        assert (mask == np.unpackbits(g[f"{name}_mask"][j])[:300]).all(), (name, j)
        assert same_model(m, g[f"{name}_model"][j]), (name, j, rod, bod, m, g[f"{name}_model"][j])
        assert st.model_score == ref[4] or abs(st.model_score - ref[4]) <= 1e-9 * abs(ref[4]), (name, j)


@pytest.mark.parametrize("name", list(OPTIONS_NAMES))
def test_corrupted_inputs_vs_reference_fixture(golden, name):
    """tests/golden/bad_inputs_ref.npz: a tenth of the correspondences with zero / negative / NaN depths (either image), NaN / inf coordinates, or all
    identical.  The reference drops the reprojection terms of non-positive AND of NaN depths (the LM cost stays finite), never counts a NaN / inf
    correspondence as an inlier, and lets a NaN coordinate poison the focal estimators' normalisation scale (0 inliers, NaN model).  Stats, mask and
    model identical on 27 of the 4 x 7 cases (the other one, garbage in both, is not compared: helpers.BAD_INPUT_SKIP)."""
    from helpers import BAD_INPUT_LO_DEVIATIONS, BAD_INPUT_MODES, BAD_INPUT_SKIP, bad_input_pair, same_model
    g = golden("bad_inputs_ref")
    kind, es, rf = OPTIONS_KINDS[name]
    cam = po.cam_flat(0, [800.0, 0.0, 0.0]) if kind == 0 else None
    for j, mode in enumerate(BAD_INPUT_MODES):
        if (name, mode) in BAD_INPUT_SKIP:
            continue
        p = bad_input_pair(name, mode)
        ro = po.ransac_opt(max_iterations=500, min_iterations=500, max_epipolar_error=2.0, max_reproj_error=16.0, seed=2, estimate_shift=es)
        m, st, mask = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], ro, po.bundle_opt(max_iterations=100, loss_type=4, loss_scale=1.0, gradient_tol=1e-10), cam, cam)
        ref = g[f"{name}_stats"][j]
        assert (st.iterations, st.num_inliers) == (int(ref[1]), int(ref[2])), (name, mode, st.iterations, st.num_inliers, ref)
        assert st.refinements - int(ref[0]) == BAD_INPUT_LO_DEVIATIONS.get((name, mode), 0), (name, mode, st.refinements, ref[0])
        assert (mask == np.unpackbits(g[f"{name}_mask"][j])[:300]).all(), (name, mode)
        assert same_model(m, g[f"{name}_model"][j]), (name, mode, m, g[f"{name}_model"][j])


@pytest.mark.parametrize("name", list(OPTIONS_NAMES))
def test_degenerate_geometry_vs_reference_fixture(golden, name):
    """tests/golden/degenerate_ref.npz: pure rotation, a planar scene, a baseline of 1e-4, motion along the optical axis (6 seeds each; 25 % outliers):
    stats (LO count included), mask and model identical to the reference binary on all 24 cases per estimator (one model 5e-6, enumerated)."""
    from helpers import DEGENERATE_MODEL_TOL, DEGENERATE_MODES, DEGENERATE_SEEDS, degenerate_pair, same_model
    g = golden("degenerate_ref")
    kind, es, rf = OPTIONS_KINDS[name]
    cam = po.cam_flat(0, [800.0, 0.0, 0.0]) if kind == 0 else None
    k = 0
    for mode in DEGENERATE_MODES:
        for seed in range(DEGENERATE_SEEDS):
            p = degenerate_pair(name, mode, seed)
            assert input_digest(p) == g[f"{name}_digest"][k]
            ro = po.ransac_opt(max_iterations=1000, min_iterations=1000, max_epipolar_error=2.0, max_reproj_error=16.0, seed=seed, estimate_shift=es)
            m, st, mask = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], ro, po.bundle_opt(max_iterations=100, loss_type=4, loss_scale=1.0, gradient_tol=1e-10), cam, cam)
            ref = g[f"{name}_stats"][k]
            assert (st.refinements, st.iterations, st.num_inliers) == tuple(int(v) for v in ref[:3]), (name, mode, seed, st.refinements, st.num_inliers, ref)
            assert (mask == np.unpackbits(g[f"{name}_mask"][k])[:400]).all(), (name, mode, seed)
            assert same_model(m, g[f"{name}_model"][k], DEGENERATE_MODEL_TOL.get((name, mode, seed), 1e-6)), (name, mode, seed)
            k += 1


def test_estimate_matches_reference(golden):
    g = golden("estimate")
    for case in g["cases"]:
        i, kind, es, noise, of, max_it, min_it, seed, lt = case
        i, kind, es = int(i), int(kind), int(es)
        ro = po.ransac_opt(max_iterations=int(max_it), min_iterations=int(min_it), max_epipolar_error=2.0,
                           max_reproj_error=16.0, seed=int(seed), estimate_shift=bool(es))
        bo = po.bundle_opt(loss_type=int(lt))
        c1, c2 = g[f"cam1_{i}"], g[f"cam2_{i}"]
        cam1 = po.cam_flat(int(c1[0]), list(c1[2:2 + int(c1[1])])) if kind == 0 else None
        cam2 = po.cam_flat(int(c2[0]), list(c2[2:2 + int(c2[1])])) if kind == 0 else None
        m, st, mask = po.estimate(kind, g[f"x1_{i}"], g[f"x2_{i}"], g[f"d1_{i}"], g[f"d2_{i}"], ro, bo, cam1, cam2)
        ref_m, ref_st, ref_mask = g[f"model_{i}"], g[f"stats_{i}"], g[f"mask_{i}"]
        assert st.iterations == int(ref_st[1])
        assert model_diff(m, ref_m) < 1e-6, (case, model_diff(m, ref_m))
        if noise > 0:
            # on noisy data the whole trajectory is reproduced: same LO count, inliers, score and mask
            assert st.refinements == int(ref_st[0])
            assert st.num_inliers == int(ref_st[2])
            assert abs(st.model_score - ref_st[4]) <= 1e-9 * ref_st[4]
            assert (mask == ref_mask).all()
        else:
            # noise-free: scores are ~1e-30 rounding noise, LO count may differ; results must not
            assert st.num_inliers == int(ref_st[2]) and (mask == ref_mask).all()
            assert st.model_score < 1e-20 or abs(st.model_score - ref_st[4]) <= 1e-6 * ref_st[4], (case, st.model_score, ref_st[4])


def full_size_cases(g):
    """(index, kind, estimate_shift, n, reference model, stats, mask) of tests/golden/estimate_full.npz"""
    for case in g["cases"]:
        i, kind, es, n = int(case[0]), int(case[1]), int(case[2]), int(case[3])
        yield i, kind, es, n, g[f"model_{i}"], g[f"stats_{i}"], np.unpackbits(g[f"mask_{i}"])[:n]


def test_estimate_full_size_matches_reference(golden):
    """BASELINE.json's full-size shapes (N = 2000 / 5000, 10^4 iterations, 50 % and 0 % outliers, every estimator incl.
    varying focal with the shift flag set) captured from the reference binary: the oracle lands on the same
    trajectory — iterations, LO count, inlier count, score, mask, model.  The LO count differs on exactly the two
    cases enumerated (with their cause) in helpers.KNOWN_LO_COUNT_DEVIATIONS."""
    g = golden("estimate_full")
    cam = po.cam_flat(0, [800.0, 0.0, 0.0])
    for i, kind, es, n, ref_m, ref_st, ref_mask in full_size_cases(g):
        ro = po.ransac_opt(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, max_reproj_error=16.0, estimate_shift=bool(es))
        m, st, mask = po.estimate(kind, g[f"x1_{i}"], g[f"x2_{i}"], g[f"d1_{i}"], g[f"d2_{i}"], ro, po.bundle_opt(loss_type=4),
                                  cam if kind == 0 else None, cam if kind == 0 else None)
        assert st.iterations == int(ref_st[1]) == 10000
        assert st.num_inliers == int(ref_st[2]), (i, st.num_inliers, ref_st[2])
        assert (mask == ref_mask).all(), i
        assert model_diff(m, ref_m) < 1e-6, (i, model_diff(m, ref_m))
        assert abs(st.model_score - ref_st[4]) <= 1e-9 * ref_st[4]
        assert st.refinements == int(ref_st[0]) + KNOWN_LO_COUNT_DEVIATIONS.get(i, 0), (i, st.refinements, ref_st[0])


def initial_cases(g):
    for case in g["cases"]:
        i, kind, flag, its, seed, deg = [int(v) for v in case]
        yield i, kind, bool(flag), its, seed, bool(deg)


def test_initial_model_and_score_initial_match_reference(golden):
    """initial_pose / score_initial_model (_core.pyi:455; ransac<> @0x22f2c8) against the reference binary: the pose handed in
    is never read, its scale survives only when nothing is adopted, and the flag costs one refinement — normal and fully
    degenerate data, 1 and 200 iterations, all three estimators."""
    g = golden("initial")
    for i, kind, flag, its, seed, deg in initial_cases(g):
        if i in INITIAL_ORACLE_DEVIATIONS:
            continue
        ro = po.ransac_opt(max_iterations=its, min_iterations=its, max_epipolar_error=2.0, max_reproj_error=16.0, seed=seed, score_initial_model=flag)
        cam = po.cam_flat(0, [800.0, 640.0, 480.0]) if kind == 0 else None
        ini = g[f"initial_{i}"]
        m, st, mask = po.estimate(kind, g[f"x1_{i}"], g[f"x2_{i}"], g[f"d1_{i}"], g[f"d2_{i}"], ro, po.bundle_opt(loss_type=4), cam, cam,
                                  initial=np.r_[ini, 1.0, 1.0] if kind == 0 else ini)
        ref_m, ref_st = g[f"model_{i}"], g[f"stats_{i}"]
        assert (st.refinements, st.iterations, st.num_inliers) == tuple(int(v) for v in ref_st[:3]), (i, st.refinements, ref_st)
        assert abs(st.model_score - ref_st[4]) <= 1e-9 * ref_st[4]
        assert model_diff(m, ref_m) < 1e-6, i
        assert (mask == g[f"mask_{i}"]).all()


def test_wide_full_size_pin_subsample(golden):
    """tests/golden/estimate_wide.npz (32 reference-binary runs per estimator at BASELINE's full shapes, seeds + outputs
    only): the stored input digests match what mdrp_amd.synth generates today for ALL 128 pairs, the stored port results
    were result-identical to the reference on all of them, and the oracle reproduces four pairs per estimator now
    (the GPU suite checks all 128 against the same file)."""
    import hashlib
    from mdrp_amd import synth
    g = golden("estimate_wide")
    cam = po.cam_flat(0, [800.0, 0.0, 0.0])
    deviations = 0
    for j, name in enumerate(g["names"]):
        kind, es, n = (int(v) for v in g["cases"][j])
        of = float(g["outlier_frac"][j])
        assert g[f"{name}_oracle_same"].all()
        deviations += int((g[f"{name}_oracle_refinements"] != g[f"{name}_stats"][:, 0].astype(int)).sum())
        for k, index in enumerate(g["indices"]):
            p = synth.make_pair(int(index), n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=[None, "shared", "varying"][kind],
                                shift1=0.2 if es and kind == 0 else 0.0, shift2=-0.1 if es and kind == 0 else 0.0)
            h = hashlib.sha256()
            for key in ("x1", "x2", "d1", "d2"):
                h.update(np.ascontiguousarray(p[key], dtype=np.float64).tobytes())
            assert np.frombuffer(h.digest()[:8], dtype=np.uint64)[0] == g[f"{name}_digest"][k], (name, index)
            if k % 8 != 3:
                continue
            ro = po.ransac_opt(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, max_reproj_error=16.0, estimate_shift=bool(es))
            m, st, mask = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], ro, po.bundle_opt(loss_type=4),
                                      cam if kind == 0 else None, cam if kind == 0 else None)
            ref_st = g[f"{name}_stats"][k]
            assert st.iterations == int(ref_st[1]) and st.num_inliers == int(ref_st[2]), (name, index)
            assert (mask == np.unpackbits(g[f"{name}_mask"][k])[:n]).all() and model_diff(m, g[f"{name}_model"][k]) < 1e-6, (name, index)
            assert st.refinements == int(g[f"{name}_oracle_refinements"][k]), (name, index)
    assert deviations == 2  # 2 of 128 pairs differ from the reference in the LO COUNT only (3 until the reference's NaN P3P poses were reproduced)


HEADLINE_SHAPES = {
    # workload: kind, shift flag, n, outlier_frac, random_focal, depth shifts  (bench.py WORKLOADS)
    "calib_p3p_n2000_i10k": (0, False, 2000, 0.5, None, (0.0, 0.0)),
    "calib_shift_n2000_i10k": (0, True, 2000, 0.5, None, (0.2, -0.1)),
    "shared_n2000_i10k": (1, False, 2000, 0.5, "shared", (0.0, 0.0)),
    "varying_n5000_i10k": (2, True, 5000, 0.5, "varying", (0.0, 0.0)),
}


@pytest.mark.parametrize("workload", list(HEADLINE_SHAPES))
def test_headline_oracle_fixture_vs_reference_fixture(golden, workload):
    """Round 5: the oracle's output (headline_<w>.npz) against the REFERENCE binary's (headline_ref_<w>.npz) for all 1024 pairs of each
    timed batch — 4096 full-size reference runs.  Iterations, inlier count and inlier mask identical on every pair; model within 1e-6
    on all but one enumerated pair; the LO count differs exactly on the pairs listed with their cause in headline_ref_deviations.json
    (tests/tools/classify_ref_deviations.py).  Two pairs per workload are re-run through the oracle now (the rest is fixture against fixture)."""
    import hashlib
    import json
    import os
    from mdrp_amd import synth
    o, r = golden(f"headline_{workload}"), golden(f"headline_ref_{workload}")
    js = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "headline_ref_deviations.json")))
    dev = js["deviations"][workload]
    assert (o["digest"] == r["digest"]).all()
    assert (o["istats"][:, 1:] == r["istats"][:, 1:]).all()  # iterations, inliers
    assert (o["mask"] == r["mask"]).all()
    md = np.array([model_diff(o["model"][i], r["model"][i]) for i in range(1024)])
    big = set(np.nonzero(md >= 1e-6)[0].tolist())
    assert big == ({897} if workload == "calib_shift_n2000_i10k" else set()), (workload, big)
    for i in big:
        assert "ref_" in dev[str(i)]["cause"]
    dlo = o["istats"][:, 0] - r["istats"][:, 0]
    assert {str(i): int(dlo[i]) for i in np.nonzero(dlo)[0]} == {k: v["oracle_minus_reference"] for k, v in dev.items() if v["oracle_minus_reference"]}, workload
    assert js["summary"][workload]["lo_count_differs"] == int((dlo != 0).sum())
    sc_off = np.nonzero(~np.isclose(o["fstats"][:, 1], r["fstats"][:, 1], rtol=1e-9, atol=0))[0]
    assert all(str(i) in dev and "ref_" in dev[str(i)]["cause"] for i in sc_off), (workload, sc_off)
    kind, es, n, of, rf, (s1, s2) = HEADLINE_SHAPES[workload]
    cam = po.cam_flat(0, [800.0, 0.0, 0.0])
    for i in (5, 1000):
        p = synth.make_pair(i, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf, shift1=s1, shift2=s2)
        h = hashlib.sha256()
        for key in ("x1", "x2", "d1", "d2"):
            h.update(np.ascontiguousarray(p[key], dtype=np.float64).tobytes())
        assert np.frombuffer(h.digest()[:8], dtype=np.uint64)[0] == r["digest"][i]
        ro = po.ransac_opt(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, max_reproj_error=16.0, estimate_shift=es)
        m, st, mask = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], ro, po.bundle_opt(loss_type=4), cam if kind == 0 else None, cam if kind == 0 else None)
        assert (st.refinements, st.iterations, st.num_inliers) == tuple(int(v) for v in o["istats"][i]), (workload, i)
        assert (np.packbits(mask) == r["mask"][i]).all() and model_diff(m, r["model"][i]) < 1e-6, (workload, i)


DYNAMIC_CASES = {"calib_p3p": (0, False, 2000, None, (0.0, 0.0)), "calib_shift": (0, True, 2000, None, (0.2, -0.1)),
                 "shared": (1, False, 2000, "shared", (0.0, 0.0)), "varying": (2, False, 3000, "varying", (0.0, 0.0))}


def dynamic_pair(g, name, j):
    from mdrp_amd import synth
    kind, es, n, rf, (s1, s2) = DYNAMIC_CASES[name]
    outl = g["outliers"]
    return synth.make_pair(int(g["first_index"]) + j, n, noise_px=0.5, depth_noise=0.02, outlier_frac=float(outl[j % len(outl)]), random_focal=rf, shift1=s1, shift2=s2)


@pytest.mark.parametrize("name", list(DYNAMIC_CASES))
def test_dynamic_stopping_full_size_vs_reference_fixture(golden, name):
    """tests/golden/dynamic_ref.npz (tests/tools/gen_golden_dynamic_ref.py): the reference binary with its default iteration budget (max 100000, min 1000)
    on 96 full-size pairs per estimator at 50-85 % outliers — runs end between 1001 and 28714 iterations by ransac<>'s dynamic rule.  The oracle reproduces
    every eighth pair now: `iterations`, inlier count and mask identical, model within 1e-6, `refinements` equal except on the enumerated solver classes of
    DESIGN.md 5 (counted, not hidden).  The GPU suite runs all 4 x 96 pairs against the same file."""
    import hashlib
    g = golden("dynamic_ref")
    kind, es, n, rf, _ = DYNAMIC_CASES[name]
    cam = po.cam_flat(0, [800.0, 0.0, 0.0])
    ist = g[f"{name}_istats"]
    assert ist[:, 1].min() > 1000 and ist[:, 1].max() < 100000 and len(set(ist[:, 1].tolist())) > 20  # the dynamic rule ended them, at many different counts
    lo_off = 0
    # (pair 27 of the shared-focal set: the reference's solver returns a root ours does not at iteration 552 and ends on another winner —
    # 2 mask bits, 1.3e-4 in the model; enumerated in tests/test_gpu_headline.py, which checks the HIP path against the oracle there)
    for j in range(0, len(ist), 8):
        p = dynamic_pair(g, name, j)
        h = hashlib.sha256()
        for key in ("x1", "x2", "d1", "d2"):
            h.update(np.ascontiguousarray(p[key], dtype=np.float64).tobytes())
        assert np.frombuffer(h.digest()[:8], dtype=np.uint64)[0] == g[f"{name}_digest"][j]
        ro = po.ransac_opt(max_iterations=100000, min_iterations=1000, max_epipolar_error=2.0, max_reproj_error=16.0, estimate_shift=es)
        m, st, mask = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], ro, po.bundle_opt(loss_type=4), cam if kind == 0 else None, cam if kind == 0 else None)
        assert (st.iterations, st.num_inliers) == (int(ist[j, 1]), int(ist[j, 2])), (name, j, st.iterations, ist[j])
        assert (np.packbits(mask) == g[f"{name}_mask"][j]).all() and model_diff(m, g[f"{name}_model"][j]) < 1e-6, (name, j)
        lo_off += int(st.refinements != int(ist[j, 0]))
    assert lo_off <= 2, (name, lo_off)


@pytest.mark.parametrize("workload,kind,pairs,outl,rf,lo_dev", [("relpose_5pt_n2000_i10k", 3, 1024, 0.5, None, 12), ("fundamental_7pt_n2000_i10k", 5, 1024, 0.5, None, 0),
                                                               ("shared_6pt_n2000_i10k", 4, 256, 0.5, "shared", 7), ("calib_p3p_n2000_i10k_clean", 0, 1024, 0.0, None, 0)])
def test_baseline_and_clean_headline_fixtures(golden, workload, kind, pairs, outl, rf, lo_dev):
    """tests/golden/headline_ref_<workload>.npz (tests/tools/gen_golden_headline_ref_classic.py): the reference binary on every pair of the batches bench.py
    times for the 5- / 6- / 7-point baselines and the outlier-free shape, with the oracle run beside it when the fixture was made: oracle result == reference
    result (iterations, inliers, mask, model 1e-6) on EVERY pair; the LO count differs on 12 / 0 / 7 / 0 pairs.  One pair per workload is re-run now."""
    import hashlib
    from mdrp_amd import synth
    g = golden(f"headline_ref_{workload}")
    assert len(g["istats"]) == pairs and g["oracle_same"].all() and float(g["oracle_model_diff"].max()) < 1e-6
    assert int((g["oracle_refinements"] != g["istats"][:, 0]).sum()) == lo_dev
    i = 7
    p = synth.make_pair(i, 2000, noise_px=0.5, depth_noise=0.02, outlier_frac=outl, random_focal=rf)
    h = hashlib.sha256()
    for key in ("x1", "x2", "d1", "d2"):
        h.update(np.ascontiguousarray(p[key], dtype=np.float64).tobytes())
    assert np.frombuffer(h.digest()[:8], dtype=np.uint64)[0] == g["digest"][i]
    cam = po.cam_flat(0, [800.0, 0.0, 0.0])
    if kind == 0:
        ro = po.ransac_opt(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, max_reproj_error=16.0)
        m, st, mask = po.estimate(0, p["x1"], p["x2"], p["d1"], p["d2"], ro, po.bundle_opt(loss_type=4), cam, cam)
    else:
        ro = po.ransac_opt(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0)
        m, st, mask = po.estimate_classic(kind, p["x1"], p["x2"], ro, po.bundle_opt(loss_type=4), cam if kind == 3 else None, cam if kind == 3 else None, pp=(0.0, 0.0))
    assert (st.refinements, st.iterations, st.num_inliers) == (int(g["oracle_refinements"][i]), int(g["istats"][i, 1]), int(g["istats"][i, 2]))
    assert (np.packbits(mask) == g["mask"][i]).all()


def test_model_diff_propagates_nan_in_any_component():
    """NaN poses / NaN scales are legitimate outputs (the reference's P3P); a GPU model with a NaN translation, scale, shift or
    focal must never compare as `< tol` against a finite reference just because its rotation is finite — and the other way round."""
    import numpy as np
    from helpers import model_diff
    ref = np.array([1.0, 0, 0, 0, 0.1, 0.2, 0.3, 1.0, 0.0, 0.0, 1.0, 1.0])
    assert model_diff(ref, ref) == 0.0
    for col in range(12):
        bad = ref.copy(); bad[col] = np.nan
        assert not (model_diff(bad, ref) < 1e-6), col
        assert not (model_diff(ref, bad) < 1e-6), col
