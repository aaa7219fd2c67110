"""mdrp_amd/evalio.py — the reference's dataset loop (eval.py:307-379) around the batched estimator.
CPU: helpers against golden vectors produced by the reference's own utils/data.py (tests/tools/gen_golden_evalio.py),
the H5 layout / option mapping / result records with a stub estimator.  GPU: end to end on a synthetic H5-like dict."""
import json
import os

import numpy as np
import pytest

from mdrp_amd import evalio, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_helpers_match_reference_golden():
    g = np.load(os.path.join(HERE, "golden", "evalio.npz"))
    for i in range(len(g["R"])):
        assert evalio.rotation_error_deg(g["R_gt"][i], g["R"][i]) == pytest.approx(g["R_err"][i], rel=1e-12, abs=1e-12)
        assert evalio.translation_error_deg(g["t_gt"][i], g["t"][i]) == pytest.approx(g["t_err"][i], rel=1e-12, abs=1e-12)
    assert np.array_equal(evalio.invalid_depth_mask(g["d"]), g["invalid_mask"])
    assert [list(evalio.depth_indices(k)) for k in range(1, 13)] == g["depth_columns"].tolist()


def fake_h5(n_pairs=6, n=300, depth=10, seed=0):
    """a dict with the key layout of the RePoseD .h5 files (eval.py:307-336), built from synthetic pairs"""
    h5 = {}
    gt = []
    for i in range(n_pairs):
        p = synth.make_pair(500 + seed + i, n if i != 2 else 4, noise_px=0.3, depth_noise=0.01, outlier_frac=0.3 if i % 2 else 0.0,
                            pp=(640.0, 480.0), f1=900.0, f2=700.0)
        a, b = f"img{i:02d}a_o", f"img{i:02d}b"
        data = np.zeros((len(p["x1"]), 32))
        data[:, :2] = p["x1"]; data[:, 2:4] = p["x2"]
        c1, c2 = evalio.depth_indices(depth)
        data[:, c1] = p["d1"]; data[:, c2] = p["d2"]
        if len(data) > 10:
            data[3, c1] = np.inf; data[5, c2] = np.nan; data[7, c1] = -1.0   # invalid depths -> 1.0 (eval.py:344-346)
        h5[f"corr_{a}_{b}"] = data
        h5[f"pose_{a}_{b}"] = np.c_[p["R"], p["t"]]
        h5[f"K_{a}"] = np.array([[900.0, 0, 640.0], [0, 900.0, 480.0], [0, 0, 1]])
        h5[f"K_{b}"] = np.array([[700.0, 0, 640.0], [0, 700.0, 480.0], [0, 0, 1]])
        gt.append(p)
    return h5, gt


def test_layout_options_and_records_with_stub_estimator(tmp_path):
    h5, gt = fake_h5()
    pairs = evalio.list_pairs(h5)
    assert pairs[0] == ("img00a_o", "img00b") and len(pairs) == 6 and len(evalio.list_pairs(h5, first=2)) == 2
    p = evalio.load_pair(h5, *pairs[0], depth=10)
    assert p["d"][3, 0] == 1.0 and p["d"][5, 1] == 1.0 and p["d"][7, 0] == 1.0 and p["kp1"].shape == (300, 2)
    assert np.all(evalio.load_pair(h5, *pairs[0])["d"] == 1.0)
    ro, bo = evalio.experiment_options("3p_ours_shift_scale_hybrid_ctruncated+10", iters=2000, threshold=1.5, reproj_threshold=12.0)
    assert ro["max_iterations"] == ro["min_iterations"] == 2000 and ro["max_epipolar_error"] == 1.5 and ro["max_reproj_error"] == 12.0
    assert ro["use_ours"] and ro["solver_shift"] and ro["solver_scale"] and ro["optimize_hybrid"] and not ro["use_p3p"]
    assert bo == {"max_iterations": 100, "verbose": False, "loss_type": "TRUNCATED_CAUCHY"}
    assert evalio.experiment_options("p3p_nLO+1")[1]["max_iterations"] == 0

    seen = {}

    def stub(k1, k2, d1, d2, c1, c2, ro_, bo_):   # returns the ground truth of the pair it was handed
        from mdrp_amd.poselib import CameraPose, MonoDepthTwoViewGeometry
        seen["ro"], seen["n"] = ro_, len(k1)
        out = []
        for kp in k1:
            g = next(q for q in gt if len(q["x1"]) == len(kp) and np.allclose(q["x1"], kp))
            geom = MonoDepthTwoViewGeometry.__new__(MonoDepthTwoViewGeometry)
            pose = CameraPose.__new__(CameraPose)
            pose.__dict__.update(_R=g["R"], _t=g["t"])
            geom.pose, geom.scale, geom.shift1, geom.shift2 = type("P", (), {"R": g["R"], "t": g["t"]})(), 1.0, 0.0, 0.0
            out.append(geom)
        infos = [{"refinements": 1, "iterations": 10, "num_inliers": len(kp), "inlier_ratio": 1.0, "model_score": 0.0, "inliers": [True] * len(kp)}
                 for kp in k1]
        return out, infos

    exps = ["3p_ours_shift_scale_hybrid-s+10", "p3p_hybrid+10"]
    res = evalio.evaluate_calibrated(h5, exps, iters=500, batch=4, estimate_batch=stub)
    assert len(res) == 2 * 5                      # the 4-correspondence pair is skipped (eval.py:338-339)
    assert seen["ro"]["monodepth_estimate_shift"] is False and seen["n"] == 1    # last call: p3p, second batch of 5 pairs
    r0 = res[0]
    assert set(r0) == {"R", "R_gt", "t", "t_gt", "R_err", "t_err", "info", "experiment"} and r0["info"]["inliers"] == []
    assert r0["R_err"] < 1e-6 and r0["t_err"] < 1e-4 and r0["experiment"] == exps[0] and "runtime" in r0["info"]
    rows = evalio.summarize(exps, res)
    assert rows[0][0] == exps[0] and rows[0][2] == 1.0 and rows[0][4] == 1.0        # every pose error below 1 degree
    res[0]["R_err"] = float("nan"); res[1]["t_err"] = 7.5
    rows = evalio.summarize(exps, res)
    assert rows[0][2] == pytest.approx((7 * 3 / 5 + 3 * 4 / 5) / 10)                  # nan -> 180 never passes; 7.5 deg passes t = 8, 9, 10
    assert "pose mAA" in evalio.format_table(rows)
    out = tmp_path / "calibrated-test.json"
    evalio.write_results(out, res)
    back = json.load(open(out))
    assert len(back) == len(res) and back[2]["experiment"] == exps[0]
    with pytest.raises(NotImplementedError):
        evalio.evaluate_calibrated(h5, ["3p_reldepth+10"], estimate_batch=stub)       # fork-only variant

    # the 5-point baseline row (eval.py:134-137): routed to estimate_relative_pose with the upstream RansacOptions keys only
    def stub5(k1, k2, c1, c2, ro_, bo_):
        seen["ro5"], seen["cams5"] = ro_, (c1[0], c2[0])
        poses = [type("P", (), {"R": q["R"], "t": q["t"]})() for kp in k1 for q in gt if len(q["x1"]) == len(kp) and np.allclose(q["x1"], kp)]
        return poses, [{"refinements": 1, "iterations": 10, "num_inliers": len(kp), "inlier_ratio": 1.0, "model_score": 0.0, "inliers": []} for kp in k1]

    res5 = evalio.evaluate_calibrated(h5, ["5p+10"], iters=700, threshold=1.5, batch=8, estimate_batch=stub, estimate_5pt_batch=stub5)
    assert len(res5) == 5 and res5[0]["R_err"] < 1e-6 and res5[0]["experiment"] == "5p+10"
    assert set(seen["ro5"]) <= set(evalio.CORE_RANSAC_KEYS) and seen["ro5"]["max_iterations"] == 700 and seen["ro5"]["max_epipolar_error"] == 1.5
    assert seen["cams5"][0]["model"] == "PINHOLE" and seen["cams5"][0]["params"][0] == 900.0 and seen["cams5"][1]["params"][0] == 700.0


@pytest.mark.gpu
def test_evaluate_calibrated_end_to_end_gpu():
    h5, gt = fake_h5(n_pairs=8, n=400, seed=40)
    exps = ["p3p_hybrid_ctruncated+10", "3p_ours_shift_scale_hybrid-s_ctruncated+10", "5p_ctruncated+10"]
    res = evalio.evaluate_calibrated(h5, exps, iters=1000, threshold=2.0)
    assert len(res) == 3 * 7
    rows = evalio.summarize(exps, res)
    for exp, med, maa, ms, inl in rows:
        assert med < 0.5 and maa > 0.9 and ms > 0 and 0.5 < inl <= 1.0, rows
    # every per-pair record against the CPU oracle run on the same pair with the options eval.py would build
    # (eval.py:93-160): pose, stats and both error metrics (the 5-point rows: direction of t, its norm is a gauge)
    from oracle import pyorc as po
    pairs = [evalio.load_pair(h5, a, b, depth=10) for a, b in evalio.list_pairs(h5)]
    pairs = [p for p in pairs if len(p["kp1"]) >= 5]
    assert len(pairs) == 7
    for e, exp in enumerate(exps):
        ro, bo = evalio.experiment_options(exp, iters=1000, threshold=2.0)
        for j, p in enumerate(pairs):
            rec = res[e * 7 + j]
            c1 = po.cam_flat(1, [p["K1"][0, 0], p["K1"][1, 1], p["K1"][0, 2], p["K1"][1, 2]])
            c2 = po.cam_flat(1, [p["K2"][0, 0], p["K2"][1, 1], p["K2"][0, 2], p["K2"][1, 2]])
            oro = po.ransac_opt(max_iterations=1000, min_iterations=1000, max_epipolar_error=2.0, max_reproj_error=16.0,
                                estimate_shift=exp.startswith("3p_ours_shift"))
            obo = po.bundle_opt(loss_type={"TRUNCATED": 1, "TRUNCATED_CAUCHY": 4, "CAUCHY": 3}[bo["loss_type"]], loss_scale=bo.get("loss_scale", 1.0))
            if exp.startswith("5p"):
                m, st, mk = po.estimate_classic(3, p["kp1"], p["kp2"], oro, obo, c1, c2)
            else:
                m, st, mk = po.estimate(po.CALIB, p["kp1"], p["kp2"], p["d"][:, 0], p["d"][:, 1], oro, obo, c1, c2)
            Ro = po.quat_to_rotmat(m[:4])
            where = (exp, j)
            assert rec["info"]["num_inliers"] == st.num_inliers and rec["info"]["iterations"] == st.iterations == 1000, where
            assert rec["info"]["refinements"] == st.refinements, where
            assert np.abs(np.array(rec["R"]) - Ro).max() < 1e-6, where
            to, tg = m[4:7], np.array(rec["t"])
            if exp.startswith("5p"):
                assert np.abs(tg / np.linalg.norm(tg) - to / np.linalg.norm(to)).max() < 1e-6, where
            else:
                assert np.abs(tg - to).max() < 1e-6 * (1 + np.abs(to).max()), where
            assert rec["R_err"] == pytest.approx(evalio.rotation_error_deg(p["R_gt"], Ro), abs=1e-6)
            assert rec["t_err"] == pytest.approx(evalio.translation_error_deg(p["t_gt"], to), abs=1e-5)


def fake_h5_focal(n_pairs=6, n=400, depth=12, seed=0, varying=False):
    h5 = {}
    for i in range(n_pairs):
        p = synth.make_pair(800 + seed + i, n if i != 1 else 5, noise_px=0.3, depth_noise=0.01, outlier_frac=0.25,
                            random_focal="varying" if varying else "shared", pp=(0.0, 0.0))
        a, b = f"f{i:02d}a_o", f"f{i:02d}b"
        pp1, pp2 = np.array([512.0, 384.0]), np.array([500.0, 400.0])
        data = np.zeros((len(p["x1"]), 32))
        data[:, :2] = p["x1"] + pp1; data[:, 2:4] = p["x2"] + pp2
        c1, c2 = evalio.depth_indices(depth)
        data[:, c1] = p["d1"]; data[:, c2] = p["d2"]
        h5[f"corr_{a}_{b}"] = data
        h5[f"pose_{a}_{b}"] = np.c_[p["R"], p["t"]]
        h5[f"K_{a}"] = np.array([[p["f1"], 0, pp1[0]], [0, p["f1"], pp1[1]], [0, 0, 1]])
        h5[f"K_{b}"] = np.array([[p["f2"], 0, pp2[0]], [0, p["f2"], pp2[1]], [0, 0, 1]])
    return h5


def test_focal_loader_and_summary():
    h5 = fake_h5_focal()
    a, b = evalio.list_pairs(h5)[0]
    p = evalio.load_pair_focal(h5, a, b, depth=12, shared=True)
    assert abs(p["kp1"]).max() < 1700 and p["kp1"].shape == (400, 2) and np.allclose(p["K1"][0, 0], p["K2"][0, 0])
    K2 = np.array(h5[f"K_{b}"]); h5[f"K_{b}"] = 2.0 * K2        # different focal: second image rescaled (eval_shared_f.py:351-353)
    h5[f"K_{b}"][2, 2] = 1.0
    q = evalio.load_pair_focal(h5, a, b, depth=12, shared=True)
    raw2 = np.array(h5[f"corr_{a}_{b}"])[:, 2:4] - h5[f"K_{b}"][:2, 2]
    assert np.allclose(q["kp2"], 0.5 * raw2) and np.allclose(q["K2"][0, 0], q["K1"][0, 0])
    assert np.allclose(evalio.load_pair_focal(h5, a, b, depth=12, shared=False)["kp2"], raw2)
    ro, bo = evalio.focal_options("3p_ours_scale_hybrid_ctruncated+12", iters=300, varying=False)
    assert ro["use_ours"] and ro["optimize_hybrid"] and "use_4p4d" not in ro and bo["loss_type"] == "TRUNCATED_CAUCHY"
    assert evalio.focal_options("4p4d+1", varying=True)[0]["use_4p4d"]
    recs = [{"experiment": "e", "R_err": 0.5, "t_err": 2.5, "f_err": 0.035, "info": {"runtime": 1.0, "inlier_ratio": 0.5}},
            {"experiment": "e", "R_err": float("nan"), "t_err": 0.1, "f_err": float("nan"), "info": {"runtime": 3.0, "inlier_ratio": 0.7}}]
    (row,) = evalio.summarize_focal(["e"], recs)
    assert row[1] == pytest.approx((2.5 + 180) / 2) and row[2] == pytest.approx((0.035 + 1.0) / 2)
    assert row[3] == pytest.approx(8 * 0.5 / 10) and row[4] == pytest.approx(7 * 0.5 / 10) and row[5] == 2.0 and row[6] == pytest.approx(0.6)


@pytest.mark.gpu
@pytest.mark.parametrize("shared", [True, False])
def test_evaluate_focal_end_to_end_gpu(shared):
    h5 = fake_h5_focal(n_pairs=8, n=500, seed=20, varying=not shared)
    exps = ["3p_ours_scale_hybrid_ctruncated+12"]
    res = evalio.evaluate_focal(h5, exps, shared=shared, iters=1000, threshold=2.0)
    assert len(res) == 7 and {"f1", "f2", "f_err", "f1_gt"} <= set(res[0])
    (row,) = evalio.summarize_focal(exps, res)
    assert row[1] < 1.0 and row[2] < 0.02 and row[3] > 0.85 and row[4] > 0.85, row
    # every per-pair record (R, t, focals, stats) against the CPU oracle on the same centred pixels (eval_shared_f.py:111-183)
    from oracle import pyorc as po
    pairs = [evalio.load_pair_focal(h5, a, b, depth=12, shared=shared) for a, b in evalio.list_pairs(h5)]
    pairs = [p for p in pairs if len(p["kp1"]) >= (6 if shared else 7)]
    assert len(pairs) == len(res)
    ro, bo = evalio.focal_options(exps[0], iters=1000, threshold=2.0, varying=not shared)
    oro = po.ransac_opt(max_iterations=1000, min_iterations=1000, max_epipolar_error=2.0, max_reproj_error=16.0)
    obo = po.bundle_opt(loss_type={"TRUNCATED": 1, "TRUNCATED_CAUCHY": 4, "CAUCHY": 3}[bo["loss_type"]], loss_scale=bo.get("loss_scale", 1.0))
    for j, p in enumerate(pairs):
        m, st, mk = po.estimate(po.SHARED if shared else po.VARYING, p["kp1"], p["kp2"], p["d"][:, 0], p["d"][:, 1], oro, obo)
        rec = res[j]
        assert rec["info"]["num_inliers"] == st.num_inliers and rec["info"]["iterations"] == st.iterations and rec["info"]["refinements"] == st.refinements, j
        assert np.abs(np.array(rec["R"]) - po.quat_to_rotmat(m[:4])).max() < 1e-6, j
        assert np.abs(np.array(rec["t"]) - m[4:7]).max() < 1e-6 * (1 + np.abs(m[4:7]).max()), j
        assert rec["f1"] == pytest.approx(m[10], rel=1e-6) and rec["f2"] == pytest.approx(m[11], rel=1e-6), j
    if shared:  # the 6-point row of the shared-focal tables (eval_shared_f.py:159-162): no depths, upstream RansacOptions keys only
        res6 = evalio.evaluate_focal(h5, ["6p"], shared=True, iters=1000, threshold=2.0)
        assert len(res6) == len(pairs)
        for j, p in enumerate(pairs):
            m, st, mk = po.estimate_classic(4, p["kp1"], p["kp2"], oro, po.bundle_opt(loss_type=3), pp=(0.0, 0.0))
            rec = res6[j]
            assert rec["info"]["num_inliers"] == st.num_inliers and rec["info"]["iterations"] == st.iterations, j
            assert np.abs(np.array(rec["R"]) - po.quat_to_rotmat(m[:4])).max() < 1e-6 and rec["f1"] == pytest.approx(m[10], rel=1e-6), j
        assert np.median([r["R_err"] for r in res6]) < 1.0 and np.median([r["f_err"] for r in res6]) < 0.05


def test_baseline_options_that_are_not_built_raise():
    import mdrp_amd.poselib as poselib
    x = np.zeros((10, 2))
    with pytest.raises(NotImplementedError):
        poselib.estimate_fundamental(x, x, {"real_focal_check": True}, {})
    with pytest.raises(NotImplementedError):
        poselib.estimate_relative_pose(x, x, {"model": "SIMPLE_PINHOLE", "width": 1, "height": 1, "params": [1.0, 0, 0]}, {"model": "SIMPLE_PINHOLE", "width": 1, "height": 1, "params": [1.0, 0, 0]},
                                       {"progressive_sampling": True}, {})


def test_fork_flag_table_covers_every_experiment_name():
    """Every experiment family of the reference's lists (utils/data.py:86-200 get_experiments, eval.py:212-239) goes through
    the option builders and mdrp_amd.poselib._map_fork_options: the ones that ARE the released PR-152 estimators map, with the
    right monodepth_estimate_shift; every other one raises NotImplementedError — none runs the default estimator silently."""
    from mdrp_amd import _capi, poselib
    calib_ok = {"3p_ours_shift_scale_hybrid-s+10": True, "3p_ours_shift_scale_hybrid-s_truncated+10": True,
                "3p_ours_shift_scale_hybrid-s_ctruncated+6": True, "p3p_hybrid+12": False, "p3p_hybrid_ctruncated+1": False}
    calib_fork = ["3p_reldepth+10", "3p_ours_shift_scale+10", "3p_ours_shift_scale_reproj+10", "3p_ours_shift_scale_sym_reproj+10",
                  "3p_ours_shift_scale_reproj-s+10", "3p_ours_shift_scale_reproj-sfix+10", "3p_ours_shift_scale_hybrid+10",
                  "3p_ours_shift_scale_hybrid_reproj+10", "3p_ours_shift_scale_hybrid-s_reproj+10", "p3p+10", "p3p_reproj+10",
                  "p3p_reproj-s+10", "p3p_sym_reproj+10", "p3p_hybrid_reproj+10", "p3p_hybrid-s+10", "p3p_hybrid-s_reproj+10",
                  "mad_poselib_shift_scale+10", "mad_poselib_shift_scale_reproj+10", "mad_poselib_shift_scale_reproj-s+10",
                  "3p_ours_scale_hybrid+10", "3p_ours_hybrid+10", "3p_ours_shift_scale_hybrid-s_GLO+10", "p3p_hybrid_nLO+10"]
    for name, shift in calib_ok.items():
        ro = poselib._map_fork_options(evalio.experiment_options(name)[0], _capi.CALIB)
        assert ro["monodepth_estimate_shift"] is shift and ro["monodepth_weight_sampson"] == 1.0, name
    for name in calib_fork:
        with pytest.raises(NotImplementedError):
            poselib._map_fork_options(evalio.experiment_options(name)[0], _capi.CALIB)
    for varying in (False, True):
        kind = _capi.VARYING_FOCAL if varying else _capi.SHARED_FOCAL
        for name in ("3p_ours_scale_hybrid+10", "3p_ours_scale_hybrid_ctruncated+12"):
            ro = poselib._map_fork_options(evalio.focal_options(name, varying=varying)[0], kind)
            assert "monodepth_estimate_shift" not in ro or not ro["monodepth_estimate_shift"]
        fork = ["3p_ours_scale+10", "3p_ours_scale_reproj+10", "4p_ours_scale_shift+10", "4p_ours_scale_shift_reproj+10", "3p_reldepth+10",
                "mad_poselib_shift_scale+10", "3p_ours_scale_hybrid_sym_reproj+10", "3p_ours_scale_hybrid_NN+10", "3p_ours_scale_hybrid_GLO+10",
                "p3p_hybrid+10"]
        fork += ["4p4d+10", "7p", "3p_ours_scale_hybrid_eigen+10"] if varying else ["3p_ours_scale_hybrid_perm+10"]
        for name in fork:
            with pytest.raises(NotImplementedError):
                poselib._map_fork_options(evalio.focal_options(name, varying=varying)[0], kind)
    # plain PR-152 dicts (make_pair.py:31-33, notebook) are not touched
    plain = {"max_epipolar_error": 2.0, "max_reproj_error": 16.0, "lo_iterations_typo": 3, "monodepth_estimate_shift": True}
    assert poselib._map_fork_options(plain) == plain
