"""Dataset I/O and scoring around the accelerated estimators — the reference's evaluation loop, batched.

The reference's `eval.py` walks an H5 file pair by pair and calls PoseLib once per pair in a process pool
(`eval.py:307-359`).  Here the same file layout is read (any mapping with the same keys works: `h5py.File`, a dict of
arrays), all pairs of one experiment go to the GPU in batches, and the same result records / summary numbers come out:

    corr_{a}_{b}   (N, 32) float   cols 0-3 = x1, y1, x2, y2; depth column pairs as in `utils/data.py:22-46`
    pose_{a}_{b}   (3, 4)          ground-truth [R | t]
    K_{img}        (3, 3)          intrinsics                                               (eval.py:318-336)

Only the calibrated monodepth experiments that map onto the upstream-PoseLib (PR 152) estimators are runnable; fork-only
variants raise NotImplementedError (see `mdrp_amd.poselib._map_fork_options`).  h5py is optional: `open_h5` imports it
lazily.
"""
import json
import time

import numpy as np

from . import _capi

# utils/data.py:22-46 — depth estimator id -> the two columns of corr_* holding (depth in image 1, depth in image 2)
_DEPTH_COLUMNS = {1: (8, 9), 2: (10, 11), 3: (12, 13), 4: (14, 15), 5: (16, 17), 6: (18, 19), 7: (20, 21), 8: (22, 23),
                  9: (24, 25), 10: (26, 27), 11: (28, 29), 12: (30, 31)}
DEPTH_NAMES = {1: "real", 2: "midas", 3: "dpt", 4: "zoe", 5: "depth-any-v1", 6: "depth-any-v2", 7: "depth-pro", 8: "metric3d",
               9: "marigold-e2e", 10: "moge", 11: "marigold", 12: "unidepth"}


def depth_indices(depth):
    """utils/data.py:22-46"""
    return _DEPTH_COLUMNS[int(depth)]


def invalid_depth_mask(d):
    """utils/data.py:13-19 (`get_valid_depth_mask` — despite its name it marks the INVALID rows: inf, nan or negative)"""
    d = np.asarray(d)
    return np.isinf(d[:, 0]) | np.isinf(d[:, 1]) | np.isnan(d[:, 0]) | np.isnan(d[:, 1]) | (d[:, 0] < 0) | (d[:, 1] < 0)


def rotation_error_deg(R_gt, R):
    """utils/data.py:49-61: chordal form, 2 asin(|R_gt - R|_F / (2 sqrt 2))"""
    s = np.linalg.norm(np.asarray(R_gt, dtype=np.float64) - np.asarray(R, dtype=np.float64)) / (2.0 * np.sqrt(2.0))
    return float(np.rad2deg(2.0 * np.arcsin(max(min(1.0, s), -1.0))))


def translation_error_deg(t_gt, t):
    """utils/data.py:64-82: angle between the translation DIRECTIONS, sign-agnostic"""
    t = np.asarray(t, dtype=np.float64).flatten()
    t_gt = np.asarray(t_gt, dtype=np.float64).flatten()
    eps = 1e-15
    t = t / (np.linalg.norm(t) + eps)
    t_gt = t_gt / (np.linalg.norm(t_gt) + eps)
    loss_t = np.maximum(eps, 1.0 - np.sum(t * t_gt) ** 2)
    return float(np.rad2deg(np.arccos(np.sqrt(1.0 - loss_t))))


def open_h5(path):
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - h5py is not part of the build image
        raise ImportError("reading the RePoseD .h5 files needs h5py; any dict-like with the same keys works too") from e
    return h5py.File(path, "r")


def list_pairs(h5, first=None):
    """eval.py:309-314: image-name pairs from the `corr_{a}_o_{b}` keys"""
    prelim = [k.split("corr_")[1] for k in h5.keys() if "corr_" in k]
    pairs = [(p.split("_o_")[0] + "_o", p.split("_o_")[1]) for p in prelim]
    return pairs[:first] if first is not None else pairs


def load_pair(h5, name1, name2, depth=None):
    """eval.py:318-345: keypoints, depth columns (invalid -> 1.0), ground truth, intrinsics of one pair"""
    Rt = np.array(h5[f"pose_{name1}_{name2}"], dtype=np.float64)
    data = np.array(h5[f"corr_{name1}_{name2}"], dtype=np.float64)
    kp1, kp2 = data[:, :2].copy(), data[:, 2:4].copy()
    if depth is not None:
        d = data[:, list(depth_indices(depth))].copy()
    else:
        d = np.ones_like(kp1)
    d[invalid_depth_mask(d)] = 1.0
    return {"kp1": kp1, "kp2": kp2, "d": d, "R_gt": Rt[:3, :3], "t_gt": Rt[:, 3],
            "K1": np.array(h5[f"K_{name1}"], dtype=np.float64), "K2": np.array(h5[f"K_{name2}"], dtype=np.float64)}


def experiment_options(experiment, iters=None, threshold=1.0, reproj_threshold=16.0):
    """eval.py:93-127: option dicts of an experiment name (e.g. '3p_ours_shift_scale+10', 'p3p+12')"""
    lo_iterations = 0 if "nLO" in experiment else 25
    it = 1000 if iters is None else int(iters)
    ro = {"max_iterations": it, "min_iterations": it, "max_epipolar_error": threshold, "progressive_sampling": False,
          "lo_iterations": lo_iterations, "max_reproj_error": reproj_threshold, "all_permutations": True,
          "use_reldepth": "reldepth" in experiment, "use_p3p": "p3p" in experiment, "use_ours": "ours" in experiment,
          "use_madpose": "mad_poselib" in experiment, "solver_shift": "shift" in experiment, "solver_scale": "scale" in experiment,
          "use_reproj": "reproj" in experiment, "optimize_symmetric": "sym_reproj" in experiment,
          "optimize_hybrid": "hybrid" in experiment, "optimize_shift": "reproj-s" in experiment or "hybrid-s" in experiment,
          "use_madpose_shift_optim": "noshift" not in experiment, "weight_sampson": 1.0,
          "graduated_steps": 3 if "GLO" in experiment else 0}
    bo = {"max_iterations": 0 if lo_iterations == 0 else 100, "verbose": False}
    if "truncated" in experiment:
        bo["loss_type"] = "TRUNCATED"
    if "ctruncated" in experiment:
        bo["loss_type"] = "TRUNCATED_CAUCHY"
    return ro, bo


def _pinhole(K):
    return {"model": "PINHOLE", "width": -1, "height": -1, "params": [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]}  # eval.py:129-130


def result_record(experiment, info, R, t, R_gt, t_gt):
    """eval.py:48-67 `get_result_dict`"""
    out = {"R": np.asarray(R).tolist(), "R_gt": np.asarray(R_gt).tolist(), "t": np.asarray(t).tolist(), "t_gt": np.asarray(t_gt).tolist()}
    out["R_err"] = rotation_error_deg(R_gt, R)
    out["t_err"] = translation_error_deg(t_gt, t)
    info = dict(info)
    info["inliers"] = []
    out["info"] = info
    out["experiment"] = experiment
    return out


CORE_RANSAC_KEYS = ("max_iterations", "min_iterations", "dyn_num_trials_mult", "success_prob", "max_reproj_error", "max_epipolar_error",
                    "seed", "progressive_sampling", "max_prosac_iterations", "real_focal_check", "score_initial_model")


def evaluate_calibrated(h5, experiments, iters=None, threshold=1.0, reproj_threshold=16.0, first=None, batch=4096,
                        estimate_batch=None, device=0, estimate_5pt_batch=None):
    """eval.py:316-359 for the calibrated estimator, batched: every experiment's pairs go to the GPU `batch` at a time.
    `estimate_batch(kp1s, kp2s, d1s, d2s, cams1, cams2, ransac_opt, bundle_opt)` defaults to the accelerated
    `poselib.estimate_monodepth_relative_pose_batch` (injectable for tests).  Experiment names with '5p' are the 5-point
    baseline row (eval.py:134-137: `poselib.estimate_relative_pose`, which reads only the upstream RansacOptions keys of the
    dict) and go to `poselib.estimate_relative_pose_batch`.  Pairs with fewer than 5 correspondences are skipped as in the
    reference.  `info['runtime']` is the batch wall time divided by the batch size, in ms."""
    from . import poselib
    if estimate_batch is None:
        def estimate_batch(k1, k2, a, b, c1, c2, ro, bo):
            return poselib.estimate_monodepth_relative_pose_batch(k1, k2, a, b, c1, c2, ro, bo, device=device)
    if estimate_5pt_batch is None:
        def estimate_5pt_batch(k1, k2, c1, c2, ro, bo):
            return poselib.estimate_relative_pose_batch(k1, k2, c1, c2, ro, bo, device=device)
    pairs = list_pairs(h5, first)
    results = []
    for experiment in experiments:
        depth = int(experiment.split("+")[1]) if "+" in experiment else None
        ro, bo = experiment_options(experiment, iters, threshold, reproj_threshold)
        five_point = "5p" in experiment
        ro = {k: v for k, v in ro.items() if k in CORE_RANSAC_KEYS} if five_point else poselib._map_fork_options(ro)
        loaded = []
        for a, b in pairs:
            p = load_pair(h5, a, b, depth)
            if len(p["kp1"]) >= 5:
                loaded.append(p)
        for s in range(0, len(loaded), batch):
            chunk = loaded[s:s + batch]
            t0 = time.perf_counter()
            if five_point:
                geoms, infos = estimate_5pt_batch([p["kp1"] for p in chunk], [p["kp2"] for p in chunk], [_pinhole(p["K1"]) for p in chunk],
                                                  [_pinhole(p["K2"]) for p in chunk], ro, bo)
            else:
                geoms, infos = estimate_batch([p["kp1"] for p in chunk], [p["kp2"] for p in chunk], [p["d"][:, 0] for p in chunk],
                                              [p["d"][:, 1] for p in chunk], [_pinhole(p["K1"]) for p in chunk],
                                              [_pinhole(p["K2"]) for p in chunk], ro, bo)
            ms = 1000.0 * (time.perf_counter() - t0) / max(len(chunk), 1)
            for p, g, info in zip(chunk, geoms, infos):
                info = dict(info)
                info["runtime"] = ms
                pose = g if five_point else g.pose   # estimate_relative_pose returns the CameraPose itself
                results.append(result_record(experiment, info, pose.R, pose.t, p["R_gt"], p["t_gt"]))
    return results


def summarize(experiments, results):
    """utils/eval_utils.py:41-66 `print_results`, as rows: (experiment, median pose error, pose mAA over 1..10 degrees,
    mean runtime ms, mean inlier ratio).  NaN errors count as 180 degrees."""
    rows = []
    for exp in experiments:
        rs = [x for x in results if x["experiment"] == exp]
        if not rs:
            continue
        p = np.array([max(r["R_err"], r["t_err"]) for r in rs], dtype=np.float64)
        p[np.isnan(p)] = 180.0
        maa = float(np.mean([np.sum(p < t) / len(p) for t in range(1, 11)]))
        rows.append((exp, float(np.median(p)), maa, float(np.mean([x["info"]["runtime"] for x in rs])),
                     float(np.mean([x["info"]["inlier_ratio"] for x in rs]))))
    return rows


def format_table(rows):
    lines = [f"{'solver':32s} {'median pose err':>16s} {'pose mAA':>10s} {'mean time':>10s} {'mean inliers':>13s}"]
    for r in rows:
        lines.append(f"{r[0]:32s} {r[1]:16.4f} {r[2]:10.4f} {r[3]:10.4f} {r[4]:13.4f}")
    return "\n".join(lines)


def write_results(path, results):
    """eval.py:378-379: the `results_new/calibrated-*.json` list of records"""
    with open(path, "w") as f:
        json.dump(results, f)


# ------------------------------------------------------------------------------------------------ focal loops
def focal_options(experiment, iters=None, threshold=1.0, reproj_threshold=16.0, varying=False):
    """eval_shared_f.py:111-152 / eval_varying_f.py:105-146"""
    lo_iterations = 0 if "nLO" in experiment else 25
    it = 1000 if iters is None else int(iters)
    ro = {"max_iterations": it, "min_iterations": it, "max_epipolar_error": threshold, "progressive_sampling": False,
          "lo_iterations": lo_iterations, "max_reproj_error": reproj_threshold,
          "use_p3p": "p3p" in experiment, "use_ours": "ours" in experiment, "use_madpose": "mad_poselib" in experiment,
          "solver_shift": "shift" in experiment, "solver_scale": "scale" in experiment, "use_reproj": "reproj" in experiment,
          "optimize_shift": "reproj-s" in experiment, "use_madpose_shift_optim": "noshift" not in experiment,
          "graduated_steps": 3 if "GLO" in experiment else 0}
    ro.update({"optimize_hybrid": "hybrid" in experiment, "sym_repro": "sym_reproj" in experiment, "no_normalization": "NN" in experiment})
    if varying:  # eval_varying_f.py:134-136
        ro.update({"use_fundamental": "7p" in experiment, "use_4p4d": "4p4d" in experiment, "use_eigen": "eigen" in experiment})
    else:        # eval_shared_f.py:130-131
        ro.update({"all_permutations": "perm" in experiment, "use_reldepth": "reldepth" in experiment})
    bo = {"max_iterations": 0 if lo_iterations == 0 else 100, "verbose": False}
    if "truncated" in experiment:
        bo["loss_type"] = "TRUNCATED"
    if "ctruncated" in experiment:
        bo["loss_type"] = "TRUNCATED_CAUCHY"
    return ro, bo


def load_pair_focal(h5, name1, name2, depth=None, shared=True):
    """eval_shared_f.py:330-360 / eval_varying_f.py:329-357: principal points removed; for the shared-focal loop the second
    image is rescaled to the first one's focal length.  Depth columns are passed as stored (no invalid -> 1.0 here)."""
    Rt = np.array(h5[f"pose_{name1}_{name2}"], dtype=np.float64)
    K1 = np.array(h5[f"K_{name1}"], dtype=np.float64)
    K2 = np.array(h5[f"K_{name2}"], dtype=np.float64)
    data = np.array(h5[f"corr_{name1}_{name2}"], dtype=np.float64)
    kp1 = data[:, :2] - K1[:2, 2]
    kp2 = data[:, 2:4] - K2[:2, 2]
    if shared and (K1[0, 0] + K1[1, 1]) != (K2[0, 0] + K2[1, 1]):
        r = (K1[0, 0] + K1[1, 1]) / (K2[0, 0] + K2[1, 1])
        kp2 = kp2 * r
        K2 = r * K2
    d = data[:, list(depth_indices(depth))].copy() if depth is not None else np.ones_like(kp1)
    return {"kp1": kp1, "kp2": kp2, "d": d, "R_gt": Rt[:3, :3], "t_gt": Rt[:, 3], "K1": K1, "K2": K2}


def result_record_focal(experiment, info, pair, R_gt, t_gt, f1_gt, f2_gt):
    """eval_shared_f.py:80-108 `get_result_dict`"""
    out = result_record(experiment, info, pair.pose.R, pair.pose.t, R_gt, t_gt)
    out["f1_gt"], out["f1"] = float(f1_gt), float(pair.camera1.focal())
    out["f2_gt"], out["f2"] = float(f2_gt), float(pair.camera2.focal())
    out["f1_err"] = abs(out["f1"] - f1_gt) / f1_gt
    out["f2_err"] = abs(out["f2"] - f2_gt) / f2_gt
    out["f_err"] = float(np.sqrt(out["f1_err"] * out["f2_err"]))
    return out


def evaluate_focal(h5, experiments, shared=True, iters=None, threshold=1.0, reproj_threshold=16.0, first=None, batch=4096,
                   estimate_batch=None, device=0):
    """eval_shared_f.py / eval_varying_f.py main loop for the monodepth focal estimators, batched.  Pairs with fewer than
    6 (shared) / 7 (varying) correspondences are skipped as in the reference."""
    from . import poselib
    if estimate_batch is None:
        fn = poselib.estimate_monodepth_shared_focal_relative_pose_batch if shared else poselib.estimate_monodepth_varying_focal_relative_pose_batch

        def estimate_batch(k1, k2, a, b, ro, bo):
            return fn(k1, k2, a, b, ro, bo, device=device)
    min_n = 6 if shared else 7
    pairs = list_pairs(h5, first)
    results = []
    for experiment in experiments:
        depth = int(experiment.split("+")[1]) if "+" in experiment else None
        ro, bo = focal_options(experiment, iters, threshold, reproj_threshold, varying=not shared)
        six_point = shared and "6p" in experiment  # eval_shared_f.py:159-162: the non-monodepth 6-point row
        ro = {k: v for k, v in ro.items() if k in CORE_RANSAC_KEYS} if six_point else \
            poselib._map_fork_options(ro, _capi.SHARED_FOCAL if shared else _capi.VARYING_FOCAL)
        loaded = [p for p in (load_pair_focal(h5, a, b, depth, shared) for a, b in pairs) if len(p["kp1"]) >= min_n]
        for s in range(0, len(loaded), batch):
            chunk = loaded[s:s + batch]
            t0 = time.perf_counter()
            if six_point:
                out, infos = poselib.estimate_shared_focal_relative_pose_batch([p["kp1"] for p in chunk], [p["kp2"] for p in chunk], None, ro, bo,
                                                                               device=device)
            else:
                out, infos = estimate_batch([p["kp1"] for p in chunk], [p["kp2"] for p in chunk], [p["d"][:, 0] for p in chunk],
                                            [p["d"][:, 1] for p in chunk], ro, bo)
            ms = 1000.0 * (time.perf_counter() - t0) / max(len(chunk), 1)
            for p, ip, info in zip(chunk, out, infos):
                info = dict(info)
                info["runtime"] = ms
                f1_gt = (p["K1"][0, 0] + p["K1"][1, 1]) / 2
                f2_gt = (p["K2"][0, 0] + p["K2"][1, 1]) / 2
                results.append(result_record_focal(experiment, info, ip, p["R_gt"], p["t_gt"], f1_gt, f2_gt))
    return results


def summarize_focal(experiments, results):
    """utils/eval_utils.py:8-38 `print_results_focal`: (experiment, median pose error, median focal error, pose mAA over
    1..10 degrees, focal mAA over 1..10 %, mean runtime ms, mean inlier ratio)"""
    rows = []
    for exp in experiments:
        rs = [x for x in results if x["experiment"] == exp]
        if not rs:
            continue
        p = np.array([max(r["R_err"], r["t_err"]) for r in rs], dtype=np.float64)
        f = np.array([r["f_err"] for r in rs], dtype=np.float64)
        p[np.isnan(p)] = 180.0
        f[np.isnan(f)] = 1.0
        rows.append((exp, float(np.median(p)), float(np.median(f)),
                     float(np.mean([np.sum(p < t) / len(p) for t in range(1, 11)])),
                     float(np.mean([np.sum(f < t / 100) / len(f) for t in range(1, 11)])),
                     float(np.mean([x["info"]["runtime"] for x in rs])), float(np.mean([x["info"]["inlier_ratio"] for x in rs]))))
    return rows
