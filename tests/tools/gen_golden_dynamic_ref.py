#!/usr/bin/env python3
"""REFERENCE-BINARY fixture for the DYNAMIC stopping rule at full size (tests/golden/dynamic_ref.npz): the reference's PoseLib binary on 96 pairs
per estimator with the reference's default iteration budget (max_iterations = 100000, min_iterations = 1000, success_prob = 0.9999,
dyn_num_trials_mult = 3) and outlier fractions 0.5 / 0.6 / 0.7 / 0.75 / 0.8 / 0.85, so that ransac<>'s `iterations > dynamic_max_iter` test ends the runs
anywhere between 1000 and ~30000 iterations — several super-chunks of the HIP path, every one ending in a host read-back of the walk's verdict.
Stored: seeds + outputs only (`iterations`, `refinements`, `num_inliers`, `inlier_ratio`, `model_score`, the 12-wide model, packed mask, input digest).

Runs only in the build container:   bash oracle/build_ref.sh && python3 tests/tools/gen_golden_dynamic_ref.py     (8 workers, ~1 minute)"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import gen_golden_headline_ref as gh  # noqa: E402

CASES = {
    # name: kind, estimate_shift, n, random_focal, depth shifts
    "calib_p3p": (0, False, 2000, None, (0.0, 0.0)),
    "calib_shift": (0, True, 2000, None, (0.2, -0.1)),
    "shared": (1, False, 2000, "shared", (0.0, 0.0)),
    "varying": (2, False, 3000, "varying", (0.0, 0.0)),
}
OUTLIERS = (0.5, 0.6, 0.7, 0.75, 0.8, 0.85)
PAIRS = 96
FIRST = 20000  # synth indices 20000 .. 20095: none of the other fixtures' pairs
OPTS = dict(max_iterations=100000, min_iterations=1000, max_epipolar_error=2.0, max_reproj_error=16.0, seed=0)


def make_pair(name, j):
    from mdrp_amd import synth
    kind, es, n, rf, (s1, s2) = CASES[name]
    return synth.make_pair(FIRST + j, n, noise_px=0.5, depth_noise=0.02, outlier_frac=OUTLIERS[j % len(OUTLIERS)], random_focal=rf, shift1=s1, shift2=s2)


def _work(args):
    name, j = args
    import refshim as rs
    kind, es, n, rf, _ = CASES[name]
    cam = rs.cam_flat(0, 1600, 1200, [800.0, 0.0, 0.0])
    p = make_pair(name, j)
    gh._srand(1)
    m, st, mask = rs.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], rs.ropt(estimate_shift=es, **OPTS), rs.bopt(loss_type=4),
                              cam if kind == 0 else None, cam if kind == 0 else None)
    m12 = np.r_[m, 1.0, 1.0] if kind == 0 else np.asarray(m)
    return name, j, m12, (int(st[0]), int(st[1]), int(st[2])), (float(st[3]), float(st[4])), np.packbits(mask), gh.input_digest(p)


def main():
    jobs = [(name, j) for name in CASES for j in range(PAIRS)]
    with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
        rows = pool.map(_work, jobs, chunksize=2)
    d = {"names": np.array(list(CASES)), "first_index": np.array(FIRST), "outliers": np.array(OUTLIERS),
         "cases": np.array([[CASES[k][0], int(CASES[k][1]), CASES[k][2]] for k in CASES])}
    for name in CASES:
        rs_ = sorted((r for r in rows if r[0] == name), key=lambda r: r[1])
        d[f"{name}_model"] = np.array([r[2] for r in rs_]); d[f"{name}_istats"] = np.array([r[3] for r in rs_], dtype=np.int64)
        d[f"{name}_fstats"] = np.array([r[4] for r in rs_]); d[f"{name}_mask"] = np.array([r[5] for r in rs_]); d[f"{name}_digest"] = np.array([r[6] for r in rs_], dtype=np.uint64)
        it = d[f"{name}_istats"][:, 1]
        print(name, "iterations min / median / max", it.min(), int(np.median(it)), it.max(), "refinements mean", d[f"{name}_istats"][:, 0].mean(), flush=True)
    out = os.path.join(HERE, "..", "golden", "dynamic_ref.npz")
    np.savez_compressed(out, **d)
    print(os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
