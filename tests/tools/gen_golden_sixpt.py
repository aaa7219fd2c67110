#!/usr/bin/env python3
"""tests/golden/sixpt.npz — the 6-point shared-focal baseline of the REFERENCE binary (relpose_6pt_shared_focal,
ransac_shared_focal_relpose / estimate_shared_focal_relative_pose; /root/reference/eval_shared_f.py:161):
  * solver: 96 problems (geometric with and without noise, and random), inputs + the binary's solutions (q, t, f);
  * estimator: 24 small runs (every loss type, fixed and dynamic stopping) with inputs, and 8 full-size runs
    (N = 2000, 10^4 iterations, 50 % outliers) as seeds + outputs.
Runs only in the build container:   python3 tests/tools/gen_golden_sixpt.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import refshim as rs  # noqa: E402
from mdrp_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "..", "golden", "sixpt.npz")


def unit_rows(x):
    h = np.c_[x, np.ones(len(x))]
    return np.ascontiguousarray(h / np.linalg.norm(h, axis=1, keepdims=True))


def sixpt_problem(i, rng):
    """six correspondences in scale-normalised coordinates (what SharedFocalRelativePoseEstimator hands its solver)"""
    if i % 3 != 2:
        pr = synth.make_pair(9100 + i, 6, noise_px=0.0 if i % 3 == 0 else 0.5, random_focal="shared")
        sc = (np.sqrt((pr["x1"] ** 2).sum(1)).sum() + np.sqrt((pr["x2"] ** 2).sum(1)).sum()) / (np.sqrt(2.0) * 6)
        return unit_rows(pr["x1"] / sc), unit_rows(pr["x2"] / sc)
    return unit_rows(rng.uniform(-1, 1, (6, 2))), unit_rows(rng.uniform(-1, 1, (6, 2)))


SMALL = [  # index -> n, outliers, iterations, min_iterations, loss, threshold, seed
    (n, outl, its, (its if k % 4 < 2 else 20), [4, 1, 3, 0, 5, 2][k % 6], [2.0, 1.0][k % 2], k % 5)
    for k, (n, outl, its) in enumerate([(100, 0.1, 300), (200, 0.3, 500), (400, 0.5, 1000), (1000, 0.0, 300)] * 6)
]
FULL_INDICES = [0, 33, 66, 99, 132, 165, 198, 231]


def small_pair(k):
    n, outl = SMALL[k][0], SMALL[k][1]
    return synth.make_pair(9300 + k, n, noise_px=0.5, outlier_frac=outl, random_focal="shared", pp=(0.0, 0.0))


def full_pair(index):
    return synth.make_pair(index, 2000, noise_px=0.5, outlier_frac=0.5, random_focal="shared", pp=(0.0, 0.0))


def main():
    d = {}
    rng = np.random.default_rng(66)
    xs1, xs2, sols, cnts = [], [], [], []
    for i in range(96):
        a, b = sixpt_problem(i, rng)
        out = rs.relpose_6pt(a, b)
        full = np.full((15, 8), np.nan)
        full[: len(out)] = out[:15]
        xs1.append(a); xs2.append(b); sols.append(full); cnts.append(len(out))
    d["solver_x1"] = np.array(xs1); d["solver_x2"] = np.array(xs2); d["solver_sols"] = np.array(sols); d["solver_n"] = np.array(cnts)
    print("6-point solver, solutions per problem:", np.bincount(cnts), flush=True)
    cases = []
    for k, (n, outl, its, min_its, loss, thr, seed) in enumerate(SMALL):
        pr = small_pair(k)
        pp = (3.0, -2.0) if k % 3 == 1 else (0.0, 0.0)
        x1, x2 = pr["x1"] + pp, pr["x2"] + pp
        ro = rs.ropt(max_iterations=its, min_iterations=min_its, max_epipolar_error=thr, seed=seed)
        bo = rs.bopt(loss_type=loss, loss_scale=thr)
        m, st, mask = rs.estimate_classic(4, x1, x2, ro, bo, pp=pp)
        d.update({f"est_x1_{k}": x1, f"est_x2_{k}": x2, f"est_model_{k}": m, f"est_stats_{k}": st, f"est_mask_{k}": mask})
        cases.append([k, n, its, min_its, loss, thr, seed, pp[0], pp[1]])
        print("6-point estimate", k, n, its, st, "f", m[7], "gt", pr["f1"], flush=True)
    d["est_cases"] = np.array(cases)
    fm, fs, fk = [], [], []
    for index in FULL_INDICES:
        pr = full_pair(index)
        ro = rs.ropt(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, seed=0)
        m, st, mask = rs.estimate_classic(4, pr["x1"], pr["x2"], ro, rs.bopt(loss_type=4), pp=(0.0, 0.0))
        fm.append(m); fs.append(st); fk.append(np.packbits(mask))
        print("6-point full size", index, st, "f", m[7], "gt", pr["f1"], flush=True)
    d["full_indices"] = np.array(FULL_INDICES); d["full_model"] = np.array(fm); d["full_stats"] = np.array(fs); d["full_mask"] = np.array(fk)
    np.savez_compressed(OUT, **d)
    print(os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
