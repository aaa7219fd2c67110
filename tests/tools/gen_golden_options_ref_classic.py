#!/usr/bin/env python3
"""REFERENCE-BINARY fixture for the randomised OPTIONS campaign of the comparison rows (tests/golden/options_ref_classic.npz): 5-point relative pose,
6-point shared focal and 7-point fundamental matrix x 64 cases, every case with its own size, outlier share, noise, threshold, seed, fixed or dynamic
iteration budget, loss type, loss scale, bundle iteration cap and (5-point) cameras / (6-point) principal point.  Outputs + the case table only; the
inputs regenerate from mdrp_amd.synth (tests/helpers.py classic_options_pair).

Build container only:   bash oracle/build_ref.sh && python3 tests/tools/gen_golden_options_ref_classic.py"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import gen_golden_headline_ref as gh  # noqa: E402
from helpers import CLASSIC_OPTIONS_COLS as COLS, CLASSIC_OPTIONS_KINDS as KINDS, classic_options_cameras, classic_options_pair, input_digest  # noqa: E402

CASES = 64


def case_table():
    rng = np.random.default_rng(20261004)
    t = np.zeros((CASES, len(COLS)))
    for j in range(CASES):
        budget = [(300, 300), (1500, 1500), (2000, 100), (100000, 1000)][int(rng.integers(0, 4))]
        t[j] = (int(rng.choice([40, 60, 150, 400, 900, 1500])), float(rng.choice([0.0, 0.2, 0.4, 0.6])), float(rng.choice([0.25, 0.5, 1.0])),
                float(rng.choice([0.5, 1.0, 2.0, 4.0])), int(rng.integers(0, 1000)), budget[0], budget[1], j % 6, float(rng.choice([0.5, 1.0, 3.0])),
                int(rng.choice([0, 5, 100, 100])), float(rng.choice([500.0, 800.0, 1400.0])), float(rng.choice([500.0, 800.0, 1400.0])),
                float(rng.choice([0.0, 640.0, 3.0])), float(rng.choice([0.0, 480.0, -2.0])), int(rng.integers(0, 2)))
    return t


def _work(args):
    name, j, row = args
    import refshim as rs
    kind = KINDS[name]
    p = classic_options_pair(name, j, row)
    ro = rs.ropt(max_iterations=int(row[5]), min_iterations=int(row[6]), max_epipolar_error=float(row[3]), seed=int(row[4]))
    bo = rs.bopt(max_iterations=int(row[9]), loss_type=int(row[7]), loss_scale=float(row[8]), gradient_tol=1e-10)
    c1, c2 = classic_options_cameras(row)
    cam1, cam2 = (rs.cam_flat(c1[0], 1600, 1200, c1[1]), rs.cam_flat(c2[0], 1600, 1200, c2[1])) if kind == 3 else (None, None)
    gh._srand(1)
    m, st, mask = rs.estimate_classic(kind, p["x1"], p["x2"], ro, bo, cam1, cam2, pp=(float(row[12]), float(row[13])))
    full = np.zeros(12); m = np.asarray(m, float).reshape(-1); full[: len(m)] = m
    mk = np.zeros(1500, dtype=np.uint8); mk[: len(mask)] = mask
    return name, j, full, (int(st[0]), int(st[1]), int(st[2])), (float(st[3]), float(st[4])), np.packbits(mk), input_digest(p)


def main():
    t = case_table()
    jobs = [(name, j, t[j]) for name in KINDS for j in range(CASES)]
    with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
        rows = pool.map(_work, jobs, chunksize=2)
    d = {"names": np.array(list(KINDS)), "columns": np.array(COLS), "cases": t}
    for name in KINDS:
        rs_ = sorted((r for r in rows if r[0] == name), key=lambda r: r[1])
        d[f"{name}_model"] = np.array([r[2] for r in rs_]); d[f"{name}_istats"] = np.array([r[3] for r in rs_], dtype=np.int64)
        d[f"{name}_fstats"] = np.array([r[4] for r in rs_]); d[f"{name}_mask"] = np.array([r[5] for r in rs_]); d[f"{name}_digest"] = np.array([r[6] for r in rs_], dtype=np.uint64)
        print(name, "iterations", d[f"{name}_istats"][:, 1].min(), d[f"{name}_istats"][:, 1].max(), "NaN models", int(np.isnan(d[f"{name}_model"]).any(axis=1).sum()),
              "no inliers", int((d[f"{name}_istats"][:, 2] == 0).sum()), flush=True)
    out = os.path.join(HERE, "..", "golden", "options_ref_classic.npz")
    np.savez_compressed(out, **d)
    print(os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
