#!/usr/bin/env python3
"""REFERENCE-BINARY fixture for a randomised OPTIONS campaign (tests/golden/options_ref.npz): 4 estimators x 96 cases, every case with its own
problem size (N from 40 to 1500), outlier fraction, noise level, thresholds, Sampson weight, seed, iteration budget (fixed or dynamic) and BundleOptions
(all six loss types, loss scale, iteration cap incl. 0, tolerances, damping and its bounds), its stopping rule (success_prob, dyn_num_trials_mult) and, for the
calibrated estimators, its two cameras (focal lengths, principal point, SIMPLE_PINHOLE / PINHOLE) — what the drop-in boundary hands through, varied together.  The case table is stored with the
outputs, the inputs regenerate from mdrp_amd.synth.

Runs only in the build container:   bash oracle/build_ref.sh && python3 tests/tools/gen_golden_options_ref.py      (8 workers, < 1 minute)"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import gen_golden_headline_ref as gh  # noqa: E402

sys.path.insert(0, os.path.join(HERE, ".."))
from helpers import OPTIONS_COLS as COLS, OPTIONS_FIRST as FIRST, OPTIONS_KINDS as KINDS, OPTIONS_NAMES as NAMES, options_cameras, options_dicts, options_pair as make_pair  # noqa: E402

CASES = 96


def case_table():
    rng = np.random.default_rng(20261003)
    t = np.zeros((CASES, len(COLS)))
    for j in range(CASES):
        n = int(rng.choice([40, 60, 150, 400, 900, 1500]))
        budget = [(300, 300), (1500, 1500), (2000, 100), (100000, 1000)][int(rng.integers(0, 4))]
        lam = [(1e-10, 1e10), (1e-6, 1e3)][int(rng.integers(0, 2))]
        t[j] = (n, float(rng.choice([0.0, 0.2, 0.4, 0.6])), float(rng.choice([0.25, 0.5, 1.0])), float(rng.choice([0.5, 1.0, 2.0, 4.0])),
                float(rng.choice([4.0, 12.0, 16.0, 32.0])), float(rng.choice([1.0, 0.5, 2.0, 0.7, 1.3])), int(rng.integers(0, 1000)), budget[0], budget[1],
                j % 6, float(rng.choice([0.5, 1.0, 3.0])), int(rng.choice([0, 5, 100, 100])),
                float(rng.choice([0.9999, 0.99, 0.9])), float(rng.choice([3.0, 1.0, 5.0])), float(rng.choice([1e-10, 1e-8, 1e-6])), float(rng.choice([1e-8, 1e-6])),
                float(rng.choice([1e-3, 1e-2, 1.0])), lam[0], lam[1], float(rng.choice([500.0, 800.0, 1400.0])), float(rng.choice([500.0, 800.0, 1400.0])),
                float(rng.choice([0.0, 640.0, 3.0])), float(rng.choice([0.0, 480.0, -2.0])), int(rng.integers(0, 2)))
    return t


def _work(args):
    name, j, row = args
    import refshim as rs
    kind, es, rf = KINDS[name]
    c1, c2 = options_cameras(row)
    cam1, cam2 = (rs.cam_flat(c1[0], 1600, 1200, c1[1]), rs.cam_flat(c2[0], 1600, 1200, c2[1])) if kind == 0 else (None, None)
    p = make_pair(name, j, row)
    rod, bod = options_dicts(row, es)
    ro, bo = rs.ropt(**rod), rs.bopt(**bod)
    gh._srand(1)
    m, st, mask = rs.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], ro, bo, cam1, cam2)
    m12 = np.r_[m, 1.0, 1.0] if kind == 0 else np.asarray(m)
    mk = np.zeros(1500, dtype=np.uint8); mk[:len(mask)] = mask
    return name, j, m12, (int(st[0]), int(st[1]), int(st[2])), (float(st[3]), float(st[4])), np.packbits(mk), gh.input_digest(p)


def main():
    t = case_table()
    jobs = [(name, j, t[j]) for name in NAMES for j in range(CASES)]
    with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
        rows = pool.map(_work, jobs, chunksize=2)
    d = {"names": np.array(NAMES), "columns": np.array(COLS), "cases": t, "first_index": np.array(FIRST)}
    for name in NAMES:
        rs_ = sorted((r for r in rows if r[0] == name), key=lambda r: r[1])
        d[f"{name}_model"] = np.array([r[2] for r in rs_]); d[f"{name}_istats"] = np.array([r[3] for r in rs_], dtype=np.int64)
        d[f"{name}_fstats"] = np.array([r[4] for r in rs_]); d[f"{name}_mask"] = np.array([r[5] for r in rs_]); d[f"{name}_digest"] = np.array([r[6] for r in rs_], dtype=np.uint64)
        print(name, "iterations", d[f"{name}_istats"][:, 1].min(), d[f"{name}_istats"][:, 1].max(), "NaN models", int(np.isnan(d[f"{name}_model"]).any(axis=1).sum()),
              "no inliers", int((d[f"{name}_istats"][:, 2] == 0).sum()), flush=True)
    out = os.path.join(HERE, "..", "golden", "options_ref.npz")
    np.savez_compressed(out, **d)
    print(os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
