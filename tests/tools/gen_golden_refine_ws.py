#!/usr/bin/env python3
"""REFERENCE-BINARY fixture for the Sampson weight inside the three monodepth refiners (tests/golden/refine_ws.npz).

`RansacOptions::monodepth_weight_sampson` is 1.0 in every experiment of the reference's eval.py, and at 1.0 every reading of "weight" coincides; away from
1.0 the binary (a) multiplies the Sampson term of the COST by ws, (b) multiplies the Sampson row of the NORMAL EQUATIONS by ws^2, and (c) evaluates that
row's IRLS loss weight at r^2 in refine_monodepth_relpose but at ws r^2 in the shared- and varying-focal refiners.  This fixture pins all three: the nine
problems of refine.npz (inputs are read from there, not stored again) x ws in {0.3, 0.5, 0.7, 1.3, 2, 3} x six losses x {1, 25} LM iterations (the
non-robust TRIVIAL and HUBER losses at {1, 3}: with 25 % outliers their long trajectories amplify rounding) x {no, seeded} per-correspondence weights.

Build container only:   bash oracle/build_ref.sh && python3 tests/tools/gen_golden_refine_ws.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
import refshim as rs  # noqa: E402
from helpers import refine_ws_weights as point_weights  # noqa: E402

WS = (0.3, 0.5, 0.7, 1.3, 2.0, 3.0)


def main():
    g = np.load(os.path.join(HERE, "..", "golden", "refine.npz"))
    cases, out = [], []
    for i in range(9):
        kind = i % 3
        es = (i // 3) % 2 if kind == 0 else 0
        sc = 800.0 if kind == 0 else 700.0
        x1, x2, d1, d2, m = g[f"x1_{i}"], g[f"x2_{i}"], g[f"d1_{i}"], g[f"d2_{i}"], g[f"model_{i}"]
        for ws in WS:
            for lt in range(6):
                for its in ((1, 3) if lt in (0, 2) else (1, 25)):
                    for weighted in (0, 1):
                        bo = rs.bopt(max_iterations=its, loss_type=lt, loss_scale=2.0 / sc, gradient_tol=1e-10)
                        w = point_weights(i, len(x1)) if weighted else None
                        if kind == 0:
                            r, st = rs.refine_calib(x1, x2, d1, d2, m[:10], 1 / 64.0, ws, bo, es, w)
                            r = np.r_[r, 1.0, 1.0]
                        else:
                            r, st = rs.refine_focal(kind == 2, x1, x2, d1, d2, m, 1 / 64.0, ws, bo, w)
                        cases.append([i, kind, es, ws, lt, its, weighted, 2.0 / sc])
                        out.append(np.r_[r, st])
    path = os.path.join(HERE, "..", "golden", "refine_ws.npz")
    np.savez_compressed(path, cases=np.array(cases), out=np.array(out))
    print(len(cases), "cases,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
