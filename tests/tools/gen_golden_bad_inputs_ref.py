#!/usr/bin/env python3
"""REFERENCE-BINARY fixture for corrupted inputs (tests/golden/bad_inputs_ref.npz): per estimator one N = 300 pair with a tenth of its correspondences
corrupted in seven ways (tests/helpers.py bad_input_pair: depths 0 / negative / NaN in either image, NaN / inf coordinates, identical correspondences),
500 iterations, the reference's own options.  What the binary does with them: zero and negative depths drop the reprojection terms of those
correspondences, NaN depths too (the terms are skipped, the cost stays finite), NaN / inf coordinates never become inliers.  Outputs only.

Build container only:   bash oracle/build_ref.sh && python3 tests/tools/gen_golden_bad_inputs_ref.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import gen_golden_headline_ref as gh  # noqa: E402
import refshim as rs  # noqa: E402
from helpers import BAD_INPUT_MODES, OPTIONS_KINDS, OPTIONS_NAMES, bad_input_pair  # noqa: E402

RO = dict(max_iterations=500, min_iterations=500, max_epipolar_error=2.0, max_reproj_error=16.0, seed=2)
BO = dict(max_iterations=100, loss_type=4, loss_scale=1.0, gradient_tol=1e-10)


def main():
    d = {"names": np.array(OPTIONS_NAMES), "modes": np.array(BAD_INPUT_MODES)}
    cam = rs.cam_flat(0, 1600, 1200, [800.0, 0.0, 0.0])
    for name in OPTIONS_NAMES:
        kind, es, rf = OPTIONS_KINDS[name]
        models, stats, masks = [], [], []
        for mode in BAD_INPUT_MODES:
            p = bad_input_pair(name, mode)
            gh._srand(1)
            m, st, mask = rs.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], rs.ropt(estimate_shift=es, **RO), rs.bopt(**BO), cam if kind == 0 else None, cam if kind == 0 else None)
            models.append(np.r_[m, 1.0, 1.0] if kind == 0 else np.asarray(m)); stats.append(st); masks.append(np.packbits(mask))
        d[f"{name}_model"] = np.array(models); d[f"{name}_stats"] = np.array(stats); d[f"{name}_mask"] = np.array(masks)
        print(name, [tuple(int(v) for v in s[:3]) for s in stats], flush=True)
    out = os.path.join(HERE, "..", "golden", "bad_inputs_ref.npz")
    np.savez_compressed(out, **d)
    print(os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
