#!/usr/bin/env python3
"""REFERENCE-BINARY fixture for degenerate geometry (tests/golden/degenerate_ref.npz): per estimator 4 scene types (pure rotation, planar scene, a baseline
of 1e-4, motion along the optical axis) x 6 seeds, N = 400, 25 % outliers, 1000 iterations, the reference's own options.  Outputs only; the inputs
regenerate from tests/helpers.py degenerate_pair.

Build container only:   bash oracle/build_ref.sh && python3 tests/tools/gen_golden_degenerate_ref.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import gen_golden_headline_ref as gh  # noqa: E402
import refshim as rs  # noqa: E402
from helpers import DEGENERATE_MODES, DEGENERATE_SEEDS, OPTIONS_KINDS, OPTIONS_NAMES, degenerate_pair, input_digest  # noqa: E402


def main():
    d = {"names": np.array(OPTIONS_NAMES), "modes": np.array(DEGENERATE_MODES)}
    cam = rs.cam_flat(0, 1600, 1200, [800.0, 0.0, 0.0])
    for name in OPTIONS_NAMES:
        kind, es, rf = OPTIONS_KINDS[name]
        models, stats, masks, digs = [], [], [], []
        for mode in DEGENERATE_MODES:
            for seed in range(DEGENERATE_SEEDS):
                p = degenerate_pair(name, mode, seed)
                ro = rs.ropt(max_iterations=1000, min_iterations=1000, max_epipolar_error=2.0, max_reproj_error=16.0, seed=seed, estimate_shift=es)
                gh._srand(1)
                m, st, mask = rs.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], ro, rs.bopt(max_iterations=100, loss_type=4, loss_scale=1.0, gradient_tol=1e-10),
                                          cam if kind == 0 else None, cam if kind == 0 else None)
                models.append(np.r_[m, 1.0, 1.0] if kind == 0 else np.asarray(m)); stats.append(st); masks.append(np.packbits(mask)); digs.append(input_digest(p))
        d[f"{name}_model"] = np.array(models); d[f"{name}_stats"] = np.array(stats); d[f"{name}_mask"] = np.array(masks); d[f"{name}_digest"] = np.array(digs, dtype=np.uint64)
        print(name, "inliers", [int(s[2]) for s in stats], flush=True)
    out = os.path.join(HERE, "..", "golden", "degenerate_ref.npz")
    np.savez_compressed(out, **d)
    print(os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
