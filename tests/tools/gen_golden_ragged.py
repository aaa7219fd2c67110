#!/usr/bin/env python3
"""Reference-binary LO counts / inlier counts for the 4 x 136 ragged pairs of tests/test_gpu_wide.py::test_large_ragged_batch_vs_oracle
(seeds + outputs only; inputs regenerate from mdrp_amd.synth).  The GPU test compares everything with the CPU oracle and
lets `refinements` equal either the oracle's or the reference's: the two differ on a few pairs through the solver-level
deviation classes of DESIGN.md §5, and the HIP solvers side with one or the other.

Runs only in the build container:   python3 tests/tools/gen_golden_ragged.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import refshim as rs  # noqa: E402
from mdrp_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "..", "golden", "ragged_lo.npz")
CASES = [(0, False, None), (0, True, None), (1, False, "shared"), (2, False, "varying")]


def ragged_sizes(kind, es, B=136):
    rng = np.random.default_rng(77 + kind + int(es))
    ns = rng.integers(40, 301, B)
    ns[:4] = [300, 40, 64, 65]
    return ns


def ragged_pair(kind, es, rf, i, n):
    return synth.make_pair(12000 + 50 * kind + i, int(n), noise_px=0.6, depth_noise=0.02, outlier_frac=[0.0, 0.25, 0.5][i % 3],
                           random_focal=rf, shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0)


def main():
    d = {}
    cam = rs.cam_flat(0, 1600, 1200, [800.0, 0.0, 0.0])
    for kind, es, rf in CASES:
        ns = ragged_sizes(kind, es)
        out = []
        for i, n in enumerate(ns):
            p = ragged_pair(kind, es, rf, i, n)
            kw = dict(max_iterations=700, min_iterations=700, max_epipolar_error=2.0, max_reproj_error=16.0, estimate_shift=es)
            m, st, mask = rs.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], rs.ropt(**kw), rs.bopt(loss_type=4),
                                      cam if kind == 0 else None, cam if kind == 0 else None)
            out.append([int(st[0]), int(st[2])])
        d[f"k{kind}_s{int(es)}"] = np.array(out, dtype=np.int32)
        print(kind, es, "done", flush=True)
    np.savez_compressed(OUT, **d)


if __name__ == "__main__":
    main()
