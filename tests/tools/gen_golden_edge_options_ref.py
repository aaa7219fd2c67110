#!/usr/bin/env python3
"""REFERENCE-BINARY fixture for options at the EDGES of their ranges (tests/golden/edge_options_ref.npz): one N = 300 pair per estimator, one option moved to
an edge per case (tests/helpers.py edge_cases): max_iterations 0 / 1 / below min_iterations, success_prob 0 and 1, dyn_num_trials_mult 0, thresholds 0 /
1e-3 / 100 px, max_reproj_error 0 (reprojection terms off), weight_sampson 0 and negative, a 41-bit seed, loss_scale 0, lambda pinned, tolerances of 1.
Outputs only (model, stats, mask); inputs regenerate from mdrp_amd.synth.

Build container only:   bash oracle/build_ref.sh && python3 tests/tools/gen_golden_edge_options_ref.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import gen_golden_headline_ref as gh  # noqa: E402
import refshim as rs  # noqa: E402
from helpers import OPTIONS_KINDS, OPTIONS_NAMES, edge_cases, edge_pair, input_digest  # noqa: E402


def main():
    d = {"names": np.array(OPTIONS_NAMES)}
    cam = rs.cam_flat(0, 1600, 1200, [800.0, 0.0, 0.0])
    for name in OPTIONS_NAMES:
        kind, es, rf = OPTIONS_KINDS[name]
        p = edge_pair(name)
        models, stats, masks = [], [], []
        for rod, bod in edge_cases():
            gh._srand(1)
            m, st, mask = rs.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], rs.ropt(estimate_shift=es, **rod), rs.bopt(**bod), cam if kind == 0 else None, cam if kind == 0 else None)
            models.append(np.r_[m, 1.0, 1.0] if kind == 0 else np.asarray(m)); stats.append(st); masks.append(np.packbits(mask))
        d[f"{name}_model"] = np.array(models); d[f"{name}_stats"] = np.array(stats); d[f"{name}_mask"] = np.array(masks); d[f"{name}_digest"] = np.array(input_digest(p), dtype=np.uint64)
        print(name, "iterations", sorted(set(int(s[1]) for s in stats)), "NaN models", int(np.isnan(np.array(models)).any(axis=1).sum()), flush=True)
    out = os.path.join(HERE, "..", "golden", "edge_options_ref.npz")
    np.savez_compressed(out, **d)
    print(os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
