#!/usr/bin/env python3
"""REFERENCE-BINARY fixture for options at the edges of their ranges, comparison rows (tests/golden/edge_options_ref_classic.npz): one N = 300 pair per
estimator (5-point, 6-point shared focal, 7-point), one option at an edge per case (tests/helpers.py classic_edge_cases).  Outputs only.

Build container only:   bash oracle/build_ref.sh && python3 tests/tools/gen_golden_edge_options_ref_classic.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import gen_golden_headline_ref as gh  # noqa: E402
import refshim as rs  # noqa: E402
from helpers import CLASSIC_OPTIONS_KINDS, classic_edge_cases, classic_edge_pair, input_digest  # noqa: E402


def main():
    d = {"names": np.array(list(CLASSIC_OPTIONS_KINDS))}
    cam = rs.cam_flat(0, 1600, 1200, [800.0, 0.0, 0.0])
    for name, kind in CLASSIC_OPTIONS_KINDS.items():
        p = classic_edge_pair(name)
        models, stats, masks = [], [], []
        for rod, bod in classic_edge_cases():
            gh._srand(1)
            m, st, mask = rs.estimate_classic(kind, p["x1"], p["x2"], rs.ropt(**rod), rs.bopt(**bod), cam if kind == 3 else None, cam if kind == 3 else None, pp=(0.0, 0.0))
            full = np.zeros(12); m = np.asarray(m, float).reshape(-1); full[: len(m)] = m
            models.append(full); stats.append(st); masks.append(np.packbits(mask))
        d[f"{name}_model"] = np.array(models); d[f"{name}_stats"] = np.array(stats); d[f"{name}_mask"] = np.array(masks); d[f"{name}_digest"] = np.array(input_digest(p), dtype=np.uint64)
        print(name, "iterations", sorted(set(int(s[1]) for s in stats)), flush=True)
    out = os.path.join(HERE, "..", "golden", "edge_options_ref_classic.npz")
    np.savez_compressed(out, **d)
    print(os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
