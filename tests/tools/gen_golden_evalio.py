#!/usr/bin/env python3
"""tests/golden/evalio.npz from the reference's own Python helpers (utils/data.py), imported in the build container.
Fixtures are data only: inputs and the reference functions' outputs.      python3 tests/tools/gen_golden_evalio.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from utils.data import R_err_fun, t_err_fun, get_valid_depth_mask, depth_indices  # noqa: E402

sys.path.insert(0, os.path.join(HERE, "..", ".."))
from mdrp_amd import synth  # noqa: E402

rng = np.random.default_rng(77)
n = 64
R_gt = np.array([synth.rodrigues(rng.normal(0, 0.8, 3)) for _ in range(n)])
R = np.array([R_gt[i] @ synth.rodrigues(rng.normal(0, 10.0 ** rng.uniform(-6, 0.3), 3)) for i in range(n)])
R[5] = R_gt[5]                                   # exact match
R[6] = R_gt[6] @ synth.rodrigues(np.array([np.pi, 0, 0]))  # 180 degrees
t_gt = rng.normal(size=(n, 3))
t = t_gt * rng.uniform(0.2, 3, (n, 1)) + rng.normal(size=(n, 3)) * (10.0 ** rng.uniform(-8, 0.5, (n, 1)))
t[3] = -2.0 * t_gt[3]                            # opposite direction: sign-agnostic error = 0
t[4] = 0.0                                       # degenerate
R_err = np.array([R_err_fun({"R_gt": R_gt[i], "R": R[i]}) for i in range(n)])
t_err = np.array([t_err_fun({"t_gt": t_gt[i], "t": t[i]}) for i in range(n)])
d = rng.uniform(0.1, 9, (200, 2))
d[rng.integers(0, 200, 20), rng.integers(0, 2, 20)] = np.inf
d[rng.integers(0, 200, 20), rng.integers(0, 2, 20)] = np.nan
d[rng.integers(0, 200, 20), rng.integers(0, 2, 20)] = -1.0
d[7] = [0.0, 0.0]
mask = get_valid_depth_mask(d.copy())
cols = np.array([depth_indices(k) for k in range(1, 13)])
np.savez_compressed(os.path.join(HERE, "..", "golden", "evalio.npz"), R_gt=R_gt, R=R, t_gt=t_gt, t=t, R_err=R_err, t_err=t_err, d=d,
                    invalid_mask=mask, depth_columns=cols)
print("evalio.npz", n, "pose pairs,", int(mask.sum()), "invalid depth rows")
