#!/usr/bin/env python3
"""GPU minimal solver vs the CPU oracle on the exact sample sequence of one estimator run (run on the GPU box):
finds the iterations whose solution sets differ — the solver-level cause of an LO-count difference between the HIP path
and the oracle.   usage: diag_gpu_solver.py KIND ES PAIR_INDEX_IN_RAGGED_TEST"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from mdrp_amd import _capi, synth  # noqa: E402
from oracle import pyorc as po  # noqa: E402
from helpers import match_solution_sets  # noqa: E402

kind, es, i = int(sys.argv[1]), bool(int(sys.argv[2])), int(sys.argv[3])
rf = [None, "shared", "varying"][kind]
rng = np.random.default_rng(77 + kind + int(es))
ns = rng.integers(40, 301, 136)
ns[:4] = [300, 40, 64, 65]
n = int(ns[i])
p = synth.make_pair(12000 + 50 * kind + i, n, noise_px=0.6, depth_noise=0.02, outlier_frac=[0.0, 0.25, 0.5][i % 3],
                    random_focal=rf, shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0)
assert kind == 0, "calibrated only: the focal estimators normalise by a data-dependent scale"
f = 800.0
x1, x2 = p["x1"] / f, p["x2"] / f
samples = po.draw_samples(0, n, 700)
x1h = np.concatenate([x1[samples], np.ones((700, 3, 1))], axis=2)
x2h = np.concatenate([x2[samples], np.ones((700, 3, 1))], axis=2)
d1, d2 = p["d1"][samples], p["d2"][samples]
h = _capi.default_handle(0)
solver = 1 if es else 0
out, cnt = h.solver_batch(solver, x1h, x2h, d1, d2)
ofn = po.solver_calib_shift if es else po.solver_calib_p3p
bad = 0
for it in range(700):
    mine = [_capi.model_to_array(m) for m in out[it, : cnt[it]]]
    ref = list(ofn(x1h[it], x2h[it], d1[it], d2[it]))
    if not match_solution_sets(ref, mine, 1e-6):
        bad += 1
        print("iteration", it, "sample", samples[it], "gpu", len(mine), "oracle", len(ref))
        for m in mine:
            print("   gpu   ", np.array2string(m[:8], precision=6))
        for m in ref:
            print("   oracle", np.array2string(np.asarray(m)[:8], precision=6))
print("differing iterations:", bad)
