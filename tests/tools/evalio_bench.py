#!/usr/bin/env python3
"""End-to-end rate of the dataset loop (mdrp_amd.evalio) on a synthetic H5-like dict: host stacking + H2D + GPU + records."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mdrp_amd import evalio, synth

P, N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 1500
h5 = {}
b = synth.make_batch(12000, P, N, noise_px=0.5, depth_noise=0.02, outlier_frac=0.4, pp=(640.0, 480.0))
for i in range(P):
    a, c = f"i{i:05d}a_o", f"i{i:05d}b"
    data = np.zeros((N, 32)); data[:, :2] = b["x1"][i]; data[:, 2:4] = b["x2"][i]; data[:, 26] = b["d1"][i]; data[:, 27] = b["d2"][i]
    h5[f"corr_{a}_{c}"] = data; h5[f"pose_{a}_{c}"] = np.c_[b["gt"][i]["R"], b["gt"][i]["t"]]
    K = np.array([[800.0, 0, 640.0], [0, 800.0, 480.0], [0, 0, 1]]); h5[f"K_{a}"] = K; h5[f"K_{c}"] = K
exps = ["3p_ours_scale_hybrid_ctruncated+10"]
evalio.evaluate_calibrated(h5, exps, iters=1000, threshold=2.0, first=64)   # warm-up
t0 = time.perf_counter()
res = evalio.evaluate_calibrated(h5, exps, iters=1000, threshold=2.0)
dt = time.perf_counter() - t0
print(f"{P} pairs x {N} correspondences, 1000 iterations: {dt:.2f} s end to end = {P / dt:.0f} pairs/s")
print(evalio.format_table(evalio.summarize(exps, res)))
