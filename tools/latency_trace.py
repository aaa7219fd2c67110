#!/usr/bin/env python3
"""A few single-pair calls of the drop-in entry point (headline shape: N = 2000, 10^4 iterations) — the program to put behind
`rocprofv3 --kernel-trace` to see where a B = 1 call spends its time (tools/rocpd_timeline.py reads the database).
usage: latency_trace.py [n = 2000] [iterations = 10000] [calls = 6]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import mdrp_amd.poselib as poselib
from mdrp_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
it = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cam = {"model": "SIMPLE_PINHOLE", "width": 1600, "height": 1200, "params": [800.0, 0.0, 0.0]}
ro = {"max_iterations": it, "min_iterations": it, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}
b = synth.make_batch(0, 8, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
ts = []
for k in range(calls + 2):
    i = k % 8
    t0 = time.perf_counter()
    poselib.estimate_monodepth_relative_pose(b["x1"][i], b["x2"][i], b["d1"][i], b["d2"][i], cam, cam, ro, {"loss_type": "TRUNCATED_CAUCHY"})
    ts.append(time.perf_counter() - t0)
print(f"B = 1, N = {n}, {it} iterations: {[round(1e3 * t, 2) for t in ts[2:]]} ms, median {1e3 * np.median(ts[2:]):.2f}")
