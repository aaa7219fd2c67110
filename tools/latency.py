#!/usr/bin/env python3
"""Single-pair latency of the drop-in API (what make_pair.py / the demo notebook experience), on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import mdrp_amd.poselib as poselib
from mdrp_amd import synth

cam = {"model": "SIMPLE_PINHOLE", "width": 1600, "height": 1200, "params": [800.0, 0.0, 0.0]}
for n, ro in ((150, {"max_epipolar_error": 2.0, "max_reproj_error": 16.0}),
              (2000, {"max_epipolar_error": 2.0, "max_reproj_error": 16.0}),
              (2000, {"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})):
    p = synth.make_pair(1, n, outlier_frac=0.3)
    poselib.estimate_monodepth_relative_pose(p["x1"], p["x2"], p["d1"], p["d2"], cam, cam, ro, {"loss_type": "TRUNCATED_CAUCHY"})
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        g, info = poselib.estimate_monodepth_relative_pose(p["x1"], p["x2"], p["d1"], p["d2"], cam, cam, ro, {"loss_type": "TRUNCATED_CAUCHY"})
        ts.append(time.perf_counter() - t0)
    print(f"N={n} opts={ro}: median {1e3 * np.median(ts):.2f} ms  (iterations {info['iterations']}, inliers {info['num_inliers']})")

# the non-monodepth baselines through their drop-in signatures (single pair)
for name, fn in (("estimate_relative_pose (5-point)", lambda p, ro: poselib.estimate_relative_pose(p["x1"], p["x2"], cam, cam, ro, {"loss_type": "TRUNCATED_CAUCHY"})),
                 ("estimate_fundamental (7-point)", lambda p, ro: poselib.estimate_fundamental(p["x1"], p["x2"], ro, {"loss_type": "TRUNCATED_CAUCHY"}))):
    for n, ro in ((150, {"max_epipolar_error": 2.0}), (2000, {"max_epipolar_error": 2.0}),
                  (2000, {"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0})):
        p = synth.make_pair(1, n, outlier_frac=0.3)
        fn(p, ro)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            m, info = fn(p, ro)
            ts.append(time.perf_counter() - t0)
        print(f"{name} N={n} opts={ro}: median {1e3 * np.median(ts):.2f} ms  (iterations {info['iterations']}, inliers {info['num_inliers']})")
