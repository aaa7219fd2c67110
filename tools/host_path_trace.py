#!/usr/bin/env python3
"""A few steps of the headline batch through the HOST-buffer entry point (mdrp_estimate_batch, MDRP_MEM_HOST) — the program to put behind
`rocprofv3 --kernel-trace --memory-copy-trace` to see whether the H2D slices overlap the kernels (tools/host_path_timeline.py reads the database).
usage: host_path_trace.py [pairs = 1024] [steps = 4]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from mdrp_amd import _capi, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
b = synth.make_batch(0, B, 2000, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
cams = np.zeros(B, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
ro = _capi.ransac_opt_from_dict({"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
h = _capi.Handle(0)
xs = [b["x1"], b["x2"], b["d1"], b["d2"]]
h.estimate_batch(0, *xs, ro, bo, None, cams, cams)
ts = []
for _ in range(steps):
    t0 = time.perf_counter()
    h.estimate_batch(0, *xs, ro, bo, None, cams, cams)
    ts.append(time.perf_counter() - t0)
print(f"host-buffer steps of {B} pairs: {[round(1e3 * t, 2) for t in ts]} ms -> {B / np.median(ts):.0f} pairs/s")
h.close()
