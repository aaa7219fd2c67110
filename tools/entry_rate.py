#!/usr/bin/env python3
"""pairs/s through the Python drop-in entry points for a large batch: B headline pairs (the 1024-pair bench batch repeated) in ONE call of
estimate_batch_torch (resident tensors) and estimate_monodepth_relative_pose_batch(as_arrays=True) (pageable host buffers).  The chunking knobs
MDRP_PIPELINE_MIN is read when mdrp_amd.pipeline is imported: one process per setting.
    python tools/entry_rate.py [B] [repeats]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from mdrp_amd import pipeline, poselib, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 2000
b = synth.make_batch(0, 1024, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
idx = np.arange(B) % 1024
x1, x2, d1, d2 = (np.ascontiguousarray(b[k][idx]) for k in ("x1", "x2", "d1", "d2"))
cam = {"model": "SIMPLE_PINHOLE", "width": 1600, "height": 1200, "params": [800.0, 0.0, 0.0]}
ro = {"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}
bo = {"loss_type": "TRUNCATED_CAUCHY"}
dev = torch.device("cuda", 0)
t = [torch.from_numpy(a).to(dev) for a in (x1, x2, d1, d2)]
torch.cuda.synchronize()
poselib.estimate_batch_torch("calibrated", *t, cam, cam, ro, bo)
best_d = 0.0
for _ in range(reps):
    t0 = time.perf_counter()
    poselib.estimate_batch_torch("calibrated", *t, cam, cam, ro, bo)
    torch.cuda.synchronize()
    best_d = max(best_d, B / (time.perf_counter() - t0))
poselib.estimate_monodepth_relative_pose_batch(x1, x2, d1, d2, cam, cam, ro, bo, as_arrays=True)
best_h = 0.0
for _ in range(reps):
    t0 = time.perf_counter()
    poselib.estimate_monodepth_relative_pose_batch(x1, x2, d1, d2, cam, cam, ro, bo, as_arrays=True)
    best_h = max(best_h, B / (time.perf_counter() - t0))
print(f"B {B} chunks {len(pipeline.chunk_bounds(B))} x {pipeline.PIPELINE_CHUNK} (min {pipeline.PIPELINE_MIN}) depth {pipeline.PIPELINE_DEPTH}: "
      f"resident tensors {best_d:.0f} pairs/s, pageable host buffers {best_h:.0f} pairs/s (best of {reps})")
