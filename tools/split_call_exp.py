#!/usr/bin/env python3
"""Experiment: ONE 1024-pair step cut into P parts that go through a BatchPipeline(depth D) and are all waited for before the next step starts
(no overlap across steps: what a single estimate call could do internally).  usage: split_call_exp.py P D [P D ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mdrp_amd import _capi, synth
from mdrp_amd.pipeline import BatchPipeline

B, n, iters, K = 1024, 2000, 10000, int(os.environ.get("K", 16))
b = synth.make_batch(0, B, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
dev = torch.device("cuda", 0)
x1 = torch.from_numpy(b["x1"]).to(dev); x2 = torch.from_numpy(b["x2"]).to(dev)
d1 = torch.from_numpy(b["d1"]).to(dev); d2 = torch.from_numpy(b["d2"]).to(dev)
mask = torch.zeros((B, n), dtype=torch.uint8, device=dev)
cams = np.zeros(B, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
ro = _capi.ransac_opt_from_dict({"max_iterations": iters, "min_iterations": iters, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
args = [int(a) for a in sys.argv[1:]] or [1, 1, 2, 2, 4, 2]
for P, D in zip(args[::2], args[1::2]):
    pipe = BatchPipeline(depth=D, device=0)
    cuts = [(B * k // P, B * (k + 1) // P) for k in range(P)]
    def step():
        futs = [pipe.submit_device(0, x1.data_ptr() + 16 * n * lo, x2.data_ptr() + 16 * n * lo, d1.data_ptr() + 8 * n * lo, d2.data_ptr() + 8 * n * lo, hi - lo, n, ro, bo,
                                   None, cams[lo:hi], cams[lo:hi], mask.data_ptr() + n * lo) for lo, hi in cuts]
        return np.concatenate([f.result() for f in futs])
    for _ in range(3): r = step()
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K): r = step()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"parts {P} depth {D}: {1e3 * t / K:.2f} ms/step  {B * K / t:.0f} pairs/s  inl {np.mean(r['num_inliers']) / n:.5f}", flush=True)
    pipe.close()
