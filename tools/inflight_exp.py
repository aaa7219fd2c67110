#!/usr/bin/env python3
"""Experiment: K consecutive 1024-pair steps driven by L host threads, each with its own handle and streams, so that L
steps are in flight at a time (step k + 1's solver / sweeps fill the SIMDs that step k's LM tails leave idle).
usage: inflight_exp.py [L ...]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mdrp_amd import _capi, synth

B, n, iters, K = int(os.environ.get("B", 1024)), 2000, 10000, int(os.environ.get("K", 24))
b = synth.make_batch(0, B, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
dev = torch.device("cuda", 0)
x1 = torch.from_numpy(b["x1"]).to(dev); x2 = torch.from_numpy(b["x2"]).to(dev)
d1 = torch.from_numpy(b["d1"]).to(dev); d2 = torch.from_numpy(b["d2"]).to(dev)
cams = np.zeros(B, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
ro = _capi.ransac_opt_from_dict({"max_iterations": iters, "min_iterations": iters, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
for L in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    hs = [_capi.Handle(0) for _ in range(L)]
    masks = [torch.zeros((B, n), dtype=torch.uint8, device=dev) for _ in range(L)]
    last = [None] * L
    def work(i, steps):
        for _ in range(steps):
            hs[i].estimate_batch_device(0, x1.data_ptr(), x2.data_ptr(), d1.data_ptr(), d2.data_ptr(), B, n, ro, bo, None, cams, cams, masks[i].data_ptr())
            last[i] = hs[i].fetch_results(B)
    for i in range(L):
        work(i, 1)  # warm-up (buffers)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i, K // L)) for i in range(L)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    steps = (K // L) * L
    print(f"in flight {L}: {steps} steps in {1e3 * t:.1f} ms = {1e3 * t / steps:.2f} ms/step  {B * steps / t:.0f} pairs/s  inl {np.mean(last[0]['num_inliers']) / n:.5f}", flush=True)
    for h in hs:
        h.close()
