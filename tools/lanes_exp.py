#!/usr/bin/env python3
"""Experiment: one 1024-pair step as L sub-batches on L handles / host threads (does overlapping the LM tails of one
sub-batch with the front kernels of the next pay?).  usage: lanes_exp.py [L ...]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mdrp_amd import _capi, synth

B, n, iters = 1024, 2000, 10000
b = synth.make_batch(0, B, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
dev = torch.device("cuda", 0)
x1 = torch.from_numpy(b["x1"]).to(dev); x2 = torch.from_numpy(b["x2"]).to(dev)
d1 = torch.from_numpy(b["d1"]).to(dev); d2 = torch.from_numpy(b["d2"]).to(dev)
mask = torch.zeros((B, n), dtype=torch.uint8, device=dev)
cams = np.zeros(B, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
ro = _capi.ransac_opt_from_dict({"max_iterations": iters, "min_iterations": iters, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
for L in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    hs = [_capi.Handle(0) for _ in range(L)]
    per = B // L
    out = [None] * L
    def work(i, stagger):
        time.sleep(stagger)
        o = i * per
        hs[i].estimate_batch_device(0, x1[o:].data_ptr(), x2[o:].data_ptr(), d1[o:].data_ptr(), d2[o:].data_ptr(), per, n, ro, bo, None,
                                    cams[o:o + per], cams[o:o + per], mask[o:].data_ptr())
        out[i] = hs[i].fetch_results(per)
    for stagger_ms in (0.0, 1.0, 2.0, 3.0):
        ts = []
        for rep in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(i, i * stagger_ms * 1e-3)) for i in range(L)]
            [t.start() for t in th]; [t.join() for t in th]
            ts.append(time.perf_counter() - t0)
        t = min(ts[1:])
        print(f"lanes {L} stagger {stagger_ms} ms: {1e3 * t:.2f} ms/step  {B / t:.0f} pairs/s  inl {np.mean(np.concatenate([o['num_inliers'] for o in out])) / n:.5f}", flush=True)
        if L == 1:
            break
    for h in hs:
        h.close()
