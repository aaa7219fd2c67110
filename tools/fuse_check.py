#!/usr/bin/env python3
"""Fused tail on / off: results must be bit-identical; time per step of each.  usage: fuse_check.py [workload-kind 0|1|2] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mdrp_amd import _capi, synth

kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, n, iters = int(os.environ.get("B", 1024)), 2000, 10000
b = synth.make_batch(0, B, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5, random_focal=[None, "shared", "varying"][kind])
dev = torch.device("cuda", 0)
x1 = torch.from_numpy(b["x1"]).to(dev); x2 = torch.from_numpy(b["x2"]).to(dev)
d1 = torch.from_numpy(b["d1"]).to(dev); d2 = torch.from_numpy(b["d2"]).to(dev)
cams = np.zeros(B, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
ro = _capi.ransac_opt_from_dict({"max_iterations": iters, "min_iterations": iters, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
h = _capi.Handle(0)
out = {}
for mode in os.environ.get("MODES", "0,1,0,1").split(","):
    os.environ["MDRP_FUSE_TAIL"] = mode
    mask = torch.zeros((B, n), dtype=torch.uint8, device=dev)
    def step():
        h.estimate_batch_device(kind, x1.data_ptr(), x2.data_ptr(), d1.data_ptr(), d2.data_ptr(), B, n, ro, bo, None,
                                cams if kind == 0 else None, cams if kind == 0 else None, mask.data_ptr())
        return h.fetch_results(B)
    step(); step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bad = 0
    for _ in range(reps):
        res = step()
        if "0" in out and mode != "0":  # every step against the unfused records (stress for the cross-workgroup hand-over)
            r0, m0 = out["0"]
            bad += int(not (res.tobytes() == r0.tobytes() and np.array_equal(mask.cpu().numpy(), m0)))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    if bad:
        print(f"MISMATCH in {bad} of {reps} steps", flush=True)
    m = mask.cpu().numpy()
    key = mode
    same = ""
    if "0" in out and mode != "0":
        r0, m0 = out["0"]
        same = f"  identical to fuse=0: records {all(np.array_equal(res[k], r0[k]) for k in res.dtype.names)} masks {np.array_equal(m, m0)}"
    out.setdefault(key, (res, m))
    st = h.last_stats()
    print(f"fuse={mode}: {1e3 * dt:.2f} ms/step  {B / dt:.0f} pairs/s  lo {st['lo_ms']:.2f} ms final {st['final_ms']:.2f} ms{same}", flush=True)
