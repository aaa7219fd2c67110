#!/usr/bin/env python3
"""Wide GPU-vs-oracle comparison for the non-monodepth baselines (not part of the test suite): many noisy pairs over a spread of
sizes, outlier rates, thresholds and stopping rules.  Prints, per configuration, how many pairs land on exactly the oracle's
trajectory (iterations, refinements, inliers, mask) and the worst model deviation among those.  Run on the GPU box:
    python tests/tools/stress_parity_classic.py [pairs_per_config]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mdrp_amd import _capi as capi, synth  # noqa: E402
from oracle import pyorc as po  # noqa: E402
from test_oracle_classic import fund_diff, pose_diff  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
h = capi.Handle(0, None)
cam = po.cam_flat(0, [900.0, 640.0, 480.0])
tot = same_tot = 0
t0 = time.time()
KINDS = tuple(int(k) for k in os.environ.get("MDRP_STRESS_KINDS", "3,4,5").split(","))
for kind in KINDS:
    cases = ((150, 0.2, 1.0, {}), (400, 0.5, 2.0, {"min_iterations": 500}), (1000, 0.6, 1.5, {"max_iterations": 2000, "min_iterations": 2000}),
             (64, 0.0, 1.0, {"min_iterations": 200, "seed": 7}), (2000, 0.35, 0.75, {"max_iterations": 1500, "min_iterations": 1500, "seed": 3}))
    if kind == 4:  # the 6-point solver (20 x 20 eigenproblem per sample) is two orders of magnitude slower on both sides: bounded runs
        cases = ((150, 0.2, 1.0, {"max_iterations": 300, "min_iterations": 100}), (400, 0.5, 2.0, {"max_iterations": 500, "min_iterations": 500}),
                 (64, 0.0, 1.0, {"max_iterations": 200, "min_iterations": 200, "seed": 7}), (1000, 0.35, 0.75, {"max_iterations": 400, "min_iterations": 400, "seed": 3}))
    for N, of, thr, opts in cases:
        b = synth.make_batch(9300 + 37 * N + 11 * kind, B, N, f1=900.0, f2=900.0, pp=(640.0, 480.0), noise_px=0.7, outlier_frac=of)
        ro = {"max_epipolar_error": thr, **opts}
        cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 900.0; cams["params"][:, 1] = 640.0; cams["params"][:, 2] = 480.0
        cams4 = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams4["params"][:, 0] = 640.0; cams4["params"][:, 1] = 480.0  # kind 4: the principal point travels in cam1
        res, mask = h.estimate_batch(kind, b["x1"], b["x2"], None, None, capi.ransac_opt_from_dict(ro),
                                     capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY", "loss_scale": thr}), None,
                                     cams if kind == 3 else (cams4 if kind == 4 else None), cams if kind == 3 else None)
        oro = po.ransac_opt(max_epipolar_error=thr, **opts)
        same = 0
        worst = 0.0
        for i in range(B):
            m, st, mk = po.estimate_classic(kind, b["x1"][i], b["x2"][i], oro, po.bundle_opt(loss_type=4, loss_scale=thr), cam, cam, pp=(640.0, 480.0))
            ok = (int(res[i]["iterations"]) == st.iterations and int(res[i]["refinements"]) == st.refinements
                  and int(res[i]["num_inliers"]) == st.num_inliers and (mask[i] == mk).all())
            same += ok
            if ok:
                r = res[i]["model"]
                a = np.r_[r["q"], r["t"], r["scale"], r["shift1"]]
                worst = max(worst, pose_diff(a[:7], m) if kind in (3, 4) else fund_diff(a[:9], m))
            else:
                print(f"   pair {i}: gpu (ref {int(res[i]['refinements'])}, it {int(res[i]['iterations'])}, inl {int(res[i]['num_inliers'])}) "
                      f"oracle ({st.refinements}, {st.iterations}, {st.num_inliers})", flush=True)
        tot += B; same_tot += same
        print(f"kind {kind} N {N:5d} outl {of:.2f} thr {thr} {opts}: same trajectory {same}/{B}, worst model dev on those {worst:.2e}", flush=True)
print(f"total {same_tot}/{tot} on the oracle's trajectory, {time.time() - t0:.0f} s")
