#!/usr/bin/env python3
"""Wide GPU-vs-oracle comparison (not part of the test suite): for every estimator, many noisy pairs over a spread of
sizes, outlier rates, seeds and option sets.  Prints, per configuration, how many pairs land on exactly the oracle's
trajectory (iterations, refinements, inliers, mask) and the worst model deviation among those.  Run on the GPU box:
    python tests/tools/stress_parity.py [pairs_per_config]
MDRP_STRESS_ONLY="kind,N" restricts the run to one configuration; pairs off the oracle's trajectory are listed."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mdrp_amd import _capi as capi, synth  # noqa: E402
from oracle import pyorc as po  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
h = capi.Handle(0, None)
cam = po.cam_flat(0, [800.0, 0, 0])
tot = same_tot = 0
t0 = time.time()
for kind, es, rf in ((0, False, None), (0, True, None), (1, False, "shared"), (2, False, "varying")):
    for N, of, opts in ((150, 0.2, {}), (400, 0.5, {"min_iterations": 500}), (1000, 0.6, {"max_iterations": 3000, "min_iterations": 3000}),
                        (64, 0.0, {"min_iterations": 200, "seed": 7}), (2500, 0.35, {"max_iterations": 1500, "min_iterations": 1500, "seed": 3})):
        only = os.environ.get("MDRP_STRESS_ONLY")
        if only and (kind, N) != tuple(int(v) for v in only.split(",")):
            continue
        b = synth.make_batch(9000 + 37 * N + 11 * kind + int(es), B, N, noise_px=0.7, depth_noise=0.03, outlier_frac=of, random_focal=rf,
                             shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0)
        ro = {"max_epipolar_error": 2.0, "max_reproj_error": 16.0, "monodepth_estimate_shift": es, **opts}
        cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
        res, mask = h.estimate_batch(kind, b["x1"], b["x2"], b["d1"], b["d2"], capi.ransac_opt_from_dict(ro),
                                     capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), None,
                                     cams if kind == 0 else None, cams if kind == 0 else None)
        oro = po.ransac_opt(max_epipolar_error=2.0, max_reproj_error=16.0, estimate_shift=es,
                            **{k: v for k, v in opts.items()})
        same = 0
        it_same = 0
        worst = 0.0
        dinl = []
        for i in range(B):
            m, st, mk = po.estimate(kind, b["x1"][i], b["x2"][i], b["d1"][i], b["d2"][i], oro, po.bundle_opt(loss_type=4),
                                    cam if kind == 0 else None, cam if kind == 0 else None)
            it_same += int(res[i]["iterations"]) == st.iterations
            ok = (int(res[i]["iterations"]) == st.iterations and int(res[i]["refinements"]) == st.refinements
                  and int(res[i]["num_inliers"]) == st.num_inliers and (mask[i] == mk).all())
            same += ok
            if not ok:
                print(f"   pair {i}: gpu (ref {int(res[i]['refinements'])}, it {int(res[i]['iterations'])}, inl {int(res[i]['num_inliers'])}) "
                      f"oracle ({st.refinements}, {st.iterations}, {st.num_inliers}), mask differs on {int((mask[i] != mk).sum())}", flush=True)
            dinl.append(int(res[i]["num_inliers"]) - st.num_inliers)
            if ok:
                a = capi.model_to_array(res[i]["model"])
                r = np.array(list(m.q) + list(m.t) + [m.scale, m.shift1, m.shift2, m.f1, m.f2]) if hasattr(m, "q") else np.asarray(m, dtype=float)
                worst = max(worst, float(np.max(np.abs(a[:len(r)] - r) / (1e-12 + np.maximum(1.0, np.abs(r))))))
        tot += B; same_tot += same
        print(f"kind {kind} shift {int(es)} N {N:5d} outl {of:.2f} {opts}: same trajectory {same}/{B}, same iterations {it_same}/{B}, "
              f"mean inlier diff {np.mean(dinl):+.3f}, worst model dev on same-trajectory pairs {worst:.2e}", flush=True)
print(f"total {same_tot}/{tot} on the oracle's trajectory, {time.time() - t0:.0f} s")
