import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from mdrp_amd import _capi as capi, synth
from oracle import pyorc as po
h = capi.Handle(0, None)
cam = po.cam_flat(0, [800.0, 0, 0])
for kind, rf, N, its in ((0, None, 9000, 600), (0, None, 20000, 400), (2, "varying", 12000, 300), (1, "shared", 70000, 200), (0, None, 3, 100), (0, None, 4, 100), (0, None, 5, 100)):
    B = 3
    b = synth.make_batch(7000 + N, B, N, noise_px=0.5, depth_noise=0.02, outlier_frac=0.3 if N > 10 else 0.0, random_focal=rf)
    ro = {"max_iterations": its, "min_iterations": its, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}
    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    t0 = time.time()
    res, mask = h.estimate_batch(kind, b["x1"], b["x2"], b["d1"], b["d2"], capi.ransac_opt_from_dict(ro), capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}),
                                 None, cams if kind == 0 else None, cams if kind == 0 else None)
    tg = time.time() - t0
    oro = po.ransac_opt(max_iterations=its, min_iterations=its, max_epipolar_error=2.0, max_reproj_error=16.0)
    same = 0; dev = 0.0
    for i in range(B):
        m, st, mk = po.estimate(kind, b["x1"][i], b["x2"][i], b["d1"][i], b["d2"][i], oro, po.bundle_opt(loss_type=4), cam if kind == 0 else None, cam if kind == 0 else None)
        ok = int(res[i]["iterations"]) == st.iterations and int(res[i]["refinements"]) == st.refinements and int(res[i]["num_inliers"]) == st.num_inliers and (mask[i] == mk).all()
        same += ok
        a = capi.model_to_array(res[i]["model"])
        dev = max(dev, float(np.nanmax(np.abs(a - m) / np.maximum(1.0, np.abs(m)))))
    print(f"kind {kind} N {N} its {its}: same trajectory {same}/{B}, worst model dev {dev:.2e}, inliers {res['num_inliers'].tolist()}, gpu {tg*1e3:.1f} ms", flush=True)
