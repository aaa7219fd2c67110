import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import mdrp_amd.poselib as poselib
from mdrp_amd import synth
from oracle import pyorc as po
from test_oracle_classic import pose_diff, fund_diff
for N in (9000, 30000, 70000):
    p = synth.make_pair(77, N, f1=900.0, f2=900.0, pp=(640.0, 480.0), noise_px=0.5, outlier_frac=0.4)
    cam = {"model": "SIMPLE_PINHOLE", "width": 1280, "height": 960, "params": [900.0, 640.0, 480.0]}
    ro = {"max_iterations": 300, "min_iterations": 300, "max_epipolar_error": 1.5}
    bo = {"loss_type": "TRUNCATED_CAUCHY", "loss_scale": 1.5}
    c = po.cam_flat(0, [900.0, 640.0, 480.0])
    for kind in (3, 5):
        if kind == 3:
            m, info = poselib.estimate_relative_pose(p["x1"], p["x2"], cam, cam, ro, bo); m = np.r_[m.q, m.t]
        else:
            m, info = poselib.estimate_fundamental(p["x1"], p["x2"], ro, bo); m = m.reshape(-1)
        om, st, mk = po.estimate_classic(kind, p["x1"], p["x2"], po.ransac_opt(max_iterations=300, min_iterations=300, max_epipolar_error=1.5), po.bundle_opt(loss_type=4, loss_scale=1.5), c, c)
        d = pose_diff(m, om) if kind == 3 else fund_diff(m, om)
        print(N, kind, (info["refinements"], info["iterations"], info["num_inliers"]), (st.refinements, st.iterations, st.num_inliers), "mask equal", np.array_equal(np.array(info["inliers"], dtype=np.uint8), mk), "model diff %.2e" % d, flush=True)
