"""GPU box.  One dynamic-stopping fixture case by case: reference statistics, the HIP path and the oracle side by side:  python tests/tools/dbg_dynamic.py [name]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from mdrp_amd import _capi as capi
from test_oracle_golden import DYNAMIC_CASES, dynamic_pair
from helpers import model_diff
from oracle import pyorc as po
g = np.load("tests/golden/dynamic_ref.npz")
name = sys.argv[1] if len(sys.argv) > 1 else "shared"
kind, es, n, rf, _ = DYNAMIC_CASES[name]
ist = g[f"{name}_istats"]; B = len(ist)
pairs = [dynamic_pair(g, name, j) for j in range(B)]
x1, x2, d1, d2 = (np.ascontiguousarray(np.stack([p[k] for p in pairs])) for k in ("x1", "x2", "d1", "d2"))
cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
ro = capi.ransac_opt_from_dict({"max_iterations": 100000, "min_iterations": 1000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0, "monodepth_estimate_shift": es})
h = capi.Handle(0)
res, mask = h.estimate_batch(kind, x1, x2, d1, d2, ro, capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), None, cams if kind == 0 else None, cams if kind == 0 else None)
rm = np.unpackbits(g[f"{name}_mask"], axis=1)[:, :n]
bad = np.nonzero((mask != rm).any(axis=1))[0]
print("bad pairs", bad)
for j in bad:
    dj = np.nonzero(mask[j] != rm[j])[0]
    print(j, "outl", g["outliers"][j % 6], "bits differ", len(dj), dj[:10], "gpu sum", mask[j].sum(), "ref sum", rm[j].sum(), "stats gpu", res[j]["refinements"], res[j]["iterations"], res[j]["num_inliers"], res[j]["model_score"], "ref", ist[j], g[f"{name}_fstats"][j],
          "model diff", model_diff(capi.model_to_array(res[j]["model"]), g[f"{name}_model"][j]))
    ro_o = po.ransac_opt(max_iterations=100000, min_iterations=1000, max_epipolar_error=2.0, max_reproj_error=16.0, estimate_shift=es)
    m, st, mk = po.estimate(kind, x1[j], x2[j], d1[j], d2[j], ro_o, po.bundle_opt(loss_type=4), None, None)
    print("   oracle:", st.refinements, st.iterations, st.num_inliers, st.model_score, "mask vs ref differs", int((mk != rm[j]).sum()), "vs gpu", int((mk != mask[j]).sum()), "model vs ref", model_diff(m, g[f"{name}_model"][j]))
