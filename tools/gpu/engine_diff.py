"""diagnostic: where do the LM schedules (MDRP_LM_ENGINE = 0 / 1 / 2) differ?  prints per-pair differences"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mdrp_amd import _capi as capi, synth

kind, es, rf, loss = int(sys.argv[1]), bool(int(sys.argv[2])), [None, "shared", "varying"][int(sys.argv[1])], sys.argv[3]
B, N = 24, 900
ns = [N, 700, 257, 256, 255, 64, 5, 3, 2, 0, 300, 511] * 2
x1, x2 = np.zeros((B, N, 2)), np.zeros((B, N, 2))
d1, d2 = np.ones((B, N)), np.ones((B, N))
for i, n in enumerate(ns):
    if n:
        p = synth.make_pair(9500 + i, n, noise_px=0.5, depth_noise=0.02, outlier_frac=[0.4, 0.0, 0.2][i % 3], random_focal=rf,
                            shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0)
        x1[i, :n], x2[i, :n], d1[i, :n], d2[i, :n] = p["x1"], p["x2"], p["d1"], p["d2"]
cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
ro = capi.ransac_opt_from_dict({"max_iterations": 1200, "min_iterations": 1200, "max_epipolar_error": 2.0, "max_reproj_error": 16.0, "monodepth_estimate_shift": es})
bo = capi.bundle_opt_from_dict({"loss_type": loss, "max_iterations": 40})
h = capi.Handle(0)
out = []
for eng in ("0", "1", "2"):
    os.environ["MDRP_LM_ENGINE"] = eng
    res, mask = h.estimate_batch(kind, x1, x2, d1, d2, ro, bo, np.array(ns, dtype=np.int32), cams if kind == 0 else None, cams if kind == 0 else None)
    out.append((res.copy(), mask.copy()))
def flat(m):
    return np.c_[m["q"], m["t"], m["scale"], m["shift1"], m["shift2"], m["f1"], m["f2"]]
r0, m0 = out[0]
for e, (r, m) in zip("12", out[1:]):
    for i in range(B):
        dm = np.abs(flat(r["model"])[i] - flat(r0["model"])[i]).max()
        ds = abs(r["model_score"][i] - r0["model_score"][i]) / max(abs(r0["model_score"][i]), 1e-300)
        di = [int(r[f][i]) - int(r0[f][i]) for f in ("refinements", "iterations", "num_inliers")]
        dk = int((m[i] != m0[i]).sum())
        if dm > 1e-12 or ds > 1e-12 or any(di) or dk:
            print(f"engine {e} vs 0: pair {i} n {ns[i]} model {dm:.2e} score {ds:.2e} stats {di} mask {dk} inl {int(r0['num_inliers'][i])}")
print("done")
