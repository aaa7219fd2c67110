import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mdrp_amd import _capi as capi, synth
free0 = torch.cuda.mem_get_info()[0]
b = synth.make_batch(0, 64, 500, noise_px=0.5, depth_noise=0.02, outlier_frac=0.4)
cams = np.zeros(64, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
ro = capi.ransac_opt_from_dict({"max_epipolar_error": 2.0, "max_reproj_error": 16.0})
bo = capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
ref = None
t0 = time.time()
for it in range(40):                      # create / use / destroy
    h = capi.Handle(0, None)
    res, mask = h.estimate_batch(0, b["x1"], b["x2"], b["d1"], b["d2"], ro, bo, None, cams, cams)
    if ref is None: ref = (res.tobytes(), mask.tobytes())
    assert (res.tobytes(), mask.tobytes()) == ref, it
    del h
print("create/destroy x40 ok, %.1f s" % (time.time() - t0), "free delta MB", (free0 - torch.cuda.mem_get_info()[0]) / 2**20)
h = capi.Handle(0, None)
t0 = time.time()
for it in range(400):                     # same handle, alternating shapes
    n = 64 if it % 2 else 17
    res, mask = h.estimate_batch(0, b["x1"][:n], b["x2"][:n], b["d1"][:n], b["d2"][:n], ro, bo, None, cams[:n], cams[:n])
    if n == 64: assert (res.tobytes(), mask.tobytes()) == ref, it
print("400 calls ok, %.1f s" % (time.time() - t0), "free delta MB", (free0 - torch.cuda.mem_get_info()[0]) / 2**20)
# the non-monodepth baselines interleaved with the monodepth estimator on one handle: determinism and no growth
refs = {}
t0 = time.time()
for it in range(300):
    kind = (3, 5, 0)[it % 3]
    n = (64, 17, 40)[it % 3]
    if kind == 0:
        res, mask = h.estimate_batch(0, b["x1"][:n], b["x2"][:n], b["d1"][:n], b["d2"][:n], ro, bo, None, cams[:n], cams[:n])
    else:
        res, mask = h.estimate_batch(kind, b["x1"][:n], b["x2"][:n], None, None, ro, bo, None, cams[:n] if kind == 3 else None, cams[:n] if kind == 3 else None)
    key = (kind, n)
    if key not in refs: refs[key] = (res.tobytes(), mask.tobytes())
    assert (res.tobytes(), mask.tobytes()) == refs[key], (it, key)
print("300 interleaved baseline / monodepth calls ok, %.1f s" % (time.time() - t0), "free delta MB", (free0 - torch.cuda.mem_get_info()[0]) / 2**20)
