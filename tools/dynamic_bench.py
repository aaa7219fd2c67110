import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mdrp_amd import _capi, synth
B, n = 1024, 2000
b = synth.make_batch(0, B, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
dev = torch.device("cuda", 0)
t = [torch.from_numpy(b[k]).to(dev) for k in ("x1", "x2", "d1", "d2")]
mask = torch.zeros((B, n), dtype=torch.uint8, device=dev)
cams = np.zeros(B, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
h = _capi.Handle(0)
torch.cuda.synchronize()
for name, rd in (("fixed 10000", {"max_iterations": 10000, "min_iterations": 10000}), ("defaults (max 100000, min 1000, dynamic)", {}), ("max 10000 min 1000 dynamic", {"max_iterations": 10000, "min_iterations": 1000})):
    ro = _capi.ransac_opt_from_dict({**rd, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
    ts = []
    for r in range(5):
        t0 = time.perf_counter()
        h.estimate_batch_device(0, *(x.data_ptr() for x in t), B, n, ro, bo, None, cams, cams, mask.data_ptr())
        res = h.fetch_results(B)
        ts.append(time.perf_counter() - t0)
    st = h.last_stats()
    print(f"{name}: {[round(1e3 * x, 2) for x in ts]} ms -> {B / np.median(ts):.0f} pairs/s; iterations mean {res['iterations'].mean():.0f} max {res['iterations'].max()}; kernels ms "
          f"{ {k: round(st[k], 2) for k in ('solve_ms', 'count_ms', 'bound_ms', 'sweep_ms', 'lo_ms', 'final_ms')} }")
