import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mdrp_amd import _capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
n = 2000
b = synth.make_batch(0, 1024, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
idx = np.arange(B) % 1024
dev = torch.device("cuda", 0)
t = [torch.from_numpy(b[k]).to(dev).index_select(0, torch.from_numpy(idx).to(dev)).contiguous() for k in ("x1", "x2", "d1", "d2")]
mask = torch.zeros((B, n), dtype=torch.uint8, device=dev)
cams = np.zeros(B, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
ro = _capi.ransac_opt_from_dict({"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
h = _capi.Handle(0)
torch.cuda.synchronize()
outs = []
for r in range(4):
    h.estimate_batch_device(0, *(x.data_ptr() for x in t), B, n, ro, bo, None, cams, cams, mask.data_ptr())
    outs.append(h.fetch_results(B).copy())
    print("run", r, "stats", {k: v for k, v in h.last_stats().items() if "fuse" in k})
for r in range(1, 4):
    d = np.nonzero(outs[r].view(np.uint8).reshape(B, -1) != outs[0].view(np.uint8).reshape(B, -1))
    pairs = np.unique(d[0])
    print("run", r, "vs 0: differing pairs", len(pairs), pairs[:10])
    for p in pairs[:5]:
        print("  pair", p, "(mod 1024 =", p % 1024, ")", {f: (outs[0][p][f], outs[r][p][f]) for f in ("refinements", "iterations", "num_inliers", "model_score")})
# copies of one pair within a run
for r in range(2):
    first = outs[r][:1024]
    bad = [i for i in range(1024, B) if outs[r][i].tobytes() != first[i % 1024].tobytes()]
    print("run", r, "copies that differ from copy 0:", len(bad), bad[:10])
