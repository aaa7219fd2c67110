"""Synthetic image-pair generator for parity tests and bench.py.

The reference has no synthetic generator (it reads H5 datasets, /root/reference/eval.py:314-349);
this one produces the shapes BASELINE.json's configs name.  seed = 1234 + pair_index, numpy default_rng.
Geometry convention (reference README.md:103, wheel METADATA:264-272):
    R (d1+shift1) K1^-1 x1 + t = scale (d2+shift2) K2^-1 x2
"""
import numpy as np


def rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def make_pair(index, n, *, f1=800.0, f2=800.0, pp=(0.0, 0.0), noise_px=0.5, depth_noise=0.02, outlier_frac=0.0,
              shift1=0.0, shift2=0.0, random_focal=None, width=1600, height=1200):
    """Returns dict with x1,x2 (n,2) pixels, d1,d2 (n,), and ground truth R,t,scale,shift1,shift2,f1,f2.

    random_focal: None | 'shared' | 'varying'  -> focal(s) drawn from U(300,2000).
    """
    rng = np.random.default_rng(1234 + index)
    if random_focal == "shared":
        f1 = f2 = rng.uniform(300.0, 2000.0)
    elif random_focal == "varying":
        f1, f2 = rng.uniform(300.0, 2000.0, size=2)
    R = rodrigues(rng.normal(0.0, 0.3, 3))
    t = rng.normal(0.0, 0.5, 3)
    scale = rng.uniform(0.3, 3.0)
    X1 = np.zeros((0, 3))
    while len(X1) < n:
        m = 2 * (n - len(X1)) + 16
        P = np.stack([rng.uniform(-2, 2, m), rng.uniform(-1.5, 1.5, m), rng.uniform(2, 8, m)], axis=1)
        P2 = P @ R.T + t
        X1 = np.concatenate([X1, P[P2[:, 2] >= 0.5]], axis=0)
    X1 = X1[:n]
    X2 = X1 @ R.T + t
    pp = np.asarray(pp, dtype=np.float64)
    x1 = f1 * X1[:, :2] / X1[:, 2:3] + pp + rng.normal(0.0, 1.0, (n, 2)) * noise_px
    x2 = f2 * X2[:, :2] / X2[:, 2:3] + pp + rng.normal(0.0, 1.0, (n, 2)) * noise_px
    d1 = (X1[:, 2] - shift1) * (1.0 + rng.normal(0.0, 1.0, n) * depth_noise)
    d2 = (X2[:, 2] / scale - shift2) * (1.0 + rng.normal(0.0, 1.0, n) * depth_noise)
    n_out = int(round(outlier_frac * n))
    is_outlier = np.zeros(n, dtype=bool)
    if n_out > 0:
        rows = rng.choice(n, n_out, replace=False)
        is_outlier[rows] = True
        x2[rows, 0] = rng.uniform(-width / 2, width / 2, n_out) + pp[0]
        x2[rows, 1] = rng.uniform(-height / 2, height / 2, n_out) + pp[1]
        d2[rows] = rng.uniform(1.0, 5.0, n_out)
    return dict(x1=np.ascontiguousarray(x1), x2=np.ascontiguousarray(x2), d1=d1, d2=d2, R=R, t=t, scale=scale,
                shift1=shift1, shift2=shift2, f1=float(f1), f2=float(f2), pp=pp, is_outlier=is_outlier)


def make_batch(first_index, batch, n, **kw):
    """Stacked arrays for `batch` pairs: x1,x2 (B,n,2), d1,d2 (B,n) + list of ground truths."""
    pairs = [make_pair(first_index + i, n, **kw) for i in range(batch)]
    out = {k: np.ascontiguousarray(np.stack([p[k] for p in pairs])) for k in ("x1", "x2", "d1", "d2")}
    out["gt"] = pairs
    return out


def rotation_error_deg(R_gt, R):
    """/root/reference/utils/data.py:50-57 : 2 asin(||R_gt - R||_F / (2 sqrt 2))."""
    s = np.linalg.norm(R_gt - R) / (2.0 * np.sqrt(2.0))
    return float(np.degrees(2.0 * np.arcsin(np.clip(s, 0.0, 1.0))))


def translation_error_deg(t_gt, t):
    """/root/reference/utils/data.py:60-70 : angle between directions, sign-agnostic, eps-guarded."""
    eps = 1e-15
    t = t / (np.linalg.norm(t) + eps)
    t_gt = t_gt / (np.linalg.norm(t_gt) + eps)
    loss_t = np.maximum(eps, (1.0 - np.sum(t * t_gt) ** 2))
    return float(np.degrees(np.arccos(np.sqrt(1 - loss_t))))
