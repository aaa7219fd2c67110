#!/usr/bin/env python3
"""Wide full-size pin of the non-monodepth baselines against the REFERENCE binary: 32 pairs of the 5-point and the 7-point
estimator and 16 of the 6-point one at the benchmark shape (N = 2000, 10^4 iterations, 50 % outliers), stored as seeds +
outputs only (tests/golden/classic_wide.npz); inputs regenerate from mdrp_amd.synth, a digest of each pair's inputs is stored.
The CPU oracle runs beside the reference: `oracle_refinements` / `oracle_same` record where the port itself deviates
(DESIGN.md §8a), so the GPU test's tolerance set is data, not hand-picked.

Runs only in the build container (needs oracle/_ref/librefshim.so):   python3 tests/tools/gen_golden_wide_classic.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import refshim as rs  # noqa: E402
from mdrp_amd import synth  # noqa: E402
from oracle import pyorc as po  # noqa: E402

OUT = os.path.join(HERE, "..", "golden", "classic_wide.npz")
CASES = (("relpose_5pt", 3, 32), ("shared_6pt", 4, 16), ("fundamental_7pt", 5, 32))
N, F = 2000, 800.0


def wide_pair(kind, index):
    if kind == 4:
        return synth.make_pair(7000 + index, N, noise_px=0.5, outlier_frac=0.5, random_focal="shared", pp=(0.0, 0.0))
    return synth.make_pair(7000 + index, N, f1=F, f2=F, pp=(0.0, 0.0), noise_px=0.5, outlier_frac=0.5)


def input_digest(p):
    h = hashlib.sha256()
    for k in ("x1", "x2"):
        h.update(np.ascontiguousarray(p[k], dtype=np.float64).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def pose_diff(a, b):
    dq = min(np.abs(a[:4] - b[:4]).max(), np.abs(a[:4] + b[:4]).max())
    return dq + np.abs(a[4:7] / np.linalg.norm(a[4:7]) - b[4:7] / np.linalg.norm(b[4:7])).max()


def fund_diff(a, b):
    a, b = a[:9] / np.linalg.norm(a[:9]), b[:9] / np.linalg.norm(b[:9])
    return min(np.abs(a - b).max(), np.abs(a + b).max())


def main():
    d = {"names": np.array([c[0] for c in CASES]), "kinds": np.array([c[1] for c in CASES]), "counts": np.array([c[2] for c in CASES])}
    cam_r = rs.cam_flat(0, 1600, 1200, [F, 0.0, 0.0])
    cam_o = po.cam_flat(0, [F, 0.0, 0.0])
    for name, kind, count in CASES:
        models, stats, masks, digests, orefs, osame = [], [], [], [], [], []
        for index in range(count):
            p = wide_pair(kind, index)
            kw = dict(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, seed=0)
            m, st, mask = rs.estimate_classic(kind, p["x1"], p["x2"], rs.ropt(**kw), rs.bopt(loss_type=4), cam_r if kind == 3 else None,
                                              cam_r if kind == 3 else None, pp=(0.0, 0.0))
            mo, sto, masko = po.estimate_classic(kind, p["x1"], p["x2"], po.ransac_opt(**kw), po.bundle_opt(loss_type=4),
                                                 cam_o if kind == 3 else None, cam_o if kind == 3 else None, pp=(0.0, 0.0))
            m, mo = np.asarray(m, float).reshape(-1), np.asarray(mo, float).reshape(-1)
            md = pose_diff(m, mo) if kind == 3 else (fund_diff(m, mo) if kind == 5 else pose_diff(m, mo) + abs(m[7] - mo[10]) / abs(m[7]))  # oracle: 12-wide Model, f1 at 10
            same = sto.iterations == int(st[1]) and sto.num_inliers == int(st[2]) and (masko == mask).all() and md < 1e-6
            full = np.zeros(9); full[: len(m)] = m
            models.append(full); stats.append(st); masks.append(np.packbits(mask)); digests.append(input_digest(p))
            orefs.append(sto.refinements); osame.append(same)
            print(name, index, "ref stats", st, "oracle refinements", sto.refinements, "same result", same, "model diff %.2e" % md, flush=True)
        d[f"{name}_model"] = np.array(models); d[f"{name}_stats"] = np.array(stats); d[f"{name}_mask"] = np.array(masks)
        d[f"{name}_digest"] = np.array(digests, dtype=np.uint64)
        d[f"{name}_oracle_refinements"] = np.array(orefs); d[f"{name}_oracle_same"] = np.array(osame)
    np.savez_compressed(OUT, **d)
    for name, kind, count in CASES:
        dev = d[f"{name}_oracle_refinements"] - d[f"{name}_stats"][:, 0].astype(int)
        print(name, "pairs", count, "oracle == reference result on", int(d[f"{name}_oracle_same"].sum()),
              "LO-count deviations (oracle - reference):", {i: int(v) for i, v in enumerate(dev) if v})
    print(os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
