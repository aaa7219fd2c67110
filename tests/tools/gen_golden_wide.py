#!/usr/bin/env python3
"""Wide full-size pin against the REFERENCE binary: 32 pairs per estimator at BASELINE.json's shapes, stored as
seeds + outputs only (tests/golden/estimate_wide.npz).  Inputs regenerate from mdrp_amd.synth (the generator bench.py
uses); a checksum of each pair's inputs is stored so that a drifting generator is detected instead of misread as a
parity failure.  The calibrated P3P pairs are indices 0, 33, 66, ..., 1023 of bench.py's 1024-pair headline batch, so
the GPU test can run THAT batch and compare the pairs spread over it with the reference's own output.

For every pair the CPU oracle is run beside the reference: `oracle_refinements` records the port's LO count (the
product shares the port's solvers), so the deviation set is enumerated by data, not by hand.

Runs only in the build container (needs oracle/_ref/librefshim.so):   python3 tests/tools/gen_golden_wide.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import refshim as rs  # noqa: E402
from mdrp_amd import synth  # noqa: E402
from oracle import pyorc as po  # noqa: E402

OUT = os.path.join(HERE, "..", "golden", "estimate_wide.npz")

WIDE_CASES = (
    # name, kind, estimate_shift, n, outlier_frac, random_focal     (BASELINE.json configs[1..3], as bench.py's WORKLOADS)
    ("calib_p3p", 0, 0, 2000, 0.5, None),
    ("calib_shift", 0, 1, 2000, 0.5, None),
    ("shared", 1, 0, 2000, 0.5, "shared"),
    ("varying_shiftflag", 2, 1, 5000, 0.5, "varying"),
)
INDICES = [33 * i for i in range(32)]  # 0 .. 1023, spread over bench.py's batch


def wide_pair(kind, es, n, of, rf, index):
    return synth.make_pair(index, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf,
                           shift1=0.2 if es and kind == 0 else 0.0, shift2=-0.1 if es and kind == 0 else 0.0)


def input_digest(p):
    h = hashlib.sha256()
    for k in ("x1", "x2", "d1", "d2"):
        h.update(np.ascontiguousarray(p[k], dtype=np.float64).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def main():
    d = {"indices": np.array(INDICES), "names": np.array([c[0] for c in WIDE_CASES]),
         "cases": np.array([[c[1], c[2], c[3]] for c in WIDE_CASES]), "outlier_frac": np.array([c[4] for c in WIDE_CASES])}
    cam_r = rs.cam_flat(0, 1600, 1200, [800.0, 0.0, 0.0])
    cam_o = po.cam_flat(0, [800.0, 0.0, 0.0])
    for name, kind, es, n, of, rf in WIDE_CASES:
        models, stats, masks, digests, orefs, osame = [], [], [], [], [], []
        for index in INDICES:
            p = wide_pair(kind, es, n, of, rf, index)
            kw = dict(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, max_reproj_error=16.0, seed=0, estimate_shift=bool(es))
            m, st, mask = rs.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], rs.ropt(**kw), rs.bopt(loss_type=4),
                                      cam_r if kind == 0 else None, cam_r if kind == 0 else None)
            mo, sto, masko = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], po.ransac_opt(**kw), po.bundle_opt(loss_type=4),
                                         cam_o if kind == 0 else None, cam_o if kind == 0 else None)
            m12 = np.r_[m, 1.0, 1.0] if kind == 0 else m
            same = (sto.iterations == int(st[1]) and sto.num_inliers == int(st[2]) and (masko == mask).all()
                    and np.abs(mo - m12).max() < 1e-6 * (1 + np.abs(m12).max()))
            models.append(m12); stats.append(st); masks.append(np.packbits(mask)); digests.append(input_digest(p))
            orefs.append(sto.refinements); osame.append(same)
            print(name, index, "ref stats", st, "oracle refinements", sto.refinements, "same result", same, flush=True)
        d[f"{name}_model"] = np.array(models); d[f"{name}_stats"] = np.array(stats); d[f"{name}_mask"] = np.array(masks)
        d[f"{name}_digest"] = np.array(digests, dtype=np.uint64)
        d[f"{name}_oracle_refinements"] = np.array(orefs); d[f"{name}_oracle_same"] = np.array(osame)
    np.savez_compressed(OUT, **d)
    for name, *_ in WIDE_CASES:
        dev = d[f"{name}_oracle_refinements"] - d[f"{name}_stats"][:, 0].astype(int)
        print(name, "pairs", len(INDICES), "oracle == reference result on", int(d[f"{name}_oracle_same"].sum()),
              "LO-count deviations (oracle - reference):", {int(INDICES[i]): int(v) for i, v in enumerate(dev) if v})


if __name__ == "__main__":
    main()
