#!/usr/bin/env python3
"""Oracle fixture for EVERY pair of the batches bench.py times: the 1024 pairs of BASELINE.json
configs[1] (calib_p3p_n2000_i10k; round 5: also calib_shift_n2000_i10k, the relpose_monodepth_3pt reading of it), configs[2] (shared_n2000_i10k) and configs[3] (varying_n5000_i10k with the shift flag set),
written to tests/golden/headline_<workload>.npz as outputs only — refinements, iterations, num_inliers, inlier_ratio,
model_score, the 12-wide model and the packed inlier mask per pair, plus a digest of each pair's inputs (the inputs
regenerate from mdrp_amd.synth, the generator bench.py uses).

The checker here is the CPU oracle (oracle/*.c), which tests/golden/estimate_wide.npz pins against the reference binary
on 32 pairs spread over each of these same batches (iterations, inliers, mask, model identical on 128 / 128).

    python3 tests/tools/gen_golden_headline.py [workload ...]        (8 worker processes, ~1-2 min per workload)
"""
import hashlib
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)

HEADLINE = {
    # workload (bench.py WORKLOADS): kind, estimate_shift flag handed to the estimator, n, outlier_frac, random_focal, (shift1, shift2) of the depths
    "calib_p3p_n2000_i10k": (0, False, 2000, 0.5, None, (0.0, 0.0)),
    "calib_shift_n2000_i10k": (0, True, 2000, 0.5, None, (0.2, -0.1)),
    "shared_n2000_i10k": (1, False, 2000, 0.5, "shared", (0.0, 0.0)),
    "varying_n5000_i10k": (2, True, 5000, 0.5, "varying", (0.0, 0.0)),
}
PAIRS = 1024
OPTS = dict(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, max_reproj_error=16.0, seed=0)


def input_digest(p):
    h = hashlib.sha256()
    for k in ("x1", "x2", "d1", "d2"):
        h.update(np.ascontiguousarray(p[k], dtype=np.float64).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def _work(args):
    workload, lo, hi = args
    from mdrp_amd import synth
    from oracle import pyorc as po
    kind, es, n, of, rf, (s1, s2) = HEADLINE[workload]
    cam = po.cam_flat(0, [800.0, 0.0, 0.0])
    ro = po.ransac_opt(estimate_shift=es, **OPTS)
    bo = po.bundle_opt(loss_type=4)
    rows = []
    for i in range(lo, hi):
        p = synth.make_pair(i, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf, shift1=s1, shift2=s2)
        m, st, mask = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], ro, bo, cam if kind == 0 else None, cam if kind == 0 else None)
        rows.append((i, m.copy(), (st.refinements, st.iterations, st.num_inliers), (st.inlier_ratio, st.model_score),
                     np.packbits(mask), input_digest(p)))
    return rows


def main():
    names = sys.argv[1:] or list(HEADLINE)
    workers = min(8, os.cpu_count() or 1)
    for w in names:
        step = 8
        jobs = [(w, lo, min(lo + step, PAIRS)) for lo in range(0, PAIRS, step)]
        with mp.get_context("fork").Pool(workers) as pool:
            rows = [r for chunk in pool.imap(_work, jobs, chunksize=1) for r in chunk]
        rows.sort(key=lambda r: r[0])
        assert [r[0] for r in rows] == list(range(PAIRS))
        kind, es, n, of, rf, _sh = HEADLINE[w]
        d = {"workload": np.array(w), "case": np.array([kind, int(es), n]), "outlier_frac": np.array(of),
             "model": np.array([r[1] for r in rows]), "istats": np.array([r[2] for r in rows], dtype=np.int64),
             "fstats": np.array([r[3] for r in rows]), "mask": np.array([r[4] for r in rows]),
             "digest": np.array([r[5] for r in rows], dtype=np.uint64)}
        out = os.path.join(HERE, "..", "golden", f"headline_{w}.npz")
        np.savez_compressed(out, **d)
        print(w, "pairs", PAIRS, "inliers min/mean/max", d["istats"][:, 2].min(), d["istats"][:, 2].mean(), d["istats"][:, 2].max(),
              "refinements mean", d["istats"][:, 0].mean(), os.path.getsize(out), "bytes", flush=True)


if __name__ == "__main__":
    main()
