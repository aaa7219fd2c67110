#!/usr/bin/env python3
"""REFERENCE-BINARY fixtures for the timed batches of the comparison rows and of the outlier-free shape: every pair of
bench.py's `relpose_5pt_n2000_i10k` and `fundamental_7pt_n2000_i10k` (1024 pairs), `shared_6pt_n2000_i10k` (its 256-pair bench batch) and
`calib_p3p_n2000_i10k_clean` (1024 pairs, 0 % outliers) through the reference's own PoseLib binary (estimate_relative_pose @0x21f800,
estimate_shared_focal_relative_pose, estimate_fundamental @0x221a00, estimate_monodepth_relative_pose @0x224170), and beside it the CPU oracle's LO
count and a flag "oracle result == reference result" per pair (as tests/tools/gen_golden_wide_classic.py does for its 80 pairs).
Outputs only: tests/golden/headline_ref_<workload>.npz (stats, model, packed mask, input digest, oracle_refinements, oracle_same).

Runs only in the build container:   bash oracle/build_ref.sh && python3 tests/tools/gen_golden_headline_ref_classic.py [workload ...]      (8 workers)"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import gen_golden_headline_ref as gh  # noqa: E402
from gen_golden_wide_classic import fund_diff, pose_diff  # noqa: E402

SETS = {
    # workload (bench.py WORKLOADS): kind, pairs, outlier_frac, random_focal
    "relpose_5pt_n2000_i10k": (3, 1024, 0.5, None),
    "fundamental_7pt_n2000_i10k": (5, 1024, 0.5, None),
    "shared_6pt_n2000_i10k": (4, 256, 0.5, "shared"),
    "calib_p3p_n2000_i10k_clean": (0, 1024, 0.0, None),
}
OPTS = dict(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, max_reproj_error=16.0, seed=0)


def make_pair(workload, i):
    from mdrp_amd import synth
    kind, pairs, of, rf = SETS[workload]
    return synth.make_pair(i, 2000, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf)


def _work(args):
    workload, lo, hi = args
    import refshim as rs
    from oracle import pyorc as po
    kind, pairs, of, rf = SETS[workload]
    cam_r, cam_o = rs.cam_flat(0, 1600, 1200, [800.0, 0.0, 0.0]), po.cam_flat(0, [800.0, 0.0, 0.0])
    rows = []
    for i in range(lo, hi):
        p = make_pair(workload, i)
        gh._srand(1)
        if kind == 0:
            m, st, mask = rs.estimate(0, p["x1"], p["x2"], p["d1"], p["d2"], rs.ropt(**OPTS), rs.bopt(loss_type=4), cam_r, cam_r)
            mo, sto, masko = po.estimate(0, p["x1"], p["x2"], p["d1"], p["d2"], po.ransac_opt(**OPTS), po.bundle_opt(loss_type=4), cam_o, cam_o)
            m = np.r_[m, 1.0, 1.0]
            from helpers_path import model_diff
            md = model_diff(mo, m)
            full = m
        else:
            kw = {k: v for k, v in OPTS.items() if k != "max_reproj_error"}
            m, st, mask = rs.estimate_classic(kind, p["x1"], p["x2"], rs.ropt(**kw), rs.bopt(loss_type=4), cam_r if kind == 3 else None, cam_r if kind == 3 else None, pp=(0.0, 0.0))
            mo, sto, masko = po.estimate_classic(kind, p["x1"], p["x2"], po.ransac_opt(**kw), po.bundle_opt(loss_type=4), cam_o if kind == 3 else None, cam_o if kind == 3 else None, pp=(0.0, 0.0))
            m, mo = np.asarray(m, float).reshape(-1), np.asarray(mo, float).reshape(-1)
            md = pose_diff(m, mo) if kind == 3 else (fund_diff(m, mo) if kind == 5 else pose_diff(m, mo) + abs(m[7] - mo[10]) / abs(m[7]))
            full = np.zeros(12); full[: len(m)] = m
        same = sto.iterations == int(st[1]) and sto.num_inliers == int(st[2]) and bool((masko == mask).all()) and md < 1e-6
        rows.append((i, full, (int(st[0]), int(st[1]), int(st[2])), (float(st[3]), float(st[4])), np.packbits(mask), gh.input_digest(p), int(sto.refinements), bool(same), float(md)))
    return rows


def main():
    sys.modules["helpers_path"] = __import__("importlib").import_module("helpers") if os.path.join(HERE, "..") in sys.path else None
    names = sys.argv[1:] or list(SETS)
    for w in names:
        kind, pairs, of, rf = SETS[w]
        t0 = time.perf_counter()
        jobs = [(w, lo, min(lo + 4, pairs)) for lo in range(0, pairs, 4)]
        with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
            rows = sorted((r for chunk in pool.imap(_work, jobs, chunksize=1) for r in chunk), key=lambda r: r[0])
        d = {"workload": np.array(w), "kind": np.array(kind), "outlier_frac": np.array(of), "srand": np.array(1),
             "model": np.array([r[1] for r in rows]), "istats": np.array([r[2] for r in rows], dtype=np.int64), "fstats": np.array([r[3] for r in rows]),
             "mask": np.array([r[4] for r in rows]), "digest": np.array([r[5] for r in rows], dtype=np.uint64),
             "oracle_refinements": np.array([r[6] for r in rows], dtype=np.int64), "oracle_same": np.array([r[7] for r in rows]), "oracle_model_diff": np.array([r[8] for r in rows])}
        out = os.path.join(HERE, "..", "golden", f"headline_ref_{w}.npz")
        np.savez_compressed(out, **d)
        dev = d["oracle_refinements"] - d["istats"][:, 0]
        print(w, "pairs", pairs, "oracle == reference result on", int(d["oracle_same"].sum()), "not same:", np.nonzero(~d["oracle_same"])[0].tolist()[:20],
              "LO-count deviations (oracle - reference):", {int(i): int(v) for i, v in enumerate(dev) if v}, os.path.getsize(out), "bytes", f"{time.perf_counter() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(HERE, ".."))
    main()
