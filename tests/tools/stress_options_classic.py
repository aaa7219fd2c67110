#!/usr/bin/env python3
"""A larger one-off version of the comparison rows' randomised options fixture (tests/golden/options_ref_classic.npz: 64 cases per row), directly against the
reference binary:
    build container:  python3 tests/tools/stress_options_classic.py gen CASES SEED build/options_ref_classic_SEED.npz      (reference shim; 8 workers)
    GPU box        :  python3 tests/tools/stress_options_classic.py run CASES SEED build/options_ref_classic_SEED.npz      (HIP path, drop-in entry points)
Same case generator as the fixture's (size, outliers, noise, threshold, seed, fixed or dynamic budget, loss type / scale, bundle cap, cameras, principal
point), another seed.  Counts the cases on which iterations, inlier count, mask and model (1e-6) equal the reference's, and the LO-count differences."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", ".."))
from helpers import CLASSIC_OPTIONS_COLS as COLS, CLASSIC_OPTIONS_KINDS as KINDS, classic_options_cameras, classic_options_pair  # noqa: E402


def pose_diff(a, b):  # (tests/test_oracle_classic.py: rotation up to the quaternion's sign, translation DIRECTION — |t| is a gauge of the 5-parameter LM)
    a, b = np.asarray(a, float), np.asarray(b, float)
    dq = min(np.abs(a[:4] - b[:4]).max(), np.abs(a[:4] + b[:4]).max())
    return dq + np.abs(a[4:7] / np.linalg.norm(a[4:7]) - b[4:7] / np.linalg.norm(b[4:7])).max()


def fund_diff(a, b):
    a, b = np.asarray(a, float).reshape(-1)[:9], np.asarray(b, float).reshape(-1)[:9]
    a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max())


def case_table(seed, cases):
    rng = np.random.default_rng(seed)
    t = np.zeros((cases, len(COLS)))
    for j in range(cases):
        budget = [(300, 300), (1500, 1500), (2000, 100), (100000, 1000)][int(rng.integers(0, 4))]
        t[j] = (int(rng.choice([40, 60, 150, 400, 900, 1500])), float(rng.choice([0.0, 0.2, 0.4, 0.6])), float(rng.choice([0.25, 0.5, 1.0])),
                float(rng.choice([0.5, 1.0, 2.0, 4.0])), int(rng.integers(0, 1000)), budget[0], budget[1], j % 6, float(rng.choice([0.5, 1.0, 3.0])),
                int(rng.choice([0, 5, 100, 100])), float(rng.choice([500.0, 800.0, 1400.0])), float(rng.choice([500.0, 800.0, 1400.0])),
                float(rng.choice([0.0, 640.0, 3.0])), float(rng.choice([0.0, 480.0, -2.0])), int(rng.integers(0, 2)))
    return t


def _ref(args):
    name, j, row = args
    import gen_golden_headline_ref as gh
    import refshim as rs
    kind = KINDS[name]
    p = classic_options_pair(name, j, row)
    ro = rs.ropt(max_iterations=int(row[5]), min_iterations=int(row[6]), max_epipolar_error=float(row[3]), seed=int(row[4]))
    bo = rs.bopt(max_iterations=int(row[9]), loss_type=int(row[7]), loss_scale=float(row[8]), gradient_tol=1e-10)
    c1, c2 = classic_options_cameras(row)
    cam1, cam2 = (rs.cam_flat(c1[0], 1600, 1200, c1[1]), rs.cam_flat(c2[0], 1600, 1200, c2[1])) if kind == 3 else (None, None)
    gh._srand(1)
    m, st, mask = rs.estimate_classic(kind, p["x1"], p["x2"], ro, bo, cam1, cam2, pp=(float(row[12]), float(row[13])))
    full = np.zeros(12); m = np.asarray(m, float).reshape(-1); full[: len(m)] = m
    mk = np.zeros(1500, dtype=np.uint8); mk[: len(mask)] = mask
    return name, j, full, (int(st[0]), int(st[1]), int(st[2])), np.packbits(mk)


def gen(cases, seed, out):
    t = case_table(seed, cases)
    jobs = [(name, j, t[j]) for name in KINDS for j in range(cases)]
    with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
        rows = pool.map(_ref, jobs, chunksize=2)
    d = {"seed": seed, "cases": cases}
    for name in KINDS:
        rs_ = sorted((r for r in rows if r[0] == name), key=lambda r: r[1])
        d[f"{name}_model"] = np.array([r[2] for r in rs_]); d[f"{name}_istats"] = np.array([r[3] for r in rs_], dtype=np.int64); d[f"{name}_mask"] = np.array([r[4] for r in rs_])
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out), "bytes")


def run(cases, seed, ref):
    import mdrp_amd.poselib as poselib
    from mdrp_amd import _capi
    g = np.load(ref)
    assert int(g["seed"]) == seed and int(g["cases"]) >= cases
    t = case_table(seed, int(g["cases"]))
    loss_name = {v: k for k, v in _capi.LOSS_TYPES.items()}
    t0 = time.time()
    for name, kind in KINDS.items():
        same = lo = 0
        bad = []
        for j in range(cases):
            row = t[j]
            n = int(row[0])
            p = classic_options_pair(name, j, row)
            ro = {"max_iterations": int(row[5]), "min_iterations": int(row[6]), "max_epipolar_error": float(row[3]), "seed": int(row[4])}
            bo = {"max_iterations": int(row[9]), "loss_type": loss_name[int(row[7])], "loss_scale": float(row[8]), "gradient_tol": 1e-10}
            if kind == 3:
                c1, c2 = classic_options_cameras(row)
                cams = [{"model": "PINHOLE" if c[0] else "SIMPLE_PINHOLE", "width": 1600, "height": 1200, "params": c[1]} for c in (c1, c2)]
                pose, info = poselib.estimate_relative_pose(p["x1"], p["x2"], cams[0], cams[1], ro, bo)
                m = np.r_[pose.q, pose.t]
            elif kind == 4:
                pair, info = poselib.estimate_shared_focal_relative_pose(p["x1"], p["x2"], (float(row[12]), float(row[13])), ro, bo)
                m = np.r_[pair.pose.q, pair.pose.t, pair.camera1.params[0]]
            else:
                F, info = poselib.estimate_fundamental(p["x1"], p["x2"], ro, bo)
                m = np.asarray(F).reshape(-1)
            r, ist = g[f"{name}_model"][j], g[f"{name}_istats"][j]
            d = fund_diff(m, r[:9]) if kind == 5 else pose_diff(m[:7], r[:7])
            ok = ((info["iterations"], info["num_inliers"]) == (int(ist[1]), int(ist[2])) and np.array_equal(np.array(info["inliers"], dtype=np.uint8), np.unpackbits(g[f"{name}_mask"][j])[:n])
                  and d < 1e-6 and (kind != 4 or abs(m[7] - r[7]) < 1e-6 * abs(r[7])))
            same += ok; lo += info["refinements"] != int(ist[0])
            if not ok:
                bad.append(j)
        print(f"{name}: {same} / {cases} cases identical to the REFERENCE BINARY (iterations, inliers, mask, model 1e-6); LO count differs on {lo}; not identical: {bad}", flush=True)
    print(f"HIP path {time.time() - t0:.0f} s ({len(KINDS) * cases} calls of one pair)")


if __name__ == "__main__":
    mode, cases, seed, path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    (gen if mode == "gen" else run)(cases, seed, path)
