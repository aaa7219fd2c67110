#!/usr/bin/env python3
"""Build container only (needs the reference shim).  The reference binary's results for the randomised options table of probe_options_campaign.py:
    python3 tests/tools/gen_options_ref_table.py CASES SEED OUT.npz       (4 estimators x CASES cases, 8 workers; 2048 cases: ~2 min)
OUT.npz holds, per estimator, stats (refinements, iterations, inliers, inlier ratio, score), the model as the reference returns it, and the packed mask;
tests/tools/stress_options.py CASES SEED OUT.npz compares the HIP path with it on the GPU box.  (A one-off: the file is large and not a committed fixture;
put it under build/ — ignored by git, carried to the GPU box.)"""
import ctypes
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, HERE)
from helpers import OPTIONS_KINDS, OPTIONS_NAMES, options_cameras, options_dicts, options_pair  # noqa: E402
import probe_options_campaign as poc  # noqa: E402

libc = ctypes.CDLL("libc.so.6")


def work(a):
    name, j, row = a
    import refshim as rs
    kind, es, rf = OPTIONS_KINDS[name]
    p = options_pair(name, j + 5000, row)
    rod, bod = options_dicts(row, es)
    c1, c2 = options_cameras(row)
    cr = (rs.cam_flat(c1[0], 1600, 1200, c1[1]), rs.cam_flat(c2[0], 1600, 1200, c2[1])) if kind == 0 else (None, None)
    libc.srand(1)
    m, st, mk = rs.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], rs.ropt(**rod), rs.bopt(**bod), *cr)
    m = np.asarray(m, float)
    mask = np.zeros(2000, dtype=np.uint8); mask[:len(mk)] = mk
    return name, j, np.asarray(st, float)[:5], np.r_[m, np.full(12 - len(m), np.nan)], np.packbits(mask), len(m)


def main():
    cases, seed, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    t = poc.table(seed, cases)
    jobs = [(name, j, t[j]) for name in OPTIONS_NAMES for j in range(cases)]
    with mp.get_context("fork").Pool(8) as pool:
        res = pool.map(work, jobs, chunksize=8)
    d = {"seed": seed, "cases": cases}
    for name in OPTIONS_NAMES:
        rows = sorted([r for r in res if r[0] == name], key=lambda r: r[1])
        d[name + "_stats"] = np.stack([r[2] for r in rows]); d[name + "_model"] = np.stack([r[3] for r in rows]); d[name + "_mask"] = np.stack([r[4] for r in rows])
        d[name + "_model_len"] = np.array([r[5] for r in rows])
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
