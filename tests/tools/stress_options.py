#!/usr/bin/env python3
"""GPU box.  The large randomised options table of tests/tools/probe_options_campaign.py (same generator, same seed -> the cases that script ran against the
reference binary in the build container) through the HIP path, one call per case, against the CPU oracle run beside it (a process pool on the host cores):
    python tests/tools/stress_options.py [CASES [SEED]]  > profiles/rNN_stress_options.txt
Counts the cases on which iterations, inlier count, mask and model (1e-6, or NaN in the same places) agree, and the LO-count differences."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, HERE)
from helpers import OPTIONS_KINDS, OPTIONS_NAMES, options_cameras, options_dicts, options_pair, same_model  # noqa: E402
import probe_options_campaign as poc  # noqa: E402


def oracle_case(a):
    name, j, row = a
    from oracle import pyorc as po
    kind, es, rf = OPTIONS_KINDS[name]
    p = options_pair(name, j + 5000, row)
    rod, bod = options_dicts(row, es)
    c1, c2 = options_cameras(row)
    co = (po.cam_flat(*c1), po.cam_flat(*c2)) if kind == 0 else (None, None)
    m, st, mk = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], po.ransac_opt(**rod), po.bundle_opt(**bod), *co)
    return name, j, np.asarray(m), (st.refinements, st.iterations, st.num_inliers), np.packbits(mk)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
    ref_file = sys.argv[3] if len(sys.argv) > 3 else None   # tests/tools/gen_options_ref_table.py: the REFERENCE BINARY's results instead of the oracle's
    t = poc.table(seed, cases)
    jobs = [(name, j, t[j]) for name in OPTIONS_NAMES for j in range(cases)]
    t0 = time.time()
    if ref_file:
        g = np.load(ref_file)
        assert int(g["seed"]) == seed and int(g["cases"]) >= cases, "reference table of another seed / size"
        orc = {}
        for name in OPTIONS_NAMES:
            kind = OPTIONS_KINDS[name][0]
            for j in range(cases):
                st = g[name + "_stats"][j]
                m = g[name + "_model"][j][:int(g[name + "_model_len"][j])]
                n = int(t[j][0])
                orc[(name, j)] = (name, j, np.r_[m, 1.0, 1.0] if kind == 0 else m, (int(st[0]), int(st[1]), int(st[2])), np.packbits(np.unpackbits(g[name + "_mask"][j])[:n]))
    else:
        with mp.get_context("fork").Pool(min(32, os.cpu_count() or 1)) as pool:   # before the GPU is touched
            orc = {(r[0], r[1]): r for r in pool.map(oracle_case, jobs, chunksize=8)}
    t_orc = time.time() - t0
    against = "the REFERENCE BINARY" if ref_file else "the oracle"
    from mdrp_amd import _capi as capi
    h = capi.default_handle(0)
    t0 = time.time()
    for name in OPTIONS_NAMES:
        kind, es, rf = OPTIONS_KINDS[name]
        same = lo = 0
        bad = []
        for j in range(cases):
            row = t[j]
            p = options_pair(name, j + 5000, row)
            rod, bod = options_dicts(row, es)
            c1, c2 = options_cameras(row)
            cams = []
            for c in (c1, c2):
                r = np.zeros(1, dtype=capi.CAMERA_DTYPE); r["model_id"] = c[0]; r["params"][0, :len(c[1])] = c[1]; cams.append(r)
            ro = capi.ransac_opt_from_dict({("monodepth_" + k if k in ("estimate_shift", "weight_sampson") else k): v for k, v in rod.items()})
            res, mask = h.estimate_batch(kind, p["x1"][None], p["x2"][None], p["d1"][None], p["d2"][None], ro, capi.bundle_opt_from_dict(bod), None,
                                         cams[0] if kind == 0 else None, cams[1] if kind == 0 else None)
            r = res[0]
            _, _, m, st, mk = orc[(name, j)]
            ok = (int(r["iterations"]), int(r["num_inliers"])) == st[1:] and np.array_equal(np.packbits(mask[0]), mk) and same_model(capi.model_to_array(r["model"]), m)
            same += ok; lo += int(r["refinements"]) != st[0]
            if not ok:
                bad.append(j)
        print(f"{name}: {same} / {cases} cases identical to {against} (iterations, inliers, mask, model 1e-6); LO count differs on {lo}; not identical: {bad}", flush=True)
    print(f"{'reference table loaded in' if ref_file else 'oracle'} {t_orc:.0f} s on the host cores, HIP path {time.time() - t0:.0f} s ({4 * cases} calls of one pair)")


if __name__ == "__main__":
    main()
