#!/usr/bin/env python3
"""GPU box.  One case of tests/tools/stress_options.py in detail:  python tests/tools/diag_options_case.py NAME SEED CASES J [J ...]
prints options, the oracle's and the HIP path's statistics, the model difference and the number of differing mask bits."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")  # tests/tools -> repository root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
from helpers import OPTIONS_KINDS, options_cameras, options_dicts, options_pair, model_diff  # noqa: E402
import probe_options_campaign as poc  # noqa: E402
from oracle import pyorc as po  # noqa: E402


def main():
    name, seed, cases = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    t = poc.table(seed, cases)
    from mdrp_amd import _capi as capi
    h = capi.Handle(0)
    for j in [int(a) for a in sys.argv[4:]]:
        row = t[j]
        kind, es, rf = OPTIONS_KINDS[name]
        p = options_pair(name, j + 5000, row)
        rod, bod = options_dicts(row, es)
        c1, c2 = options_cameras(row)
        co = (po.cam_flat(*c1), po.cam_flat(*c2)) if kind == 0 else (None, None)
        m, st, mk = po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], po.ransac_opt(**rod), po.bundle_opt(**bod), *co)
        cams = []
        for c in (c1, c2):
            r = np.zeros(1, dtype=capi.CAMERA_DTYPE); r["model_id"] = c[0]; r["params"][0, :len(c[1])] = c[1]; cams.append(r)
        ro = capi.ransac_opt_from_dict({("monodepth_" + k if k in ("estimate_shift", "weight_sampson") else k): v for k, v in rod.items()})
        res, mask = h.estimate_batch(kind, p["x1"][None], p["x2"][None], p["d1"][None], p["d2"][None], ro, capi.bundle_opt_from_dict(bod), None,
                                     cams[0] if kind == 0 else None, cams[1] if kind == 0 else None)
        r = res[0]
        gm = capi.model_to_array(r["model"])
        print(f"case {j}: N = {len(p['x1'])}  ransac {rod}  bundle {bod}  cameras {c1} {c2}")
        print(f"  oracle: refinements {st.refinements} iterations {st.iterations} inliers {st.num_inliers} score {st.model_score!r} model {np.asarray(m)}")
        print(f"  hip   : refinements {int(r['refinements'])} iterations {int(r['iterations'])} inliers {int(r['num_inliers'])} score {float(r['model_score'])!r} model {gm}")
        print(f"  model diff {model_diff(gm, np.asarray(m)):.3e}; mask bits differing {int((mask[0].astype(bool) != mk.astype(bool)).sum())}; first_chunk {h.last_stats()['first_chunk']}")
    h.close()


if __name__ == "__main__":
    main()
