#!/usr/bin/env python3
"""How fast is the CPU port (oracle/*.c — what bench.py's `cpu_baseline` times on the GPU box) against the reference's own
PoseLib binary on the same core and the same inputs?  The binary cannot travel to the GPU box, so the ratio is measured here,
in the build container, and committed as profiles/r04_port_vs_reference.json; bench.py attaches it to every `cpu_baseline`
(`port_vs_reference_binary`) so that a speed-up quoted against the port can be read against the reference itself.

Single thread, alternating reference / port per pair (same cache and clock conditions), the first pair of each untimed.

    python3 tests/tools/port_vs_reference.py [pairs per workload, default 12]
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refshim as rs  # noqa: E402
from mdrp_amd import synth  # noqa: E402
from oracle import pyorc as po  # noqa: E402

WORKLOADS = {
    # bench.py workload: kind, estimate_shift, n, outlier_frac, random_focal
    "calib_p3p_n2000_i10k": (0, False, 2000, 0.5, None),
    "calib_shift_n2000_i10k": (0, True, 2000, 0.5, None),
    "shared_n2000_i10k": (1, False, 2000, 0.5, "shared"),
    "varying_n5000_i10k": (2, True, 5000, 0.5, "varying"),
}


def cpu_model():
    for ln in open("/proc/cpuinfo"):
        if ln.startswith("model name"):
            return ln.split(":", 1)[1].strip()
    return "unknown"


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    cam_r, cam_o = rs.cam_flat(0, 1600, 1200, [800.0, 0.0, 0.0]), po.cam_flat(0, [800.0, 0.0, 0.0])
    out = {"cpu_model": cpu_model(), "threads": 1, "pairs_per_workload": pairs,
           "note": "pairs/s on one core of the build container; port = oracle/*.c (gcc -O2), reference = PoseLib 2.0.5 binary of the "
                   "wheel under /root/reference/demo, called through oracle/_ref/librefshim.so; port_over_reference < 1 means the "
                   "port is slower, i.e. a speed-up quoted against the port overstates the one against the reference by 1 / ratio",
           "workloads": {}}
    for name, (kind, es, n, of, rf) in WORKLOADS.items():
        kw = dict(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, max_reproj_error=16.0, seed=0, estimate_shift=es)
        t_ref = t_port = 0.0
        for i in range(pairs + 1):
            p = synth.make_pair(i, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf,
                                shift1=0.2 if es and kind == 0 else 0.0, shift2=-0.1 if es and kind == 0 else 0.0)
            t0 = time.perf_counter()
            rs.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], rs.ropt(**kw), rs.bopt(loss_type=4), cam_r if kind == 0 else None, cam_r if kind == 0 else None)
            t1 = time.perf_counter()
            po.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], po.ransac_opt(**kw), po.bundle_opt(loss_type=4), cam_o if kind == 0 else None, cam_o if kind == 0 else None)
            t2 = time.perf_counter()
            if i > 0:
                t_ref += t1 - t0; t_port += t2 - t1
        out["workloads"][name] = {"reference_pairs_per_s": pairs / t_ref, "port_pairs_per_s": pairs / t_port,
                                  "port_over_reference": t_ref / t_port, "pairs": pairs}
        print(name, out["workloads"][name], flush=True)
    with open(os.path.join(ROOT, "profiles", "r04_port_vs_reference.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
