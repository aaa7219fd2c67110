#!/usr/bin/env python3
"""Merge the rocprofv3 --pmc passes of one workload into profiles/rNN_pmc_<workload>.json (read by bench.py).

usage: pmc_json.py WORKLOAD PAIRS_PER_GPU STEPS_PROFILED pass1.csv [pass2.csv ...] > profiles/r02_pmc_WORKLOAD.json

Per kernel, means per launch over the profiled steps:
  hbm_bytes_corrected = 2 x FETCH_SIZE + WRITE_SIZE   (KiB -> bytes; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
                        coalesced reads, MI355X_MICROARCH.md "HBM"; WRITE_SIZE is exact)
  mfma_busy_frac      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)    cycles the matrix pipes were busy / SIMD-cycles of the
                        kernel (GRBM_GUI_ACTIVE is summed over the 8 XCDs, so GRBM / 8 = the kernel's cycles at the clock it ran at;
                        1024 SIMDs).  <= 1 by construction; numerator and denominator come from the SAME pass.
  valu_active_per_simd_cycle = 4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024)    SQ_ACTIVE_INST_* count, per WAVEFRONT, the
                        quad-cycles it spends in VALU instructions; two wavefronts' 2-cycle fp32 instructions overlap on one SIMD, so
                        this is an occupancy-weighted activity, NOT an issue-port fraction: it exceeds 1 in fp32-heavy kernels (k_count)
                        and equals the issue fraction only where every instruction holds the port for its 4 cycles (fp64: k_lo, k_final)
  cycles_per_valu     = 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mdrp_amd import build as _build  # noqa: E402  (source_hash(): the sources the profiled library was built from)

workload, pairs, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
per_pass = []
for path in sys.argv[4:]:
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[name][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    per_pass.append(acc)

kernels = {}
names = sorted({k for acc in per_pass for k in acc if k.startswith("mdrp::")})
for name in names:
    out = {}
    for acc in per_pass:
        c = acc.get(name)
        if not c:
            continue
        launches = max(v[0] for v in c.values())
        mean = {k: v[1] / v[0] for k, v in c.items()}
        out["launches_per_step"] = launches / steps
        for k, v in mean.items():
            out[k] = v
        if "GRBM_GUI_ACTIVE" in mean and mean["GRBM_GUI_ACTIVE"] > 0:
            simd_cycles = mean["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
            if "SQ_ACTIVE_INST_VALU" in mean:
                out["valu_active_per_simd_cycle"] = 4.0 * mean["SQ_ACTIVE_INST_VALU"] / simd_cycles
            if "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
                out["mfma_busy_frac"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
            if "SQ_WAVE_CYCLES" in mean:
                out["mean_waves_per_simd"] = 4.0 * mean["SQ_WAVE_CYCLES"] / simd_cycles
        if "SQ_INSTS_VALU" in mean and "SQ_ACTIVE_INST_VALU" in mean and mean["SQ_INSTS_VALU"] > 0:
            out["cycles_per_valu"] = 4.0 * mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_INSTS_VALU"]
    if "FETCH_SIZE" in out or "WRITE_SIZE" in out:
        out["hbm_bytes_corrected"] = (2.0 * out.get("FETCH_SIZE", 0.0) + out.get("WRITE_SIZE", 0.0)) * 1024.0
    kernels[name] = out
json.dump({"command": "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --host-steps 0 "
                      "(one pass per counter set; tools/profile_round.sh)",
           # bench.py attaches these numbers to its line only while the library it runs carries the same hash (mdrp_version())
           "source_hash": _build.built_hash() or _build.source_hash(),
           "workload": workload, "pairs_per_gpu": pairs, "unit": "mean per launch over the launches of the profiled steps",
           "kernels": kernels}, sys.stdout, indent=1)
print()
