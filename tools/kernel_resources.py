#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin) as one line per kernel."""
import re
import subprocess
import sys

cur = None
rows = {}
for ln in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(.+?): (\S+) \[", ln)
    if m and cur:
        rows[cur][m.group(1)] = m.group(2)
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for k, v in rows.items():
    if pat in k:
        print(f"{k[:70]:70s} VGPR {v.get('VGPRs'):>4} AGPR {v.get('AGPRs'):>3} spill {v.get('VGPRs Spill'):>3} scratch {v.get('ScratchSize [bytes/lane]'):>5} "
              f"occ {v.get('Occupancy [waves/SIMD]')} LDS {v.get('LDS Size [bytes/block]')} SGPR {v.get('TotalSGPRs')}")
