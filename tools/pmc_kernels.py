#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: pmc_kernels.py file.csv [file2.csv ...]
Prints, per kernel name and counter, the number of dispatches, the sum and the mean per dispatch."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].replace("void mdrp::", "").replace("mdrp::", "")[:36], r["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
for (name, ctr), (n, v) in sorted(acc.items()):
    if name.startswith(("k_", "kc_")):
        print(f"{name:36s} {ctr:28s} n={n:4d} sum={v:16.0f} mean={v / n:14.1f}")
