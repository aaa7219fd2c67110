#!/usr/bin/env python3
"""Timeline of the last bench step in a rocprofv3 rocpd database: start/end (ms, relative) of every kernel launch.
usage: rocpd_timeline.py results.db [n_last_kernels]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = list(c.execute("select name, start, end from kernels order by start"))[-n:]
t0 = rows[0][1]
for name, s, e in rows:
    short = name.split("(")[0].replace("void mdrp::", "")[:40]
    print(f"{short:40s} {(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f}  ({(e - s) / 1e6:7.3f} ms)")
