#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as the per-kernel stats table we commit under profiles/.
usage: rocpd_summary.py results.db [> profiles/NAME.txt]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6 "
                      "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows) or 1.0
print(f"{'kernel':64s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>9s} {'min_ms':>9s} {'max_ms':>9s} {'pct':>6s}")
for r in rows:
    print(f"{r[0][:64]:64s} {r[1]:6d} {r[2]:10.3f} {r[3]:9.3f} {r[4]:9.3f} {r[5]:9.3f} {100 * r[2] / tot:6.2f}")
