#!/usr/bin/env python3
"""Kernels AND memory copies of the last step in a rocprofv3 rocpd database, in start order: does the H2D overlap the kernels?
usage: host_path_timeline.py results.db [rows = 80]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
rows = [(s, e, name.split("(")[0].replace("void mdrp::", "")[:44]) for name, s, e in c.execute("select name, start, end from kernels")]
mc = [t for t in tabs if "memory_cop" in t.lower() or "memcpy" in t.lower()]
for t in mc[:1]:
    cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
    sz = "size" if "size" in cols else ("bytes" if "bytes" in cols else None)
    nm = "name" if "name" in cols else None
    q = f"select start, end, {nm or chr(39) + 'copy' + chr(39)}, {sz or 0} from {t}"
    for s, e, name, b in c.execute(q):
        rows.append((s, e, f"  COPY {name} {b / 1e6:.1f} MB"))
rows.sort()
rows = rows[-n:]
t0 = rows[0][0]
for s, e, name in rows:
    print(f"{name:52s} {(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f}  ({(e - s) / 1e6:7.3f} ms)")
if not mc:
    print("(no memory-copy table in this database: tables =", tabs, ")")
