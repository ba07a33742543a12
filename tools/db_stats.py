#!/usr/bin/env python3
"""Per-kernel average durations from rocprofv3's sqlite output (--kernel-trace).   usage: db_stats.py results.db ..."""
import sqlite3
import sys

for f in sys.argv[1:]:
    db = sqlite3.connect(f)
    t = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if "kernel_dispatch" in x][0]
    ks = [x for x in t if "info_kernel_symbol" in x][0]
    q = (f"select s.kernel_name, count(*), avg(d.end - d.start), min(d.end - d.start) from {kd} d "
         f"join {ks} s on d.kernel_id = s.id group by 1 order by 3 desc")
    print(f)
    for name, n, avg, mn in db.execute(q):
        print(f"  avg {avg / 1000:9.2f} us  min {mn / 1000:9.2f} us  x{n:4d}  {name[:100]}")
