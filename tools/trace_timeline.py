#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV: per-queue load, kernel concurrency histogram, per-kernel averages."""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("::")[-1][:24], r["Queue_Id"]))
rows.sort()
rows = rows[len(rows) // 2:]          # steady state
t0, t1 = rows[0][0], max(e for _, e, *_ in rows)
ev = sorted([(s, 1) for s, *_ in rows] + [(e, -1) for _, e, *_ in rows])
conc, cur, last = collections.Counter(), 0, ev[0][0]
for t, d in ev:
    conc[cur] += t - last
    last, cur = t, cur + d
print("window %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(rows)))
print("concurrency:", "  ".join("%d: %.1f%%" % (k, 100 * conc[k] / (t1 - t0)) for k in sorted(conc)))
q = collections.defaultdict(list)
for s, e, nm, qq in rows:
    q[qq].append(e - s)
for qq, v in sorted(q.items()):
    print("queue %s: %d kernels, busy %.1f%%" % (qq, len(v), 100 * sum(v) / (t1 - t0)))
d = collections.defaultdict(list)
for s, e, nm, _ in rows:
    d[nm].append(e - s)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%-26s n=%5d avg %.1f us  total %.1f%%" % (k, len(v), sum(v) / len(v) / 1e3, 100 * sum(v) / (t1 - t0)))
