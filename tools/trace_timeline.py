#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV of a bench.py run: kernel concurrency histogram (fraction of time with
0/1/2/3/4... kernels in flight, mean concurrency), per-hardware-queue load and per-kernel average durations, over the
steady part of the timed region (the window between the 25th and the 75th percentile of the PyrLK launches' start
times -- set-up, warm-up and the single-stream pass after the timed region fall outside it).
usage: trace_timeline.py <..._kernel_trace.csv>"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("vsa::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Queue_Id"]))
rows.sort()
lk = [s for s, _, nm, _ in rows if nm.startswith("k_pyrlk")]
t0, t1 = lk[len(lk) // 4], lk[3 * len(lk) // 4]
win = [(max(s, t0), min(e, t1), nm, q, e - s) for s, e, nm, q in rows if e > t0 and s < t1]
ev = sorted([(s, 1) for s, *_ in win] + [(e, -1) for _, e, *_ in win])
conc, cur, last = collections.Counter(), 0, t0
for t, d in ev:
    conc[cur] += t - last
    last, cur = t, cur + d
conc[cur] += t1 - last
span = t1 - t0
print("window %.2f ms of steady state, %d kernels" % (span / 1e6, len(win)))
print("kernels in flight:", "  ".join("%d: %.1f%%" % (k, 100 * conc[k] / span) for k in sorted(conc)))
print("mean concurrency %.2f" % (sum(k * v for k, v in conc.items()) / span))
q = collections.defaultdict(int)
for s, e, nm, qq, _ in win:
    q[qq] += e - s
for qq, v in sorted(q.items()):
    print("hardware queue %s: busy %.1f%% of the window" % (qq, 100 * v / span))
d = collections.defaultdict(list)
for s, e, nm, _, dur in win:
    d[nm].append(dur)
print("%-34s %7s %10s %16s" % ("kernel", "calls", "avg us", "sum / window"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%-34s %7d %10.1f %15.1f%%" % (k, len(v), sum(v) / len(v) / 1e3, 100 * sum(v) / span))
per_frame = sum(sum(v) for k, v in d.items()) / max(len(d.get("k_pyrlk<7, 21, true>", [])) or len(lk) // 2, 1)
print("sum of a frame's kernel durations under this load: %.1f us" % (per_frame / 1e3))
