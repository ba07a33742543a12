#!/usr/bin/env python3
"""Throughput over time in a rocprofv3 kernel trace of a bench.py run: per 2-ms bin since the first fused PyrLK launch of the camera
streams, the frames completed (merge kernels ended), mean PyrLK duration, mean number of kernels in flight and the busy share of time.
usage: trace_ramp.py <..._kernel_trace.csv> [bin_ms]"""
import csv
import sys

binw = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2e6
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("vsa::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if name.startswith("k_"):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
lk = [r for r in rows if r[2].startswith("k_pyrlk_cells")]
t0 = lk[0][0]
nb = int((rows[-1][1] - t0) / binw) + 1
fr = [0] * nb; lkd = [[] for _ in range(nb)]; busy = [0.0] * nb
for s, e, nm in rows:
    if s < t0:
        continue
    b = int((e - t0) / binw)
    if nm.startswith("k_merge"):
        fr[b] += 1
    if nm.startswith("k_pyrlk"):
        lkd[b].append(e - s)
    for bb in range(int((s - t0) / binw), int((e - t0) / binw) + 1):
        lo, hi = max(s, t0 + bb * binw), min(e, t0 + (bb + 1) * binw)
        busy[bb] += max(hi - lo, 0)
for b in range(nb):
    print("t = %5.1f ms: %4d frames (%5.1f k/s), PyrLK %5.1f us x %3d, kernels in flight %.2f" %
          (b * binw / 1e6, fr[b], fr[b] / (binw / 1e9) / 1e3, (sum(lkd[b]) / max(len(lkd[b]), 1)) / 1e3, len(lkd[b]), busy[b] / binw))
