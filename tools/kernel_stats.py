#!/usr/bin/env python3
"""rocprofv3 *_kernel_stats.csv -> a plain table (kernel names contain commas; cut(1) garbles them).
usage: kernel_stats.py <stats.csv> [out.txt]"""
import csv
import sys

lines = ["%-40s %8s %12s %10s %10s %8s" % ("kernel", "calls", "avg ns", "min ns", "max ns", "% time")]
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].replace("vsa::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    lines.append("%-40s %8s %12.0f %10s %10s %8s" % (name[:40], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
text = "\n".join(lines) + "\n"
sys.stdout.write(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
