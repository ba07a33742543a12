#!/usr/bin/env python3
"""Static resources of every kernel in the library, from the compiler's metadata (hipcc -S of each .hip with the
library's flags): VGPRs, SGPRs, private (scratch) bytes per lane, static LDS, max workgroup size.
usage: tools/kernel_resources.py [out.txt]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "visualslam_amd", "csrc")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize".split()

rows = []
with tempfile.TemporaryDirectory() as tmp:
    for f in sorted(os.listdir(SRC)):
        if not f.endswith(".hip"):
            continue
        out = os.path.join(tmp, f + ".s")
        subprocess.run(["hipcc", *FLAGS, f"-I{ROOT}/include", f"-I{SRC}", "-S", "--cuda-device-only", "-o", out, os.path.join(SRC, f)],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
        for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
            blk = m.group(0)
            g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)     # noqa: E731
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            name = name.replace("vsa::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            rows.append((f, name, int(g("vgpr_count")), int(g("sgpr_count")), int(g("private_segment_fixed_size")),
                         int(g("group_segment_fixed_size")), int(g("max_flat_workgroup_size"))))
lines = ["%-18s %-34s %5s %5s %8s %10s %6s" % ("file", "kernel", "VGPR", "SGPR", "scratch", "static LDS", "max WG")]
for r in rows:
    lines.append("%-18s %-34s %5d %5d %8d %10d %6d" % r)
text = "\n".join(lines) + "\n"
sys.stdout.write(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text)
