#!/usr/bin/env python3
"""profiles/frame_counters.json from the PMC summary of tools/gpu_profile2.sh (one camera stream's frames through the OpenVX graph,
tools/prof_kernels.py graph): per kernel the vector / scalar / LDS wave-instructions and the HBM bytes of one launch, the frame's
totals, stamped with the commit the library was built from.  bench.py reads it for roofline.traffic, roofline_valu and valu_issue.
HBM bytes: FETCH_SIZE / WRITE_SIZE are in KiB (x 1024).  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced streaming read (double it), WRITE_SIZE is exact, other access patterns must be calibrated.  Calibration on
this GPU (tools/calib_fetch.hip, profiles/r01_fetch_calibration.txt): coalesced streams of any lane width x2; PyrLK's tile gathers
(rows of 8-9 dwords at random positions) are counted as whole 128-byte lines, x1.  So hbm_bytes = FETCH x1 + WRITE for the tile-gather
kernels (k_pyrlk*, also with the Harris cell pass fused in, and the merge) and FETCH x2 + WRITE for the streaming ones (colour convert,
pyramid, the Harris cell pass alone); both readings are kept.
The stamp is the commit the kernels were built from: the script refuses to write one while visualslam_amd/csrc has uncommitted changes, and
`--check <frame_counters.json>` fails when the kernel sources have changed between the stamped commit and HEAD (profiles older than the code).
usage: make_frame_counters.py <pmc_summary.txt> <out.json>      |      make_frame_counters.py --check <frame_counters.json>"""
import collections
import json
import re
import subprocess
import sys

if len(sys.argv) == 3 and sys.argv[1] == "--check":
    stamp = json.load(open(sys.argv[2]))["commit"]
    if "+" in stamp or subprocess.run(["git", "diff", "--quiet", stamp, "HEAD", "--", "visualslam_amd/csrc"]).returncode != 0:
        sys.exit(f"profiles stamped {stamp} do not describe the kernels at HEAD: regenerate them (tools/gpu_profile_r3*.sh)")
    if subprocess.run(["git", "status", "--porcelain", "visualslam_amd/csrc"], capture_output=True, text=True).stdout.strip():
        sys.exit("visualslam_amd/csrc has uncommitted changes: the committed profiles do not describe this tree")
    print(f"profiles stamped {stamp}: kernel sources unchanged since")
    sys.exit(0)

rows = collections.defaultdict(dict)
calls = collections.defaultdict(int)
for ln in open(sys.argv[1]):
    m = re.match(r"(\S.*?)\s{2,}(\S+)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)", ln)
    if m:
        rows[m.group(1).strip()][m.group(2)] = float(m.group(4))
        calls[m.group(1).strip()] = max(calls[m.group(1).strip()], int(m.group(3)))
per_frame = {k for k, n in calls.items() if 2 * n >= max(calls.values())}    # the others ran once, in the tracker's first-frame set-up
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
if subprocess.run(["git", "status", "--porcelain", "visualslam_amd/csrc"], capture_output=True, text=True).stdout.strip():
    sys.exit("visualslam_amd/csrc has uncommitted changes: commit the kernels the profile was measured on before stamping it")
out = {"commit": commit,
       "source": "tools/gpu_profile_r3a.sh: rocprofv3 --pmc passes over tools/prof_kernels.py graph (one camera stream, 1080p / 5 levels / 4096 points / 21x21)",
       "peak_wave_instructions_per_ns": 989.0,
       "peak_note": "f32 FMA issue rate measured on this GPU with all SIMDs busy (profiles/r01_valu_issue_rates.txt): one wave-instruction per 1.035 ns per SIMD x 1024 SIMDs",
       "by_kernel": {}}
total = 0.0
for k, c in sorted(rows.items()):
    e = {"valu_wave_instructions": c.get("SQ_INSTS_VALU"), "salu_wave_instructions": c.get("SQ_INSTS_SALU"),
         "lds_wave_instructions": c.get("SQ_INSTS_LDS"), "waves": c.get("SQ_WAVES"),
         "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": c.get("SQ_LDS_IDX_ACTIVE")}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        gather = k.startswith("k_pyrlk") or k.startswith("k_merge")
        e["fetch_correction"] = "x1 (tile gathers: whole 128-B lines counted)" if gather else "x2 (coalesced stream)"
        e["hbm_bytes"] = (c["FETCH_SIZE"] * (1.0 if gather else 2.0) + c["WRITE_SIZE"]) * 1024.0
        e["hbm_bytes_fetch_x1"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        e["hbm_bytes_fetch_x2"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        e["fetch_size_kib"], e["write_size_kib"] = c["FETCH_SIZE"], c["WRITE_SIZE"]
    e["launches_in_the_profile"] = calls[k]
    e["per_frame"] = k in per_frame
    out["by_kernel"][k] = e
    if k in per_frame:
        total += c.get("SQ_INSTS_VALU") or 0.0
out["valu_wave_instructions_per_frame"] = total
lk = [k for k in out["by_kernel"] if k.startswith("k_pyrlk")]
if lk:
    out["by_kernel"]["k_pyrlk"] = dict(out["by_kernel"][lk[0]], kernel=lk[0])
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
