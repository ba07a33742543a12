#!/bin/bash
# usage: gpu_pmc.sh TAG WHAT REPS [COUNTERSET ...]   (a counterset is a quoted, space-separated --pmc list)
# Serial kernel timings (kernel-trace) of tools/prof_kernels.py WHAT REPS, then one PMC pass per counterset.
set -euo pipefail
TAG=$1; WHAT=$2; REPS=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k -- python3 "$GRAFT_REPO_ROOT/tools/prof_kernels.py" "$WHAT" "$REPS" > "$OUT/trace.log" 2>&1
cut -d, -f1-4,6,7 "$OUT/trace/k_kernel_stats.csv" | cut -c1-150
i=0
for CS in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d "$OUT/pmc$i" -o k -- python3 "$GRAFT_REPO_ROOT/tools/prof_kernels.py" "$WHAT" "$REPS" > "$OUT/pmc$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, collections, glob, sys
d = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/pmc*/k_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "vsa::" in r["Kernel_Name"]:
            nm = r["Kernel_Name"].split("vsa::(anonymous namespace)::")[1].split("(")[0]
            d[(nm, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(d.items()):
    print(f"{k:28s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):14.1f}")
PY
echo PMC_OK
