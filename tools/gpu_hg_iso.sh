#!/bin/bash
set -euo pipefail
R=${1:-hg_iso}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o hg -- python3 "$GRAFT_REPO_ROOT/tools/prof_hg.py" 100 > "$OUT/run.log" 2> "$OUT/trace.err"
cat "$OUT/run.log"
find "$OUT" -name '*kernel_stats.csv' -exec head -9 {} \;
