#!/bin/bash
# the driver-sized block (20 steps after 5 warm-up steps, pre-heated) by camera streams per GPU: first block and the spread of 15 more
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
one() { timeout -k 10 150 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["repeat_frames_per_s"]; print(round(d["value"]), "(repeats: median", round(r["median"]), "min", round(r["min"]), "max", str(round(r["max"]))+")")'; }
for S in 6 8 12 16 32; do
  for i in 1 2 3; do echo "streams $S short: $(one --steps 20 --warmup 5 --streams $S)"; done
  echo "streams $S long : $(one --streams $S)"
done
