#!/bin/bash
# more camera streams per GPU: a step is one frame on every stream, so the driver's 20-step run is longer and its start-up weighs less
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]))'; }
for s in 32 48 64 96 128; do
  echo "streams $s short: $(one --streams $s --steps 20 --warmup 5) $(one --streams $s --steps 20 --warmup 5) $(one --streams $s --steps 20 --warmup 5) | 300 steps: $(one --streams $s --steps 300)"
done
