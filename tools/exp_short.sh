#!/bin/bash
# why a 20-step run reports less than a 1000-step run: the same 20 timed steps after 5, 50 and 300 warm-up steps, and 100 / 1000 timed steps
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 120 python bench.py --no-cpu-baseline "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]))'; }
for rep in 1 2; do
  echo "rep $rep steps 20 warmup 5: $(one --steps 20 --warmup 5)"
  echo "rep $rep steps 20 warmup 50: $(one --steps 20 --warmup 50)"
  echo "rep $rep steps 20 warmup 300: $(one --steps 20 --warmup 300)"
  echo "rep $rep steps 100 warmup 5: $(one --steps 100 --warmup 5)"
  echo "rep $rep steps 1000 warmup 50: $(one --steps 1000 --warmup 50)"
done
echo SHORT_OK
