#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]))'; }
for rep in 1 2; do
  echo "rep $rep none: $(one)"
  echo "rep $rep ROC_SIGNAL_POOL_SIZE=1024: $(ROC_SIGNAL_POOL_SIZE=1024 one)"
  echo "rep $rep HSA_ENABLE_INTERRUPT=0: $(HSA_ENABLE_INTERRUPT=0 one)"
  echo "rep $rep ring 8: $(one --ring 8)"
  echo "rep $rep streams 16: $(one --streams 16)"
  echo "rep $rep python driver: $(one --driver python)"
done
echo SHORT_OK
