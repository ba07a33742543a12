#!/bin/bash
# graph streams at normal priority against streams created with a priority (a hardware-queue pool of their own): long and short runs
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 120 python bench.py --no-cpu-baseline "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]))'; }
for rep in 1 2 3 4 5; do
  echo "rep $rep normal $(VISUALSLAM_AMD_STREAM_PRIORITY=0 one) | greatest (-1) $(VISUALSLAM_AMD_STREAM_PRIORITY=-1 one) | least (1) $(VISUALSLAM_AMD_STREAM_PRIORITY=1 one)"
done
echo OK
