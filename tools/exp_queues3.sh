#!/bin/bash
# GPU_MAX_HW_QUEUES sweep on the default bench with the round-2 kernels (frames/s, isolated fused-launch us); streams per GPU 32 and 48
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 120 python bench.py --no-cpu-baseline "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["launch_ms"]*1000,1), round(d["roofline"]["launch_ms_in_timed_region"]*1000,1))'; }
for q in 4 5 6 8; do
  echo "queues $q, 32 streams: $(GPU_MAX_HW_QUEUES=$q one)"
done
for q in 4 6 8; do
  echo "queues $q, 48 streams: $(GPU_MAX_HW_QUEUES=$q one --streams 48)"
done
echo "queues 4, 64 streams: $(GPU_MAX_HW_QUEUES=4 one --streams 64)"
echo QUEUES_OK
