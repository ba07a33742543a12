#!/bin/bash
# does the slow start of a run (frames completed per ms) depend on how long the GPU has been busy before it?
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_trace
for W in 5 50 200; do
  VSA_DRIVER_TRACE=$GRAFT_REPO_ROOT/gpurun_out/r02_trace/w$W.txt timeout -k 10 150 python bench.py --no-cpu-baseline --steps 60 --warmup $W 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("warm-up '$W', 60 steps:", round(d["value"]))'
  python tools/trace_run.py gpurun_out/r02_trace/w$W.txt.32_streams_60_frames | grep "per ms"
done
