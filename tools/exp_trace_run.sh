#!/bin/bash
# where a short run's time goes: host completion time of every frame (VSA_DRIVER_TRACE, tools/trace_run.py), polling waits against
# blocking ones (VISUALSLAM_AMD_BLOCKING_SYNC=1)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_trace
one() { timeout -k 10 150 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]))'; }
for B in 0 1; do
  export VISUALSLAM_AMD_BLOCKING_SYNC=$B
  echo "== VISUALSLAM_AMD_BLOCKING_SYNC=$B"
  echo "short runs: $(one --steps 20 --warmup 5) $(one --steps 20 --warmup 5) $(one --steps 20 --warmup 5) | 1000 steps: $(one) $(one) | one stream: $(one --streams 1)"
  VSA_DRIVER_TRACE=$GRAFT_REPO_ROOT/gpurun_out/r02_trace/b$B.txt one --steps 20 --warmup 5
  python tools/trace_run.py gpurun_out/r02_trace/b$B.txt.32_streams_20_frames | cut -c1-400
done
