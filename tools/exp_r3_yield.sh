#!/bin/bash
# waiting by hipStreamSynchronize against polling with sched_yield between polls: short (driver-sized) and long runs, 32 and 16 streams
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3yield
one() { timeout -k 10 150 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["repeat_frames_per_s"]; print(round(d["value"]), "(repeats: median", round(r["median"]), "min", round(r["min"]), "max", str(round(r["max"]))+")")'; }
for Y in 0 1 0 1; do
  export VISUALSLAM_AMD_YIELD_SYNC=$Y
  echo "== YIELD_SYNC=$Y"
  echo "  32 streams short: $(one --steps 20 --warmup 5)"
  echo "  32 streams long : $(one)"
  echo "  16 streams short: $(one --steps 20 --warmup 5 --streams 16)"
  echo "  64 streams short: $(one --steps 20 --warmup 5 --streams 64)"
done
VISUALSLAM_AMD_YIELD_SYNC=1 VSA_DRIVER_TRACE=$GRAFT_REPO_ROOT/gpurun_out/r3yield/y32 timeout -k 10 150 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 4 > /dev/null 2>&1
python tools/trace_run.py gpurun_out/r3yield/y32.32_streams_20_frames | cut -c1-300
