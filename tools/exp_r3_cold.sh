#!/bin/bash
# the first timed block after a 5-step warm-up against a later block of the same run: frames completed per millisecond
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3cold
for S in 16 32; do
  VSA_DRIVER_TRACE=$GRAFT_REPO_ROOT/gpurun_out/r3cold/cold$S timeout -k 10 150 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --streams $S --repeats 0 > /dev/null 2>&1
  VSA_DRIVER_TRACE=$GRAFT_REPO_ROOT/gpurun_out/r3cold/warm$S timeout -k 10 150 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --streams $S --repeats 4 > /dev/null 2>&1
  echo "== $S streams, first block after 5 warm-up steps"; python tools/trace_run.py gpurun_out/r3cold/cold$S.${S}_streams_20_frames | cut -c1-300
  echo "== $S streams, fifth block"; python tools/trace_run.py gpurun_out/r3cold/warm$S.${S}_streams_20_frames | cut -c1-300
done
rocm-smi --showclocks 2>/dev/null | head -20 || true
