#!/bin/bash
# round 3, pass D: packed FAST pass -- whole GPU suite, then configs[2] under rocprofv3 with the concurrency timeline
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3" -o c3 -- python3 "$GRAFT_REPO_ROOT/bench.py" --config c3 --steps 100 --warmup 10 --no-cpu-baseline --repeats 0 > "$OUT/c3_under_rocprof.json" 2> "$OUT/c3.err"
cut -c1-200 "$OUT/c3_under_rocprof.json"
python3 "$GRAFT_REPO_ROOT/tools/trace_timeline.py" "$(find "$OUT/c3" -name 'c3_kernel_trace.csv' | head -1)" > "$OUT/c3_concurrency.txt"
cat "$OUT/c3_concurrency.txt"
rm -f "$(find "$OUT/c3" -name 'c3_kernel_trace.csv' | head -1)"
cut -c1-160 "$(find "$OUT/c3" -name 'c3_kernel_stats.csv' | head -1)" | head -8
echo R3D_OK
