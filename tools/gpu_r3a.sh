#!/bin/bash
# round 3, pass A: the whole GPU test suite, then the driver-sized bench line at 32 and 16 streams and the default line
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > "$OUT/bench_short32.json" 2> "$OUT/bench_short32.err"
cut -c1-400 "$OUT/bench_short32.json"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --streams 16 --no-cpu-baseline > "$OUT/bench_short16.json" 2> "$OUT/bench_short16.err"
cut -c1-200 "$OUT/bench_short16.json"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --streams 8 --no-cpu-baseline > "$OUT/bench_short8.json" 2> "$OUT/bench_short8.err"
cut -c1-200 "$OUT/bench_short8.json"
timeout -k 10 300 python bench.py --no-cpu-baseline > "$OUT/bench_long.json" 2> "$OUT/bench_long.err"
cut -c1-200 "$OUT/bench_long.json"
echo R3A_OK
