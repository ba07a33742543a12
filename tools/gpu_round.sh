#!/bin/bash
# One GPU-box pass: parity tests, the default bench line, kernel-trace stats of the same command,
# and the two HBM-traffic PMC passes.  Every GPU step has its own time limit; the first failure ends the script.
set -euo pipefail
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1
tail -2 "$OUT/pytest_gpu.log"
timeout -k 10 300 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cat "$OUT/bench.json"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
cat "$OUT/bench_under_rocprof.json"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o frame -- python3 "$GRAFT_REPO_ROOT/tools/prof_kernels.py" frame 20 > "$OUT/pmc_fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o frame -- python3 "$GRAFT_REPO_ROOT/tools/prof_kernels.py" frame 20 > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name '*.csv' | head -20
echo ROUND_OK
