#!/bin/bash
# BASELINE configs[4] on the GPU box: homography parity tests, bench line with nvxFindHomographyNode in every stream's
# graph, and kernel-trace stats of the same command.
set -euo pipefail
R=${1:-r01_hg}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -u -m pytest tests/test_gpu_homography.py -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_hg.log" 2>&1 || { tail -30 "$OUT/pytest_hg.log"; exit 1; }
tail -2 "$OUT/pytest_hg.log"
timeout -k 10 300 python bench.py --homography > "$OUT/bench_hg.json" 2> "$OUT/bench_hg.err"
cat "$OUT/bench_hg.json"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --homography --steps 100 --warmup 10 > "$OUT/bench_hg_under_rocprof.json" 2> "$OUT/trace.err"
find "$OUT" -name '*kernel_stats.csv' -exec head -8 {} \;
echo HG_OK
