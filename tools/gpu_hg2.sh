#!/bin/bash
# configs[4] on the GPU box: homography parity tests, the bench line with nvxFindHomographyNode in every stream's graph,
# and the uncontended kernel durations of one stream (tools/prof_hg.py under rocprofv3 --kernel-trace).
set -euo pipefail
R=${1:-r02_hg}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -u -m pytest tests/test_gpu_homography.py tests/test_gpu_configs.py tests/test_gpu_stabilizer.py -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_hg.log" 2>&1 || { tail -30 "$OUT/pytest_hg.log"; exit 1; }
tail -2 "$OUT/pytest_hg.log"
timeout -k 10 300 python bench.py --homography > "$OUT/bench_hg.json" 2> "$OUT/bench_hg.err"
cat "$OUT/bench_hg.json"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace1" -o one -- python3 "$GRAFT_REPO_ROOT/tools/prof_hg.py" 100 > "$OUT/one_stream.log" 2>&1
cut -d, -f1-4,6,7 "$OUT/trace1"/*/one_kernel_stats.csv 2>/dev/null | cut -c1-160 || cut -d, -f1-4,6,7 "$OUT/trace1/one_kernel_stats.csv" | cut -c1-160
echo HG_OK
