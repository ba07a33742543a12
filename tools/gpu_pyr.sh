#!/bin/bash
# pyramid / colour convert check: bit-exact tests, then the kernels alone (one stream) under rocprofv3 --kernel-trace
set -euo pipefail
TAG=${1:-r02_pyr}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -u -m pytest tests/test_gpu_thin_layer.py tests/test_gpu_golden.py tests/test_gpu_pipeline.py -m gpu -x -q --timeout 120 --timeout-method thread -k "pyramid or color or golden or pipeline" > "$OUT/pytest.log" 2>&1 || { tail -30 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k -- python3 "$GRAFT_REPO_ROOT/tools/prof_kernels.py" frame 200 > "$OUT/trace.log" 2>&1
python3 "$GRAFT_REPO_ROOT/tools/kernel_stats.py" "$(find "$OUT/trace" -name 'k_kernel_stats.csv' | head -1)" "$OUT/kernels_alone.txt"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
cut -c1-120 "$OUT/bench.json"
echo PYR_OK
