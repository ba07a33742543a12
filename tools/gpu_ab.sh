#!/bin/bash
# usage: gpu_ab.sh TAG "<pytest -k expression>" variant.so ... -- parity tests selected by -k, the kernels alone, then the A/B on the bench
set -euo pipefail
TAG=$1; KEXPR=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "$KEXPR" > "$OUT/pytest.log" 2>&1 || { tail -30 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k -- python3 "$GRAFT_REPO_ROOT/tools/prof_kernels.py" frame 200 > "$OUT/trace.log" 2>&1
python3 "$GRAFT_REPO_ROOT/tools/kernel_stats.py" "$(find "$OUT/trace" -name 'k_kernel_stats.csv' | head -1)" "$OUT/kernels_alone.txt" | head -6
cd "$GRAFT_REPO_ROOT"
bash tools/exp_lib_ab.sh "$@" | tee "$OUT/ab.txt"
echo AB_OK
