#!/bin/bash
# round 3, pass B: the wave-per-cell-rectangle FAST pass -- parity (whole GPU suite), then configs[2] on the bench and the kernels alone
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
for i in 1 2 3; do
timeout -k 10 300 python bench.py --config c3 --steps 300 --no-cpu-baseline > "$OUT/bench_c3_$i.json" 2> "$OUT/bench_c3_$i.err"
cut -c1-260 "$OUT/bench_c3_$i.json"
done
VISUALSLAM_AMD_NO_FUSION=1 timeout -k 10 300 python bench.py --config c3 --steps 300 --no-cpu-baseline > "$OUT/bench_c3_nofusion.json" 2> "$OUT/bench_c3_nofusion.err"
cut -c1-260 "$OUT/bench_c3_nofusion.json"
timeout -k 10 300 python bench.py --config c3 --steps 300 --streams 1 --no-cpu-baseline > "$OUT/bench_c3_s1.json" 2> "$OUT/bench_c3_s1.err"
cut -c1-260 "$OUT/bench_c3_s1.json"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/thin" -o thin -- python3 "$GRAFT_REPO_ROOT/tools/prof_c3.py" 30 > "$OUT/thin.log" 2>&1
rm -f "$(find "$OUT/thin" -name 'thin_kernel_trace.csv' | head -1)"
cat "$(find "$OUT/thin" -name 'thin_kernel_stats.csv' | head -1)" | cut -c1-200
echo R3B_OK
