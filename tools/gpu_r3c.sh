#!/bin/bash
# (record of a round-3 experiment: the VSA_EXP_* knobs it sets were taken out of the library again once the result was in; git log -S VSA_EXP_ finds the commits that had them)
# round 3, pass C: FAST pass with running minima -- parity, c3 bench, the kernel alone at band heights 18 / 36 / 54
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "fast or Fast or config2 or tracker or golden" > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
for i in 1 2 3; do
timeout -k 10 300 python bench.py --config c3 --steps 300 --no-cpu-baseline > "$OUT/bench_c3_$i.json" 2> "$OUT/bench_c3_$i.err"
cut -c1-260 "$OUT/bench_c3_$i.json"
done
timeout -k 10 300 python bench.py --config c3 --steps 300 --streams 1 --no-cpu-baseline > "$OUT/bench_c3_s1.json" 2> "$OUT/bench_c3_s1.err"
cut -c1-260 "$OUT/bench_c3_s1.json"
export TMPDIR=/tmp
cd /tmp
for B in 18 36 54; do
VSA_EXP_FS_BAND=$B timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/thin$B" -o thin -- python3 "$GRAFT_REPO_ROOT/tools/prof_c3.py" 30 > "$OUT/thin$B.log" 2>&1
rm -f "$(find "$OUT/thin$B" -name 'thin_kernel_trace.csv' | head -1)"
echo "band $B: $(grep k_fast_cells "$(find "$OUT/thin$B" -name 'thin_kernel_stats.csv' | head -1)" | cut -d, -f2-4)"
done
echo R3C_OK
