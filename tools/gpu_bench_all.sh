#!/bin/bash
# usage: gpu_bench_all.sh TAG  -- the bench lines of every BASELINE config + the 2-rank rehearsal on the 1-GPU box.
# Every GPU step has its own time limit; the first failure ends the script.
set -euo pipefail
TAG=${1:-r02b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
cat "$OUT/bench_c2.json"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_c2_short.json" 2> "$OUT/bench_c2_short.err"
cat "$OUT/bench_c2_short.json"
timeout -k 10 300 python bench.py --config c1 > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.err"
cat "$OUT/bench_c1.json"
timeout -k 10 400 python bench.py --config c3 --steps 300 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
cat "$OUT/bench_c3.json"
timeout -k 10 300 python bench.py --homography > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
cat "$OUT/bench_c5.json"
timeout -k 10 300 python bench.py --gpus 2 --streams 16 --steps 500 --no-cpu-baseline > "$OUT/bench_c2_2ranks_1gpu.json" 2> "$OUT/bench_c2_2ranks_1gpu.err"
cat "$OUT/bench_c2_2ranks_1gpu.json"
timeout -k 10 300 python bench.py --ring 8 --no-cpu-baseline > "$OUT/bench_c2_ring8.json" 2> "$OUT/bench_c2_ring8.err"
cat "$OUT/bench_c2_ring8.json"
echo BENCH_ALL_OK
