#!/bin/bash
# usage: gpu_profile_r3c.sh -- the bench lines alone (every BASELINE config + the 2-rank rehearsal), for profiles/r03_bench_*.json
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3r
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
cut -c1-120 "$OUT/bench_c2.json"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > "$OUT/bench_c2_short.json" 2> "$OUT/bench_c2_short.err"
cut -c1-120 "$OUT/bench_c2_short.json"
timeout -k 10 300 python bench.py --config c1 > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.err"
cut -c1-120 "$OUT/bench_c1.json"
timeout -k 10 400 python bench.py --config c3 --steps 300 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
cut -c1-120 "$OUT/bench_c3.json"
timeout -k 10 300 python bench.py --homography > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
cut -c1-120 "$OUT/bench_c5.json"
timeout -k 10 300 python bench.py --gpus 2 --steps 500 --no-cpu-baseline > "$OUT/bench_c2_2ranks_1gpu.json" 2> "$OUT/bench_c2_2ranks_1gpu.err"
cut -c1-120 "$OUT/bench_c2_2ranks_1gpu.json"
echo PROFILE_R3C_OK
