#!/bin/bash
# usage: gpu_profile_r3b.sh -- round 3, the bench lines of every BASELINE config, the 2-rank rehearsal on the 1-GPU box, kernel
# traces of configs[0] / [2] / [4] under the bench and the camera-streams-per-GPU sweep.  Every GPU step has its own time limit.
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3q
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
cut -c1-200 "$OUT/bench_c2.json"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > "$OUT/bench_c2_short.json" 2> "$OUT/bench_c2_short.err"
cut -c1-200 "$OUT/bench_c2_short.json"
timeout -k 10 300 python bench.py --config c1 > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.err"
cut -c1-200 "$OUT/bench_c1.json"
timeout -k 10 400 python bench.py --config c3 --steps 300 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
cut -c1-200 "$OUT/bench_c3.json"
timeout -k 10 300 python bench.py --homography > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
cut -c1-200 "$OUT/bench_c5.json"
timeout -k 10 300 python bench.py --gpus 2 --steps 500 --no-cpu-baseline > "$OUT/bench_c2_2ranks_1gpu.json" 2> "$OUT/bench_c2_2ranks_1gpu.err"
cut -c1-200 "$OUT/bench_c2_2ranks_1gpu.json"
export TMPDIR=/tmp
cd /tmp
for C in c1 c3; do
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$C" -o $C -- python3 "$GRAFT_REPO_ROOT/bench.py" --config $C --steps 100 --warmup 10 --no-cpu-baseline --repeats 0 --preheat-s 0 > "$OUT/${C}_under_rocprof.json" 2> "$OUT/$C.err"
  python3 "$GRAFT_REPO_ROOT/tools/trace_timeline.py" "$(find "$OUT/$C" -name "${C}_kernel_trace.csv" | head -1)" > "$OUT/${C}_concurrency.txt"
  rm -f "$(find "$OUT/$C" -name "${C}_kernel_trace.csv" | head -1)"
done
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5" -o c5 -- python3 "$GRAFT_REPO_ROOT/bench.py" --homography --steps 100 --warmup 10 --no-cpu-baseline --repeats 0 --preheat-s 0 > "$OUT/c5_under_rocprof.json" 2> "$OUT/c5.err"
python3 "$GRAFT_REPO_ROOT/tools/trace_timeline.py" "$(find "$OUT/c5" -name 'c5_kernel_trace.csv' | head -1)" > "$OUT/c5_concurrency.txt"
rm -f "$(find "$OUT/c5" -name 'c5_kernel_trace.csv' | head -1)"
cd "$GRAFT_REPO_ROOT"
for S in 1 2 4 8 16 32 48; do
  echo "streams $S: $(timeout -k 10 120 python bench.py --no-cpu-baseline --streams $S --repeats 0 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "frames/s,", round(d["roofline"]["launch_ms_in_timed_region"]*1000,1), "us per fused launch under this load")')" >> "$OUT/stream_sweep.txt"
done
cat "$OUT/stream_sweep.txt"
echo PROFILE_R3B_OK
