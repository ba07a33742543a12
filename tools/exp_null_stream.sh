# A/B: immediate-mode work on the context's own HIP stream (VSA_EXP_OWN_STREAM=1, the old layout: 3/3/5/5 graph
# streams per hardware queue) against the null stream (4/4/4/4); alternating runs, then queue counts 4 and 8
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 120 python bench.py --streams ${S:-16} --no-cpu-baseline "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["launch_ms"]*1000,1), d["ms_per_step"])'; }
for rep in 1 2 3; do
  echo "rep $rep own-stream : $(VSA_EXP_OWN_STREAM=1 one)"
  echo "rep $rep null-stream: $(one)"
done
for s in 12 16 20 24 32; do echo "null-stream S=$s: $(S=$s one)"; done
for q in 8; do for s in 16 32; do echo "null-stream Q=$q S=$s: $(GPU_MAX_HW_QUEUES=$q S=$s one)"; done; done
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/nullstream
mkdir -p $OUT
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
cat "$OUT/bench_under_rocprof.json" | cut -c1-120
python3 $GRAFT_REPO_ROOT/tools/trace_timeline.py $OUT/trace/bench_kernel_trace.csv
