#!/bin/bash
# usage: gpu_profile_r3a.sh -- round 3, the profiles behind bench.py's roofline objects (headline configuration):
#   1. the whole GPU test suite
#   2. rocprofv3 --kernel-trace --stats of the default bench command (short run) + its concurrency timeline
#   3. the same of a single-stream run (the kernels ALONE: what roofline.launch_ms must agree with) and of the thin layer's five launches
#   4. PMC passes over one stream's frames through the OpenVX graph (tools/prof_kernels.py graph): SQ instruction counts, LDS, FETCH_SIZE, WRITE_SIZE
# Every GPU step has its own time limit; the first failure ends the script.
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3p
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
tail -2 "$OUT/pytest_gpu.log"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 100 --warmup 10 --no-cpu-baseline --repeats 0 --preheat-s 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench.err"
cut -c1-200 "$OUT/bench_under_rocprof.json"
python3 "$GRAFT_REPO_ROOT/tools/trace_timeline.py" "$(find "$OUT/bench" -name 'bench_kernel_trace.csv' | head -1)" > "$OUT/bench_concurrency.txt"
cat "$OUT/bench_concurrency.txt"
rm -f "$(find "$OUT/bench" -name 'bench_kernel_trace.csv' | head -1)"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/single" -o single -- python3 "$GRAFT_REPO_ROOT/bench.py" --streams 1 --steps 300 --warmup 20 --no-cpu-baseline --repeats 0 --preheat-s 0 > "$OUT/single_under_rocprof.json" 2> "$OUT/single.err"
cut -c1-200 "$OUT/single_under_rocprof.json"
rm -f "$(find "$OUT/single" -name 'single_kernel_trace.csv' | head -1)"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/thin" -o thin -- python3 "$GRAFT_REPO_ROOT/tools/prof_kernels.py" frame 200 > "$OUT/thin.log" 2>&1
rm -f "$(find "$OUT/thin" -name 'thin_kernel_trace.csv' | head -1)"
i=0
for CS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
          "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
          "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
          "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d "$OUT/pmc$i" -o k -- python3 "$GRAFT_REPO_ROOT/tools/prof_kernels.py" graph 20 > "$OUT/pmc$i.log" 2>&1
  rm -f "$(find "$OUT/pmc$i" -name 'k_kernel_trace.csv' | head -1)"
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d "$OUT/c3pmc$i" -o k -- python3 "$GRAFT_REPO_ROOT/tools/prof_kernels.py" graph 10 c3 > "$OUT/c3pmc$i.log" 2>&1
  rm -f "$(find "$OUT/c3pmc$i" -name 'k_kernel_trace.csv' | head -1)"
done
for P in pmc c3pmc; do
python3 - "$OUT" $P <<'PY'
import csv, collections, glob, sys
d = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/" + sys.argv[2] + "[0-9]/**/k_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "vsa::" in r["Kernel_Name"]:
            nm = r["Kernel_Name"].split("vsa::(anonymous namespace)::")[1].split("(")[0]
            d[(nm, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/" + sys.argv[2] + "_summary.txt", "w") as out:
    for (k, c), v in sorted(d.items()):
        out.write(f"{k:38s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):16.1f}\n")
PY
done
head -30 "$OUT/pmc_summary.txt"
echo PROFILE_R3A_OK
