#!/bin/bash
# usage: gpu_profile3.sh TAG -- round-2 profiles beyond the headline line:
#   1. rocprofv3 --kernel-trace --stats of the bench on BASELINE configs[0], [2] and [4] (short runs)
#   2. camera streams per GPU swept on the headline config
#   3. PMC passes (instruction counts, LDS, HBM traffic) over one 4K FAST stream (configs[2], thin layer) and over the video
#      stabilizer's homography / warp kernels (tools/prof_hg.py)
# Every GPU step has its own time limit; the first failure ends the script.
set -euo pipefail
TAG=${1:-r02u}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for C in c1 c3; do
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$C" -o $C -- python3 "$GRAFT_REPO_ROOT/bench.py" --config $C --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/${C}_under_rocprof.json" 2> "$OUT/$C.err"
  rm -f "$(find "$OUT/$C" -name "${C}_kernel_trace.csv" | head -1)"
done
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5" -o c5 -- python3 "$GRAFT_REPO_ROOT/bench.py" --homography --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/c5_under_rocprof.json" 2> "$OUT/c5.err"
python3 "$GRAFT_REPO_ROOT/tools/trace_timeline.py" "$(find "$OUT/c5" -name 'c5_kernel_trace.csv' | head -1)" > "$OUT/c5_concurrency.txt"
rm -f "$(find "$OUT/c5" -name 'c5_kernel_trace.csv' | head -1)"
cd "$GRAFT_REPO_ROOT"
for S in 1 2 4 8 16 24 32 48; do
  echo "streams $S: $(timeout -k 10 120 python bench.py --no-cpu-baseline --streams $S | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "frames/s,", round(d["roofline"]["launch_ms_in_timed_region"]*1000,1), "us per fused launch under this load")')" >> "$OUT/stream_sweep.txt"
done
cat "$OUT/stream_sweep.txt"
cd /tmp
i=0
for CS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
          "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
          "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
          "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d "$OUT/c3pmc$i" -o k -- python3 "$GRAFT_REPO_ROOT/tools/prof_c3.py" 10 > "$OUT/c3pmc$i.log" 2>&1
  rm -f "$(find "$OUT/c3pmc$i" -name 'k_kernel_trace.csv' | head -1)"
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d "$OUT/hgpmc$i" -o k -- python3 "$GRAFT_REPO_ROOT/tools/prof_hg.py" > "$OUT/hgpmc$i.log" 2>&1
  rm -f "$(find "$OUT/hgpmc$i" -name 'k_kernel_trace.csv' | head -1)"
done
for P in c3pmc hgpmc; do
python3 - "$OUT" $P <<'PY'
import csv, collections, glob, sys
d = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/" + sys.argv[2] + "*/**/k_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "vsa::" in r["Kernel_Name"]:
            nm = r["Kernel_Name"].split("vsa::(anonymous namespace)::")[1].split("(")[0]
            d[(nm, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/" + sys.argv[2] + "_summary.txt", "w") as out:
    for (k, c), v in sorted(d.items()):
        out.write(f"{k:34s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):16.1f}\n")
PY
done
echo PROFILE3_OK
