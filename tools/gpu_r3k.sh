#!/bin/bash
# round 3, pass K: lean homography kernel, usable test without a branch and reads ahead of the branches in the normalisation sums -- homography tests, fuzz, timeline, configs[4]
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3k
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest tests/test_gpu_homography.py tests/test_gpu_ovx.py -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
timeout -k 10 300 python -u tools/fuzz_homography.py 300 78 > "$OUT/fuzz.log" 2>&1 || { tail -30 "$OUT/fuzz.log"; exit 1; }
tail -1 "$OUT/fuzz.log"
timeout -k 10 120 python -u tools/hg_stamps.py > "$OUT/stamps.log" 2>&1 || { tail -30 "$OUT/stamps.log"; exit 1; }
grep -v "hyp:" "$OUT/stamps.log" | tail -28
one() { timeout -k 10 150 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["repeat_frames_per_s"]; print(round(d["value"]), "(repeats: median", round(r["median"]), "min", round(r["min"]), "max", str(round(r["max"]))+")", "launch us", round(d["roofline"]["launch_ms"]*1000,2))'; }
echo "c5      : $(one --homography)"
echo "c5      : $(one --homography)"
echo "c5      : $(one --homography)"
echo R3K_OK
