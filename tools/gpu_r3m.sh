#!/bin/bash
# round 3, pass M: frames up to 640 x 480 build their whole pyramid in one launch (five fused steps) -- whole GPU suite, then configs[0] and the headline
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3m
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
one() { timeout -k 10 150 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["repeat_frames_per_s"]; print(round(d["value"]), "(repeats: median", round(r["median"]), "min", round(r["min"]), "max", str(round(r["max"]))+")", "launch us", round(d["roofline"]["launch_ms"]*1000,2))'; }
echo "c1      : $(one --config c1)"
echo "c1      : $(one --config c1)"
echo "c1 1 stream: $(one --config c1 --streams 1)"
echo "c2 long : $(one)"
echo R3M_OK
