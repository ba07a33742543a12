#!/bin/bash
# is the first 20-step block slow because the GPU has only seen 5 warm-up steps?
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3fb
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for W in 5 50 200; do for S in 32 16; do
  timeout -k 10 200 python bench.py --steps 20 --warmup $W --streams $S --no-cpu-baseline > "$OUT/w${W}_s${S}.json" 2> "$OUT/w${W}_s${S}.err"
  python3 - "$OUT/w${W}_s${S}.json" $W $S <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["repeat_frames_per_s"]
print(f"warmup {sys.argv[2]:>3} streams {sys.argv[3]:>2}: first block {d['value']:.0f}  repeats median {r['median']:.0f} min {r['min']:.0f} max {r['max']:.0f}")
PY
done; done
echo FB_OK
