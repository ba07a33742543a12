#!/bin/bash
# round 3, pass L: the homography kernels as committed -- randomised sweep (600 cases) and the concurrent soak, outputs kept for profiles/
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3l
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -u tools/fuzz_homography.py 600 2024 > "$OUT/fuzz.log" 2>&1 || { tail -30 "$OUT/fuzz.log"; exit 1; }
tail -1 "$OUT/fuzz.log"
timeout -k 10 400 python -u tools/soak_homography.py > "$OUT/soak.log" 2>&1 || { tail -30 "$OUT/soak.log"; exit 1; }
tail -1 "$OUT/soak.log"
echo R3L_OK
