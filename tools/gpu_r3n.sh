#!/bin/bash
# round 3, pass N: the any-size homography kernel rebuilt on the 96-register parts (points streamed from memory in every pass) --
# homography tests and the randomised sweep with every case forced through it (VISUALSLAM_AMD_HG_GENERIC=1) and as dispatched, then configs[4] with it
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3n
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for G in 1 0; do
  if [ $G = 1 ]; then export VISUALSLAM_AMD_HG_GENERIC=1; else unset VISUALSLAM_AMD_HG_GENERIC; fi
  timeout -k 10 600 python -u -m pytest tests/test_gpu_homography.py tests/test_gpu_ovx.py tests/test_gpu_configs.py tests/test_gpu_stabilizer.py -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_gpu_$G.log" 2>&1 || { tail -40 "$OUT/pytest_gpu_$G.log"; exit 1; }
  tail -1 "$OUT/pytest_gpu_$G.log"
  timeout -k 10 300 python -u tools/fuzz_homography.py 600 $((90 + G)) > "$OUT/fuzz_$G.log" 2>&1 || { tail -30 "$OUT/fuzz_$G.log"; exit 1; }
  tail -1 "$OUT/fuzz_$G.log"
done
one() { timeout -k 10 150 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["repeat_frames_per_s"]; print(round(d["value"]), "(repeats: median", round(r["median"]), "min", round(r["min"]), "max", str(round(r["max"]))+")")'; }
echo "c5, on-chip kernel  : $(one --homography)"
export VISUALSLAM_AMD_HG_GENERIC=1
echo "c5, streaming kernel: $(one --homography)"
echo "c5, streaming kernel: $(one --homography)"
unset VISUALSLAM_AMD_HG_GENERIC
echo "c5, on-chip kernel  : $(one --homography)"
echo R3N_OK
