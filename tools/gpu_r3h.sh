#!/bin/bash
# (record of a round-3 experiment: the VSA_EXP_* knobs it sets were taken out of the library again once the result was in; git log -S VSA_EXP_ finds the commits that had them)
# round 3, pass H: FindHomography in 96 registers (k_find_homography_lean): parity tests of everything with a homography in it, then
# configs[4] lean / fat (VSA_EXP_HG_FAT) x pair fused / apart (VSA_EXP_NO_PAIR), kernel alone, then the whole GPU suite
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3h
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest tests/test_gpu_ovx.py tests/test_gpu_homography.py tests/test_gpu_stabilizer.py tests/test_gpu_configs.py -m gpu -x -q --timeout 120 --timeout-method thread -k "homography or stabilizer or config4" > "$OUT/pytest_pair.log" 2>&1 || { tail -40 "$OUT/pytest_pair.log"; exit 1; }
tail -3 "$OUT/pytest_pair.log"
one() { timeout -k 10 150 python bench.py --no-cpu-baseline --homography --steps 300 --repeats 3 "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["repeat_frames_per_s"]; print(round(d["value"]), "(repeats: median", round(r["median"]), ")")'; }
for i in 1 2; do
echo "lean, pair fused : $(one)"
echo "lean, pair apart : $(VSA_EXP_NO_PAIR=1 one)"
echo "lean, pair fused, 16 streams : $(one --streams 16)"
done
echo "lean, pair apart, 1 stream : $(VSA_EXP_NO_PAIR=1 one --streams 1)"
echo "fat,  pair apart, 1 stream : $(VSA_EXP_HG_FAT=1 VSA_EXP_NO_PAIR=1 one --streams 1)"
echo "lean, pair fused, 1 stream : $(one --streams 1)"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
echo R3H_OK
