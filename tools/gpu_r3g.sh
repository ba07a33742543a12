#!/bin/bash
# (record of a round-3 experiment: the VSA_EXP_* knobs it sets were taken out of the library again once the result was in; git log -S VSA_EXP_ finds the commits that had them)
# round 3, pass G: FindHomography's hand-over past the L2s (no device-scope release / acquire): homography tests incl. the 12-stream
# one, configs[4] fused / apart, configs[1], then the whole GPU suite
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest tests/test_gpu_ovx.py tests/test_gpu_homography.py tests/test_gpu_stabilizer.py tests/test_gpu_configs.py -m gpu -x -q --timeout 120 --timeout-method thread -k "homography or stabilizer or config4" > "$OUT/pytest_pair.log" 2>&1 || { tail -40 "$OUT/pytest_pair.log"; exit 1; }
tail -3 "$OUT/pytest_pair.log"
one() { timeout -k 10 150 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["repeat_frames_per_s"]; print(round(d["value"]), "(repeats: median", round(r["median"]), "min", round(r["min"]), "max", str(round(r["max"]))+")", "node ms", d.get("homography",{}).get("node_ms_in_timed_region"))'; }
echo "c2            : $(one --steps 300)"
for i in 1 2; do
echo "c5 pair fused : $(one --homography --steps 300)"
echo "c5 pair apart : $(VSA_EXP_NO_PAIR=1 one --homography --steps 300)"
done
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
echo R3G_OK
