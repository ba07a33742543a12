#!/bin/bash
# (record of a round-3 experiment: the VSA_EXP_* knobs it sets were taken out of the library again once the result was in; git log -S VSA_EXP_ finds the commits that had them)
# round 3, pass F: configs[4] with the detector's merge pass inside the FindHomography launch against the two launches apart
# (VSA_EXP_NO_PAIR), configs[1] for reference, then the whole GPU suite
set -euo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
one() { timeout -k 10 150 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["repeat_frames_per_s"]; print(round(d["value"]), "(repeats: median", round(r["median"]), "min", round(r["min"]), "max", str(round(r["max"]))+")", "node ms", d.get("homography",{}).get("node_ms_in_timed_region"))'; }
echo "c2            : $(one --steps 300)"
for i in 1 2; do
echo "c5 pair fused : $(one --homography --steps 300)"
echo "c5 pair apart : $(VSA_EXP_NO_PAIR=1 one --homography --steps 300)"
done
echo "c5 pair fused, 16 streams : $(one --homography --steps 300 --streams 16)"
echo "c5 pair apart, 16 streams : $(VSA_EXP_NO_PAIR=1 one --homography --steps 300 --streams 16)"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
echo R3F_OK
