#!/bin/bash
# (record of a round-3 experiment: the VSA_EXP_* knobs it sets were taken out of the library again once the result was in; git log -S VSA_EXP_ finds the commits that had them)
# What does the FindHomography node cost configs[4]?  Hypothesis count K (through the confidence; K workgroups of 4 waves x 298 registers)
# and refits (the serial tail of the last workgroup), frames/s of bench.py --homography; VSA_EXP_NO_PAIR: merge pass in its own launch
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
one() { timeout -k 10 150 python bench.py --no-cpu-baseline --homography --steps 300 --repeats 3 "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["repeat_frames_per_s"]; print(round(d["value"]), "(repeats: median", round(r["median"]), ")")'; }
export VSA_EXP_NO_PAIR=1
for C in 0.49 0.739 0.932 0.995 0.99998; do
  for R in 10 0; do
    echo "confidence $C refits $R : $(VSA_EXP_HG_CONFIDENCE=$C VSA_EXP_HG_REFINE=$R one)"
  done
done
echo "K=56, 8 streams : $(one --streams 8)"
echo "K=56, 4 streams : $(one --streams 4)"
echo "K=56, 1 stream  : $(one --streams 1)"
