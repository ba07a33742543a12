#!/bin/bash
# (record of a round-3 experiment: the VSA_EXP_* knobs it sets were taken out of the library again once the result was in; git log -S VSA_EXP_ finds the commits that had them)
# configs[4] with do-nothing stand-ins for the FindHomography kernel: (1) a few registers, (2) the real kernel's 298-register footprint
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
one() { timeout -k 10 150 python bench.py --no-cpu-baseline --homography --steps 300 --repeats 3 "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["repeat_frames_per_s"]; print(round(d["value"]), "(repeats: median", round(r["median"]), ")")'; }
export VSA_EXP_NO_PAIR=1
for i in 1 2; do
echo "real kernel            : $(one)"
echo "dummy, few registers   : $(VSA_EXP_HG_DUMMY=1 one)"
echo "dummy, 298 registers   : $(VSA_EXP_HG_DUMMY=2 one)"
echo "dummy, few regs, K=7   : $(VSA_EXP_HG_DUMMY=1 VSA_EXP_HG_CONFIDENCE=0.49 one)"
echo "dummy, 298 regs, K=7   : $(VSA_EXP_HG_DUMMY=2 VSA_EXP_HG_CONFIDENCE=0.49 one)"
done
