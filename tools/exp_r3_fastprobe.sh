#!/bin/bash
# (record of a round-3 experiment: the VSA_EXP_* knobs it sets were taken out of the library again once the result was in; git log -S VSA_EXP_ finds the commits that had them)
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
for M in 0 1 2 4 6; do
  VSA_EXP_FAST_MODE=$M timeout -k 5 100 python tools/fast_probe.py 300
done
for B in 36 54; do
  VSA_EXP_FS_BAND=$B timeout -k 5 100 python tools/fast_probe.py 300
done
timeout -k 5 100 python tools/fast_probe.py 300 1920 1080
echo PROBE_OK
