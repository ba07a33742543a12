#!/bin/bash
# round 3 experiment: HIP / ROCr environment knobs that touch the launch path, on the headline line and on configs[0] (whose frame is
# a chain of four short launches).  One bench run per setting; nothing in the library depends on any of them.
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
one() { timeout -k 10 150 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["repeat_frames_per_s"]; print(round(d["value"]), "(repeats: median", round(r["median"]), "min", round(r["min"]), "max", str(round(r["max"]))+")")'; }
echo "default                      c2: $(one)   c1: $(one --config c1)"
for KV in HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0 HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=1000 DEBUG_HIP_KERNARG_COPY_OPT=0; do
  export "$KV"
  echo "$KV   c2: $(one)   c1: $(one --config c1)"
  unset "${KV%%=*}"
done
echo "default again                c2: $(one)   c1: $(one --config c1)"
echo ENV_KNOBS_OK
