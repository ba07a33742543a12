#!/bin/bash
# Runtime environment knobs on the default bench (frames/s, isolated PyrLK launch us): each knob alone against none, two repetitions.
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 120 python bench.py --no-cpu-baseline "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["launch_ms"]*1000,1))'; }
for rep in 1 2; do
  echo "rep $rep none: $(one)"
  echo "rep $rep HIP_FORCE_DEV_KERNARG=1: $(HIP_FORCE_DEV_KERNARG=1 one)"
  echo "rep $rep HIP_FORCE_DEV_KERNARG=0: $(HIP_FORCE_DEV_KERNARG=0 one)"
  echo "rep $rep HSA_ENABLE_INTERRUPT=0: $(HSA_ENABLE_INTERRUPT=0 one)"
  echo "rep $rep ROC_ACTIVE_WAIT_TIMEOUT=200: $(ROC_ACTIVE_WAIT_TIMEOUT=200 one)"
  echo "rep $rep DEBUG_CLR_BLIT_KERNARG_OPT / AMD_DIRECT_DISPATCH=0: $(AMD_DIRECT_DISPATCH=0 one)"
done
echo "single stream none: $(one --streams 1)"
echo "single stream HIP_FORCE_DEV_KERNARG=1: $(HIP_FORCE_DEV_KERNARG=1 one --streams 1)"
echo "single stream HSA_ENABLE_INTERRUPT=0: $(HSA_ENABLE_INTERRUPT=0 one --streams 1)"
echo ENV_OK
