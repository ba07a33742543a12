#!/bin/bash
# configs[4] (homography node in every graph) is bound by its hardware queues' chains, not by wave slots: more queues / more streams?
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 150 python bench.py --no-cpu-baseline --homography "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]))'; }
echo "default: $(one) $(one)"
for q in 6 8; do echo "GPU_MAX_HW_QUEUES=$q: $(GPU_MAX_HW_QUEUES=$q one) $(GPU_MAX_HW_QUEUES=$q one)"; done
echo "48 streams: $(one --streams 48)  64 streams: $(one --streams 64)"
echo "no stream priority: $(VISUALSLAM_AMD_STREAM_PRIORITY= one)"
