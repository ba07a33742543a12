#!/bin/bash
# (1) the driver-sized block with and without the untimed pre-heat, 32 and 16 camera streams;
# (2) hardware queues per process (GPU_MAX_HW_QUEUES; the box exports HIP's default, 4) on the latency-bound lines: configs[0], configs[4], configs[2]
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3queues
one() { timeout -k 10 150 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["repeat_frames_per_s"]; print(round(d["value"]), "(repeats: median", round(r["median"]), "min", round(r["min"]), "max", str(round(r["max"]))+")", "preheat steps", d["preheat"]["steps"])'; }
echo "== driver-sized block (--steps 20 --warmup 5)"
for P in 0 0.5 0 0.5 1.0; do
  echo "  preheat $P s, 32 streams: $(one --steps 20 --warmup 5 --preheat-s $P)"
  echo "  preheat $P s, 16 streams: $(one --steps 20 --warmup 5 --preheat-s $P --streams 16)"
done
echo "== hardware queues per process"
for Q in 4 8 16; do
  export GPU_MAX_HW_QUEUES=$Q
  echo "  queues $Q  c1: $(one --config c1 --steps 200)"
  echo "  queues $Q  c1, 64 streams: $(one --config c1 --steps 200 --streams 64)"
  echo "  queues $Q  c5: $(one --homography --steps 200)"
  echo "  queues $Q  c3: $(one --config c3 --steps 100)"
  echo "  queues $Q  c2: $(one --steps 200)"
done
