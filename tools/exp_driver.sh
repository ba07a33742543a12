# A/B: Python host threads against native host threads driving the same graphs (default bench otherwise)
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 120 python bench.py --no-cpu-baseline "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["launch_ms"]*1000,1), round(d["tracked_per_frame"],1))'; }
for rep in 1 2; do
  echo "rep $rep python S=16: $(one --driver python)"
  echo "rep $rep native S=16: $(one --driver native --steps 1000 --warmup 50)"
done
for s in 1 2 4 8 12 24 32; do echo "native S=$s: $(one --driver native --streams $s --steps 1000 --warmup 50)"; done
for q in 8; do for s in 16 32; do echo "native Q=$q S=$s: $(GPU_MAX_HW_QUEUES=$q one --streams $s --steps 1000 --warmup 50)"; done; done
