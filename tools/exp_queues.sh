set -e
cd $GRAFT_REPO_ROOT
for q in 2 4 8; do for s in 8 16; do
  echo "Q=$q S=$s: $(GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python bench.py --streams $s --no-cpu-baseline --steps 200 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["launch_ms"]*1000,1))')"
done; done
