# A/B of the bench's streams-per-GPU default: alternating runs of S=8, 12, 16 (default step count)
set -e
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for s in 8 12 16; do
    echo "rep $rep S=$s: $(timeout -k 10 120 python bench.py --streams $s --no-cpu-baseline | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["launch_ms"]*1000,1), d["ms_per_step"])')"
  done
done
