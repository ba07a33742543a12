# bench throughput against the number of camera streams per GPU (and one host thread scheduling all graphs)
set -e
cd $GRAFT_REPO_ROOT
for s in 1 2 4 8 12 16 24 32; do
  echo "S=$s threads: $(timeout -k 10 120 python bench.py --streams $s --no-cpu-baseline --steps 200 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["launch_ms"]*1000,1))')"
done
for s in 8 16 32; do
  echo "S=$s one host thread: $(timeout -k 10 120 python bench.py --streams $s --host-threads 0 --no-cpu-baseline --steps 200 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["launch_ms"]*1000,1))')"
done
