# hardware-queue count sweep with the native host threads (GPU_MAX_HW_QUEUES is read by the HIP runtime at start-up)
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --steps 1000 --warmup 50 "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["launch_ms"]*1000,1))'; }
for q in 2 3 4 5 6; do for s in 16 24; do echo "Q=$q S=$s: $(GPU_MAX_HW_QUEUES=$q one --streams $s)"; done; done
