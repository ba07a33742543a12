# camera streams per GPU with the native host threads (default bench otherwise)
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 120 python bench.py --no-cpu-baseline "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["launch_ms"]*1000,1))'; }
for rep in 1 2; do for s in 1 2 4 8 12 16 20 24 32 48; do echo "rep $rep S=$s: $(one --streams $s)"; done; done
