# A/B of library builds on the default bench: alternating runs of the in-tree library and the variants named on the command line
# (paths relative to the repository root; variants are driven by Python threads: the native driver library binds the in-tree build), three repetitions each; also the isolated PyrLK launch time of each (single stream)
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 120 python bench.py --no-cpu-baseline "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["launch_ms"]*1000,1))'; }
for rep in 1 2 3; do
  echo "rep $rep in-tree: $(one)"
  for v in "$@"; do echo "rep $rep $v: $(VISUALSLAM_AMD_LIB=$GRAFT_REPO_ROOT/$v one --driver python)"; done
done
echo "single stream in-tree: $(one --streams 1)"
for v in "$@"; do echo "single stream $v: $(VISUALSLAM_AMD_LIB=$GRAFT_REPO_ROOT/$v one --streams 1 --driver python)"; done
