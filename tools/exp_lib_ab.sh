# A/B of library builds on the default bench (or on BENCH_ARGS, e.g. BENCH_ARGS="--config c3 --steps 300"): alternating runs of the in-tree library and the variants named on the command line
# (paths relative to the repository root), three repetitions each; then the single-stream rate and isolated PyrLK launch time of each.
# A variant is copied to a directory of its own under the in-tree name, with the native driver library beside it (the driver
# binds libvisualslam_amd.so from its own directory).
set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 200 python bench.py --no-cpu-baseline $BENCH_ARGS "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["launch_ms"]*1000,1))'; }
i=0
for v in "$@"; do
  i=$((i+1)); d=/tmp/vsa_variant_$i; mkdir -p $d
  cp $GRAFT_REPO_ROOT/$v $d/libvisualslam_amd.so; cp visualslam_amd/lib/libvsa_stream_driver.so $d/
done
for rep in 1 2 3; do
  echo "rep $rep in-tree: $(one)"
  i=0; for v in "$@"; do i=$((i+1)); echo "rep $rep $v: $(VISUALSLAM_AMD_LIB=/tmp/vsa_variant_$i/libvisualslam_amd.so one)"; done
done
echo "single stream in-tree: $(one --streams 1)"
i=0; for v in "$@"; do i=$((i+1)); echo "single stream $v: $(VISUALSLAM_AMD_LIB=/tmp/vsa_variant_$i/libvisualslam_amd.so one --streams 1)"; done
