set -e
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 120 python bench.py --homography "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["homography"]["node_ms_in_timed_region"]*1000,1), d["homography"]["inliers_stream0"])'; }
i=0
for v in "$@"; do i=$((i+1)); d=/tmp/vsa_variant_$i; mkdir -p $d; cp $GRAFT_REPO_ROOT/$v $d/libvisualslam_amd.so; cp visualslam_amd/lib/libvsa_stream_driver.so $d/; done
for rep in 1 2; do
  echo "rep $rep in-tree: $(one)"
  i=0; for v in "$@"; do i=$((i+1)); echo "rep $rep $v: $(VISUALSLAM_AMD_LIB=/tmp/vsa_variant_$i/libvisualslam_amd.so one)"; done
done
echo "single stream in-tree: $(one --streams 1)"
i=0; for v in "$@"; do i=$((i+1)); echo "single stream $v: $(VISUALSLAM_AMD_LIB=/tmp/vsa_variant_$i/libvisualslam_amd.so one --streams 1)"; done
