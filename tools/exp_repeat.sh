# the bench line several times in a row with changing stream counts (stderr kept): looks for run-to-run variation or failures
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/repeat
i=0
for s in 4 8 2 8 12 8 1 8 16 8 8 8; do
  i=$((i+1))
  timeout -k 10 120 python bench.py --streams $s --no-cpu-baseline --steps 200 > gpurun_out/repeat/out$i.json 2> gpurun_out/repeat/err$i.log
  echo "run $i S=$s rc=$? $(cut -c50-90 gpurun_out/repeat/out$i.json) | $(grep -v amdgpu.ids gpurun_out/repeat/err$i.log | tail -3 | cut -c1-400)"
done
