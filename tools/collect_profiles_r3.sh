#!/bin/bash
# copies what tools/gpu_profile_r3a.sh and gpu_profile_r3b.sh left under gpurun_out/ into profiles/r03_* and stamps profiles/frame_counters.json
set -euo pipefail
cd "$(dirname "$0")/.."
P=gpurun_out/r3p; Q=gpurun_out/r3q
cp $P/bench_concurrency.txt profiles/r03_bench_concurrency.txt
cp $P/bench_under_rocprof.json profiles/r03_bench_under_rocprof.json
cp "$(find $P/bench -name 'bench_kernel_stats.csv')" profiles/r03_bench_kernel_stats.csv
cp "$(find $P/single -name 'single_kernel_stats.csv')" profiles/r03_single_stream_kernel_stats.csv
cp $P/single_under_rocprof.json profiles/r03_single_stream_under_rocprof.json
cp "$(find $P/thin -name 'thin_kernel_stats.csv')" profiles/r03_thin_layer_kernel_stats.csv
cp $P/pmc_summary.txt profiles/r03_pmc_graph_frame_counters.txt
cp $P/c3pmc_summary.txt profiles/r03_pmc_c3_4k_fast_graph_counters.txt
for c in c1 c3 c5; do
  cp $Q/${c}_concurrency.txt profiles/r03_bench_${c}_concurrency.txt
  cp $Q/${c}_under_rocprof.json profiles/r03_bench_${c}_under_rocprof.json
  cp "$(find $Q/$c -name "${c}_kernel_stats.csv")" profiles/r03_bench_${c}_kernel_stats.csv
done
for f in c1 c2 c2_short c3 c5 c2_2ranks_1gpu; do cp $Q/bench_$f.json profiles/r03_bench_$f.json; done
cp $Q/stream_sweep.txt profiles/r03_stream_sweep.txt
python tools/kernel_resources.py profiles/r03_kernel_resources.txt > /dev/null
python tools/make_frame_counters.py profiles/r03_pmc_graph_frame_counters.txt profiles/frame_counters.json > /dev/null
python tools/make_frame_counters.py --check profiles/frame_counters.json
