#!/bin/bash
# usage: collect_profiles.sh TAG -- copies what tools/gpu_bench_all.sh TAG and tools/gpu_profile2.sh TAG left under gpurun_out/TAG into
# profiles/ under round-2 names and rebuilds profiles/frame_counters.json (run in the container, from the repository root)
set -e
R=gpurun_out/$1
for c in c1 c2 c3 c5 c2_short c2_ring8 c2_2ranks_1gpu; do cp $R/bench_$c.json profiles/r02_bench_$c.json; done
cp $R/bench/bench_kernel_stats.csv profiles/r02_bench_kernel_stats.csv
cp $R/single/single_kernel_stats.csv profiles/r02_single_stream_kernel_stats.csv
cp $R/thin/thin_kernel_stats.csv profiles/r02_thin_layer_kernel_stats.csv
cp $R/bench_concurrency.txt profiles/r02_bench_concurrency.txt
cp $R/pmc_summary.txt profiles/r02_pmc_graph_frame_counters.txt
cp $R/bench_under_rocprof.json profiles/r02_bench_under_rocprof.json
cp $R/single_under_rocprof.json profiles/r02_single_stream_under_rocprof.json
python tools/make_frame_counters.py $R/pmc_summary.txt profiles/frame_counters.json > /dev/null
python tools/kernel_resources.py profiles/r02_kernel_resources.txt > /dev/null
