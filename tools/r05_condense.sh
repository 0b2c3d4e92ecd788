#!/bin/bash
# Condense what tools/r05_profile.sh <tag> collected under gpurun_out/ into the tracked files profiles/<tag>_*.
TAG=${1:-r05c}
python tools/summarize_prof.py $TAG
cp gpurun_out/${TAG}_bench_under_rocprof.log profiles/${TAG}_bench_under_rocprof.log
P=$(ls -t $(find gpurun_out/prof_stats -name "*kernel_trace.csv") | head -1); python tools/path_timeline.py $P 7 1 > profiles/${TAG}_path_timeline.txt
python tools/update_roofline_traffic.py $TAG
cp gpurun_out/${TAG}_bench.log profiles/${TAG}_bench.log
for f in lanes_sweep config3_lanes headline_law_draws headline_soak group_soak; do cp gpurun_out/${TAG}_$f.log profiles/${TAG}_$f.log; done
for f in ws_fuzz mg_fuzz carry_fuzz covariance_fuzz on_chip_fuzz edge_cases; do tail -4 gpurun_out/${TAG}_$f.log > profiles/${TAG}_$f.txt; done
ls -la profiles/${TAG}_*
