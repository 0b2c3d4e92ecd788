#!/bin/bash
# Condense what tools/r06_profile.sh <tag> collected under gpurun_out/ into the tracked files profiles/<tag>_*.
TAG=${1:-r06}
python tools/summarize_prof.py $TAG
cp gpurun_out/${TAG}_bench_under_rocprof.log profiles/${TAG}_bench_under_rocprof.log
P=$(ls -t $(find gpurun_out/prof_stats -name "*kernel_trace.csv") | head -1); python tools/path_timeline.py $P 7 1 > profiles/${TAG}_path_timeline.txt
P=$(ls -t $(find gpurun_out/prof_light -name "*kernel_trace.csv") | head -1); python tools/path_timeline.py $P 2 1 > profiles/${TAG}_light_timeline.txt
python tools/update_roofline_traffic.py $TAG
cp gpurun_out/${TAG}_bench.log profiles/${TAG}_bench.log
for f in headline_law_draws draws_ab soak_ab lanes_sweep config3_lanes headline_soak group_soak config4_share_kernels config4_over_x_kernels; do [ -f gpurun_out/${TAG}_$f.log ] && cp gpurun_out/${TAG}_$f.log profiles/${TAG}_$f.log; done
for f in ws_fuzz mg_fuzz carry_fuzz covariance_fuzz on_chip_fuzz edge_cases; do [ -f gpurun_out/${TAG}_$f.log ] && tail -4 gpurun_out/${TAG}_$f.log > profiles/${TAG}_$f.txt; done
ls -la profiles/${TAG}_*
