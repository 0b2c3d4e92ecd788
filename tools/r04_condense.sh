#!/bin/bash
# Condense what tools/r04_profile.sh <tag> collected under gpurun_out/ into the tracked files profiles/<tag>_*.
TAG=${1:-r04b}
python tools/summarize_prof.py $TAG
T=$(ls -t $(find gpurun_out/prof_t_stats -name "*kernel_stats.csv") | head -1)  # (the newest: older runs stay under gpurun_out/); cp $T profiles/${TAG}_targets_kernel_stats.csv
K=$(ls -t $(find gpurun_out/prof_t_stats -name "*kernel_trace.csv") | head -1); python tools/kernel_table.py $K > profiles/${TAG}_targets_kernel_table.txt
grep -v "^[WE]2026" gpurun_out/${TAG}_targets.log > profiles/${TAG}_targets.log
cp gpurun_out/${TAG}_bench_under_rocprof.log profiles/${TAG}_bench_under_rocprof.log
python tools/summarize_counters.py prof_t_mfma ${TAG}_targets_mfma_counters "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/r04_targets.py grams grid"
python tools/summarize_counters.py prof_t_fetch ${TAG}_targets_fetch "rocprofv3 --pmc FETCH_SIZE -- python3 tools/r04_targets.py grams grid config5 (KiB per dispatch; double for 16-byte-per-lane loads, MI355X_MICROARCH.md)"
python tools/summarize_counters.py prof_t_write ${TAG}_targets_write "rocprofv3 --pmc WRITE_SIZE -- python3 tools/r04_targets.py grams grid config5 (KiB per dispatch)"
P=$(ls -t $(find gpurun_out/prof_stats -name "*kernel_trace.csv") | head -1); python tools/path_timeline.py $P 9 1 > profiles/${TAG}_path_timeline.txt
python tools/update_roofline_traffic.py $TAG
ls -la profiles/${TAG}_*
