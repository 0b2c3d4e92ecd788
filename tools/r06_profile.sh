#!/bin/bash
# The profile set of round 6, collection: run on the GPU box from the repository root as `bash tools/r06_profile.sh <tag> [part]`
# (part 1: rocprofv3 summaries + the bench line in ONE lease -- so that the kernel average of profiles/<tag>_kernel_stats.csv and
# the line's avg_kernel_ms come from one box (round-5 verdict, item 3) -- and the timelines; part 2: soaks and fuzzers);
# everything lands under gpurun_out/ (the only directory that travels back); tools/r06_condense.sh <tag> writes profiles/<tag>_*.
# Every rocprofv3 run has the program itself after `--`; counters run in passes of their own (kernel trace / stats only beside them).
set -o pipefail
TAG=${1:-r06}
PART=${2:-1}
R=$(pwd)
if [ "$PART" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  B="python3 $R/bench.py --steps 6 --warmup 2 --cpu-budget 0 --no-extra"
  rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write $R/gpurun_out/prof_light
  $B > /dev/null 2>&1  # (a fresh box runs its first minute slower: the profiled run is not the first thing it does)
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- $B > $R/gpurun_out/${TAG}_bench_under_rocprof.log 2> $R/gpurun_out/${TAG}_bench_under_rocprof.err && echo stats ok
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-extra > /dev/null 2>&1 && echo fetch ok
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-extra > /dev/null 2>&1 && echo write ok
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_light -- python3 $R/tools/light_timeline.py 7 > $R/gpurun_out/${TAG}_light_timeline.log 2>&1 && echo light ok
  cd $R
  python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err && echo bench ok
  python tools/headline_data_seeds.py > gpurun_out/${TAG}_headline_law_draws.log 2>&1 && echo draws ok
  echo "SLM_NO_LIGHT_PASS=1" >> gpurun_out/${TAG}_headline_law_draws.log; SLM_NO_LIGHT_PASS=1 python tools/headline_data_seeds.py >> gpurun_out/${TAG}_headline_law_draws.log 2>&1
  # nine draws, settings alternating on each: the defaults, no light passes, a miss always at four appends' worth (rounds 4-5), both
  python tools/ab_knobs_draws.py "" "SLM_NO_LIGHT_PASS=1" "SLM_WS_MISS_DIV=1 SLM_WS_MISS_FACTOR=4" "SLM_NO_LAG_HANDOVER=1" "SLM_NO_LIGHT_PASS=1 SLM_WS_MISS_DIV=1 SLM_NO_LAG_HANDOVER=1" 9 > gpurun_out/${TAG}_draws_ab.log 2>&1 && echo draws ab ok
  python tools/ab_knobs_soak.py 30 "" "SLM_NO_LAG_HANDOVER=1" "SLM_WS_MISS_DIV=1" > gpurun_out/${TAG}_soak_ab.log 2>&1 && echo soak ab ok
else
  python tools/lanes_sweep.py 16 18 20 25 32 0 > gpurun_out/${TAG}_lanes_sweep.log 2>&1 && echo lanes ok
  python tools/config3_lanes.py 16 20 25 32 0 > gpurun_out/${TAG}_config3_lanes.log 2>&1 && echo config3 ok
  python tools/headline_soak.py 48 > gpurun_out/${TAG}_headline_soak.log 2>&1 && echo soak ok
  python tools/group_soak.py 24 > gpurun_out/${TAG}_group_soak.log 2>&1 && echo group soak ok
  bash tools/share_prof.sh 8 0 1 > gpurun_out/${TAG}_config4_share_kernels.log 2>&1 && echo shares ok
  bash tools/share_prof.sh 1 0 > gpurun_out/${TAG}_config4_over_x_kernels.log 2>&1 && echo grid ok
  python tools/ws_fuzz.py 800 17 > gpurun_out/${TAG}_ws_fuzz.log 2>&1; tail -2 gpurun_out/${TAG}_ws_fuzz.log
  python tools/mg_fuzz.py 40 5 > gpurun_out/${TAG}_mg_fuzz.log 2>&1; tail -2 gpurun_out/${TAG}_mg_fuzz.log
  python tools/carry_fuzz.py $(seq 0 11) > gpurun_out/${TAG}_carry_fuzz.log 2>&1; tail -2 gpurun_out/${TAG}_carry_fuzz.log
  python tools/covariance_fuzz.py 100 3 > gpurun_out/${TAG}_covariance_fuzz.log 2>&1; tail -2 gpurun_out/${TAG}_covariance_fuzz.log
  python tools/on_chip_fuzz.py 300 5 > gpurun_out/${TAG}_on_chip_fuzz.log 2>&1; tail -2 gpurun_out/${TAG}_on_chip_fuzz.log
  python tools/edge_cases.py > gpurun_out/${TAG}_edge_cases.log 2>&1; tail -2 gpurun_out/${TAG}_edge_cases.log
fi
echo collected part $PART: run tools/r06_condense.sh $TAG where gpurun_out/ has been merged back
