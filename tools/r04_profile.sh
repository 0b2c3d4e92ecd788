#!/bin/bash
# The profile set of a round, collection: run on the GPU box from the repository root as `bash tools/r04_profile.sh <tag>` (e.g. r04b);
# everything lands under gpurun_out/ (the only directory that travels back); tools/r04_condense.sh <tag> then writes profiles/<tag>_*.
# Every rocprofv3 run has the program itself after `--`; counters run in passes of their own.
set -o pipefail
TAG=${1:-r04a}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 6 --warmup 2 --cpu-budget 0 --no-extra"
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write $R/gpurun_out/prof_t_stats $R/gpurun_out/prof_t_mfma $R/gpurun_out/prof_t_fetch $R/gpurun_out/prof_t_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- $B > $R/gpurun_out/${TAG}_bench_under_rocprof.log 2> $R/gpurun_out/${TAG}_bench_under_rocprof.err && echo stats ok
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-extra > /dev/null 2>&1 && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-extra > /dev/null 2>&1 && echo write ok
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_t_stats -- python3 $R/tools/r04_targets.py > /dev/null 2> $R/gpurun_out/${TAG}_targets.log && echo targets stats ok
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_t_mfma -- python3 $R/tools/r04_targets.py grams grid > /dev/null 2>&1 && echo targets mfma ok
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_t_fetch -- python3 $R/tools/r04_targets.py grams grid config5 > /dev/null 2>&1 && echo targets fetch ok
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_t_write -- python3 $R/tools/r04_targets.py grams grid config5 > /dev/null 2>&1 && echo targets write ok
cd $R
echo collected: run tools/r04_condense.sh $TAG where gpurun_out/ has been merged back
