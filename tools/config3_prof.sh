#!/bin/bash
# rocprofv3 kernel trace of config 3's group path (tools/config3_lanes.py on the engine's choice of lanes), its last path kernel by kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/c3prof
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c3prof -o t -- python3 $R/tools/config3_lanes.py 0 > $R/gpurun_out/c3prof.log 2>&1
f=$(find $R/gpurun_out/c3prof -name "*kernel_trace.csv" | head -1)
tail -2 $R/gpurun_out/c3prof.log
python3 $R/tools/path_timeline.py $f 2 1
rm -rf $R/gpurun_out/c3prof
