#!/bin/bash
# rocprofv3 kernel trace of shares of config 4, condensed: usage share_prof.sh WORLD RANK [RANK ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=$1; shift
for rk in "$@"; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/share_${W}_$rk -o t -- python3 $R/tools/config4_share_trace.py $rk $W 3 > $R/gpurun_out/share_${W}_$rk.log 2>&1
  f=$(find $R/gpurun_out/share_${W}_$rk -name "*kernel_trace.csv" | head -1)
  echo "== share $rk of $W (three repetitions in the totals)"; grep "rank" $R/gpurun_out/share_${W}_$rk.log | tail -3
  python3 $R/tools/share_timeline.py $f
  python3 $R/tools/kernel_durations.py $f ws_gram_kernel ws_gram_reduce ws_solve_kernel
  rm -rf $R/gpurun_out/share_${W}_$rk
done
