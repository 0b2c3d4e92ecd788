#!/bin/bash
# rocprofv3 kernel trace of any tool script, every kernel of the run listed with start, gap and duration (the last N kernels):
# usage: trace_any.sh N SCRIPT [ARGS ...]      (run on the GPU box from the repository root; output under gpurun_out/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=$1; shift
rm -rf $R/gpurun_out/anyprof
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/anyprof -o t -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/anyprof.log 2>&1
f=$(find $R/gpurun_out/anyprof -name "*kernel_trace.csv" | head -1)
tail -3 $R/gpurun_out/anyprof.log | cut -c1-300
python3 - "$f" "$N" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("slm::", "").replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
seq = seq[-int(sys.argv[2]):]
t0 = seq[0][1]; prev = None
for nm, a, b in seq:
    print(f"{(a - t0) / 1e3:9.1f} us  gap {((a - prev) / 1e3 if prev else 0):7.1f}  {(b - a) / 1e3:8.1f} us  {nm[:60]}")
    prev = b
PY
rm -rf $R/gpurun_out/anyprof
