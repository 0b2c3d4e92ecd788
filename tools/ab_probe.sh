#!/bin/bash
# A/B of engine variants on ONE box: alternating runs of bench.py (headline only) under an environment switch.
# usage: tools/ab_probe.sh OUTDIR VAR   (runs VAR=0 and the default three times each, alternating)
O=$1; V=$2; mkdir -p $O
for k in 1 2 3; do
  env $V=0 python bench.py --steps 20 --warmup 3 --cpu-budget 0 --no-extra > $O/${V}_off_$k.log 2>/dev/null
  python bench.py --steps 20 --warmup 3 --cpu-budget 0 --no-extra > $O/${V}_on_$k.log 2>/dev/null
done
python tools/benchline.py $O/${V}_off_*.log $O/${V}_on_*.log
