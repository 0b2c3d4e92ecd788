#!/bin/bash
# A/B of an environment setting on ONE box: alternating runs of bench.py (headline only).
# usage: tools/ab_env.sh OUTDIR NAME=VALUE   (default vs the setting, three runs each, alternating)
O=$1; KV=$2; T=$(echo $KV | tr '=' '_'); mkdir -p $O
for k in 1 2 3; do
  python bench.py --steps 20 --warmup 3 --cpu-budget 0 --no-extra > $O/${T}_default_$k.log 2>/dev/null
  env $KV python bench.py --steps 20 --warmup 3 --cpu-budget 0 --no-extra > $O/${T}_set_$k.log 2>/dev/null
done
python tools/benchline.py $O/${T}_default_*.log $O/${T}_set_*.log
