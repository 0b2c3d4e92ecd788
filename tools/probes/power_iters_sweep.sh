# the model solver's power steps for lambda_max of a new Gram (SLM_WS_POWER_ITERS): the headline, config 3, the law's draws, soaks
sum_ms() { awk '{for(i=1;i<=NF;i++) if($i=="ms" || $i=="ms,") {s+=$(i-1); break}} /passes/ {for(i=1;i<=NF;i++) if($i=="passes" || $i=="passes,") {p+=$(i-1); break}} END {printf "total %.2f ms, %d passes\n", s, p}'; }
for it in 10 6 4 3; do
  echo "== SLM_WS_POWER_ITERS=$it"
  SLM_WS_POWER_ITERS=$it python tools/lanes_sweep.py 0 2>&1 | tail -n 1
  SLM_WS_POWER_ITERS=$it python tools/config3_lanes.py 0 2>&1 | tail -n 1
  SLM_WS_POWER_ITERS=$it python tools/headline_data_seeds.py 2>&1 | grep "lanes=18"
  SLM_WS_POWER_ITERS=$it python tools/headline_soak.py 24 2>&1 | grep "^seed" | sum_ms
  SLM_WS_POWER_ITERS=$it python tools/group_soak.py 12 2>&1 | grep "^seed" | sum_ms
done
