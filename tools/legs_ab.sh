#!/bin/bash
# A/B of environment settings on the bench's other legs: usage legs_ab.sh "SLM_X=1" "SLM_Y=2 SLM_Z=3" ...  ("" = defaults)
L=soak,config3_path,config4_grid,config4_grid_dense_regime,rowshard,literal_config2
i=0
for s in "$@"; do
  i=$((i+1))
  env $s python bench.py --cpu-budget 0 --legs $L > gpurun_out/legs_$i.log 2>gpurun_out/legs_$i.err
  echo "== [$s]"
  python tools/show_legs.py gpurun_out/legs_$i.log | python -c "
import sys,json
for line in sys.stdin:
    k,_,v=line.partition(' ')
    if k=='soak':
        j=json.loads(v); print('soak', round(j['median_fits_per_s']), round(j['worst_fits_per_s']), j['passes'], j['ms'])
    elif k=='config4_grid':
        j=json.loads(v); print('c4', j['seconds_per_grid'], j['passes_per_rank'], j['covariance']['seconds_per_grid'])
    elif k=='config4_grid_dense_regime':
        j=json.loads(v); print('c4dense', j['seconds_per_grid'], j['passes'], j['covariance']['seconds_per_grid'])
    elif k=='config3_path':
        j=json.loads(v); print('c3', j['ms_per_path'], j['passes'])
    elif k=='literal_config2':
        j=json.loads(v); print('literal', j['ms_per_path'], j['passes'])
    elif k=='rowshard':
        j=json.loads(v); print('rowshard', j['seconds_per_fit'], j['passes'])
    elif k=='config4_grid_emulated_world8':
        j=json.loads(v); print('c4 emu', j['speedup_full_over_max_share'], j['share_passes'])
    elif k.startswith('{'):
        print(k, v[:30])
"
done
