#!/bin/bash
# A/B of half-filled row groups in the persistent backward recurrence (hooks build, LAED_GRU_HALF_GROUPS)
H=$PWD/neuraldialog-laed_amd/laed_amd/liblaed_hip_hooks.so
for w in P-fixed P-varlen; do
  for n in 0 1 0 1; do
    LAED_HIP_LIB=$H LAED_GRU_HALF_GROUPS=$n timeout -k 10 120 python bench.py --workload $w --steps 300 --warmup 30 --cpu-sample 0 --no-alt --no-prof 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w half=$n', d['ms_per_step'], 'nll', d['final_nll'], 'timeouts', d.get('gru_persist_timeouts'))" || exit 1
  done
done
