#!/bin/bash
# A/B of the persistent recurrences' arrival-counter sharding (hooks build, LAED_GRU_SHARDS) on the script shapes
H=$PWD/neuraldialog-laed_amd/laed_amd/liblaed_hip_hooks.so
for w in P-fixed VST-fixed; do
  for n in 1 8 1 8 4 2; do
    LAED_HIP_LIB=$H LAED_GRU_SHARDS=$n timeout -k 10 120 python bench.py --workload $w --steps 300 --warmup 30 --cpu-sample 0 --no-alt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w shards=$n', d['ms_per_step'], 'timeouts', d.get('gru_persist_timeouts'))" || exit 1
  done
done
