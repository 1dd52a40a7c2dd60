#!/bin/bash
# Projection chunk-size sweep (rows per [chunk][ldl] logits tile) at the per-GPU batch of the 8-GPU point and at B=4096.
#   gpurun --timeout 900 -- 'bash tools/chunk_sweep.sh'
set -u
O=gpurun_out/chunk_sweep; mkdir -p $O
A="--no-alt --cpu-sample 0 --steps 6 --warmup 2"
for c in 19968 9984 8192 6656 4992 2560 1280; do
  LAED_PROJ_CHUNK_ROWS=$c timeout -k 10 120 python bench.py --global-batch 512 $A > $O/b512_$c.json 2>> $O/err.log || exit 1
  python - "$O/b512_$c.json" $c <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("B=512  chunk %6s  %8.3f ms/step" % (sys.argv[2], d["ms_per_step"]), flush=True)
PY
done
for c in 40960 16384 8192 4096 2560 1280; do
  LAED_PROJ_CHUNK_ROWS=$c timeout -k 10 200 python bench.py $A > $O/b4096_$c.json 2>> $O/err.log || exit 1
  python - "$O/b4096_$c.json" $c <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("B=4096 chunk %6s  %8.3f ms/step" % (sys.argv[2], d["ms_per_step"]), flush=True)
PY
done
