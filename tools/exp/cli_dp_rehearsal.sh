#!/bin/bash
# Rehearsal of a driver script under a launcher on the one-GPU box (gloo transport, both ranks on cuda:0); output to gpurun_out/.
# usage (on the box): bash tools/exp/cli_dp_rehearsal.sh
set -u
out=gpurun_out/r3z/clidp
mkdir -p $out/data
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, "tests")
from helpers import load_golden
G = load_golden("corpora")
d = "gpurun_out/r3z/clidp/data"
for s in ("train", "valid", "test"):
    text = G["daily_dialog/inputs/%s/txt" % s].tobytes().decode("utf-8", errors="ignore").lower()
    utts = [u.strip() for line in text.splitlines() for u in line.split("__eou__") if u.strip()]
    open(os.path.join(d, "ptb.%s.txt" % s), "w", encoding="utf-8").write("".join(" %s \n" % u for u in utts))
PY
export LAED_DP_BACKEND=gloo LAED_WATCHDOG=60
timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  neuraldialog-laed_amd/ptb-utt.py --data_dir $out/data --log_dir $out/logs --embed_size 32 --dec_cell_size 64 --max_vocab_cnt 800 \
  --max_utt_len 20 --max_dec_len 20 --early_stop false --init_lr 0.005 --enc_cell_size 64 --y_size 4 --k 5 --batch_size 32 \
  --max_epoch 2 --ckpt_step 32 --print_step 16 > $out/run.log 2>&1
echo "rc $?" >> $out/run.log
tail -60 $out/run.log
