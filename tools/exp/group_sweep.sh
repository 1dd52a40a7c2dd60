#!/bin/bash
L=$GRAFT_REPO_ROOT/neuraldialog-laed_amd/laed_amd/liblaed_hip_hooks.so
for g in 8 2 4 16 32 64; do
  echo "GROUP_M=$g"; LAED_GEMM_GROUP_M=$g timeout -k 10 200 python tools/gemm_exp.py $L 2>&1 | tr "|" "\n" | grep "TN dW+ws\|NN dh M\|logits M\|TN dW K"
done
