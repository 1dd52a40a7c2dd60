#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3
mkdir -p $O
L=$PWD/neuraldialog-laed_amd/laed_amd/liblaed_hip_hooks.so
for cfg in "0 64" "8 32" "16 32" "32 32" "64 32" "16 16" "32 16" "0 64"; do
  set -- $cfg
  echo "PHASE=$1 GROUP=$2" >> $O/phase_ab2.txt
  LAED_GEMM_PHASE=$1 LAED_GEMM_PHASE_GROUP=$2 timeout -k 10 200 python tools/gemm_exp.py $L 2>&1 | grep -v amdgpu.ids | sed 's/ | /\n  /g' | grep -E "NN dh M=40960|dW\+ws|NT square" >> $O/phase_ab2.txt || exit 1
done
