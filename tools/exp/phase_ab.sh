#!/bin/bash
# A/B of the GEMM phase lock (hooks build): LAED_GEMM_PHASE = K tiles between meetings (0 = off)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3
mkdir -p $O
L=$PWD/neuraldialog-laed_amd/laed_amd/liblaed_hip_hooks.so
for ph in 0 8; do
  LAED_HIP_LIB=$L LAED_GEMM_PHASE=$ph timeout -k 10 120 python tools/exp/phase_check.py >> $O/phase_check.txt 2>&1 || exit 1
done
for ph in 0 4 8 16 32 64 0; do
  echo "PHASE=$ph" >> $O/phase_ab.txt
  LAED_GEMM_PHASE=$ph timeout -k 10 200 python tools/gemm_exp.py $L >> $O/phase_ab.txt 2>&1 || exit 1
done
