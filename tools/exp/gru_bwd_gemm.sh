#!/bin/bash
# the GRU backward's per-step product dh_{t-1} = dgh_t W_hh (M = B, N = H, K = 3H, x W layout) under forced (tile, split-K) choices
H=$PWD/neuraldialog-laed_amd/laed_amd/liblaed_hip_hooks.so
for shape in 4096,1024,3072,NN 512,1024,3072,NN 2048,1024,3072,NN; do
  echo "== $shape auto"; LAED_HIP_LIB=$H timeout -k 10 60 python tools/gemm_shapes.py --ws $shape || exit 1
  for f in 128,1 128,2 128,3 128,4 128,6 64,1 64,2 64,3 64,4; do
    echo "-- force $f"; LAED_HIP_LIB=$H LAED_GEMM_FORCE=$f timeout -k 10 60 python tools/gemm_shapes.py --ws $shape || exit 1
  done
done
