#!/bin/bash
# EXPERIMENT: persistent GEMM blocks with cross-tile prefetch (hooks build, LAED_GEMM_PERSIST=1) against the one-tile-per-block
# form: correctness (the GEMM kernel tests with the variant switched on), then timing on the projection / script shapes.
set -u
O=gpurun_out/gemm_persist; rm -rf $O; mkdir -p $O
export LAED_HIP_LIB=$GRAFT_REPO_ROOT/neuraldialog-laed_amd/laed_amd/liblaed_hip_hooks.so
LAED_GEMM_PERSIST=1 timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q -k "sgemm or gemm" --timeout 200 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
SHAPES="40960,50000,1024,NT 9984,50000,1024,NT 8192,8192,8192,NT 50000,1024,40960,TN 1170,10004,512,NT 2340,10004,512,NT 10004,512,1170,TN 1170,512,10004,NN 1200,1536,200,NT 1536,512,1200,TN"
for i in 1 2; do
  for v in 0 1; do
    echo "== LAED_GEMM_PERSIST=$v (pass $i)" | tee -a $O/times.txt
    LAED_GEMM_PERSIST=$v timeout -k 10 300 python tools/gemm_shapes.py $SHAPES 2>&1 | tee -a $O/times.txt || exit 1
  done
done
