#!/bin/bash
# fused GRU step kernel: block geometries at H = 1024 (us per step incl. the hoisted input GEMM's share)
for wm in ${GEOMS:-auto 2 4 42 22}; do
  if [ $wm = auto ]; then unset LAED_GRU_WM; else export LAED_GRU_WM=$wm; fi
  timeout -k 10 120 python tools/gru_geom.py || exit 1
done
