#!/bin/bash
# tools/general_time.py for every variant library ("base" = in-tree), twice
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset SHW_LIB_PATH; else export SHW_LIB_PATH=$PWD/gpurun_variants/libshw_hip_$v.so; fi
  echo "== $v"; python tools/general_time.py 2>/dev/null | grep "weighted=False" | cut -c1-90
done; done
