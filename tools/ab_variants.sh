#!/bin/bash
# A/B of build variants (build/variants/libspc_*.so) with the kernel micro-benchmark, interleaved rounds
R=$GRAFT_REPO_ROOT
for round in 1 2; do
  for lib in $R/build/variants/libspc_*.so; do
    name=$(basename $lib .so)
    SPC_LIB=$lib timeout -k 10 120 python $R/tools/kbench.py --sizes ${SIZES:-1024,35718} --cbs ${CBS:-0} --tag "$name" 2>&1 | grep "n="
  done
done
