#!/bin/bash
# A/B of workgroup sizes for the first-generation kernels at small batches: build/variants/libspc_b{256,512,1024}.so
# (tools/exp_variants.sh "b512:-DSPC_BLOCK=512" ...), explicit columns per workgroup, interleaved rounds
R=$GRAFT_REPO_ROOT
for round in 1 2; do
  for spec in ${SPECS:-b256:1,2 b512:1,2,4 b1024:2,4}; do
    name=${spec%%:*}; cbs=${spec#*:}
    SPC_LIB=$R/build/variants/libspc_$name.so timeout -k 10 120 python $R/tools/kbench.py --sizes ${SIZES:-1024,2048,4096} --cbs $cbs --tag "$name" 2>&1 | grep "n="
  done
done
