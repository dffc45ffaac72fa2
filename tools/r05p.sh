#!/bin/bash
# config 5 (137 <-> 512) fp64: columns per workgroup and prologue-prefetch sweep of K1 / K3 (gpurun_out/r05p/)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
mkdir -p $O
cd $R
{
for round in 1 2; do
timeout -k 10 300 python tools/kbench.py --dtype f64 --sizes 88838 --levels 137,512 --cbs 0,1,2,4 --tag "default" | grep "n="
SPC_K3_PRE=0 SPC_K1_PRE=1 timeout -k 10 300 python tools/kbench.py --dtype f64 --sizes 88838 --levels 137,512 --cbs 0 --tag "k3pre=0,k1pre=1" | grep "n="
done
} > $O/kbench_cfg5.log 2>&1; grep -v amdgpu.ids $O/kbench_cfg5.log
