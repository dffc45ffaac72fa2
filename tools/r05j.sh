#!/bin/bash
# fp32: columns per workgroup sweep (gpurun_out/r05j/)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
{
timeout -k 10 300 python tools/kbench.py --dtype f32 --sizes 35718 --levels 91,160 --cbs 0,1,2,4,8,16,0 --tag cbs | grep "n="
timeout -k 10 300 python tools/kbench.py --dtype f32 --sizes 88838 --levels 137,512 --cbs 0,1,2,4,8,0 --tag cbs | grep "n="
timeout -k 10 300 python tools/kbench.py --dtype f32 --sizes 1024,4096,348528 --levels 91,160 --cbs 0 --tag sizes | grep "n="
} > $O/kbench_f32_cbs.log 2>&1; echo "exit=$?"; grep -v amdgpu.ids $O/kbench_f32_cbs.log
