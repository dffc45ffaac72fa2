#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05k
mkdir -p $O
cd $R
timeout -k 10 120 python tools/copy_width.py > $O/copy_width.log 2>&1; echo "copy width exit=$?"; grep -v amdgpu.ids $O/copy_width.log
timeout -k 10 600 python -m pytest tests/test_dispatch_gpu.py tests/test_parity_gpu.py -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/gpu_tests.log
{
for round in 1 2; do
timeout -k 10 300 python tools/kbench.py --dtype f32 --sizes 35718 --levels 91,160 --cbs 0,2,4 --tag cbs | grep "n="
timeout -k 10 300 python tools/kbench.py --dtype f32 --sizes 8192,174264 --levels 91,160 --cbs 0,4 --tag cbs | grep "n="
done
} > $O/kbench_f32.log 2>&1; grep -v amdgpu.ids $O/kbench_f32.log
