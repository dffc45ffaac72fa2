#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05l
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_dispatch_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "tests exit=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "^E " $O/gpu_tests.log | head -30; exit $rc; }
{
for round in 1 2; do
for v in 1 0; do
SPC_F32_VEC=$v timeout -k 10 300 python tools/kbench.py --dtype f32 --sizes 4096,35718,174264 --levels 91,160 --tag "vec=$v" | grep "n="
SPC_F32_VEC=$v timeout -k 10 300 python tools/kbench.py --dtype f32 --sizes 88838 --levels 137,512 --tag "vec=$v" | grep "n="
done
done
} > $O/kbench_f32_vec.log 2>&1; grep -v amdgpu.ids $O/kbench_f32_vec.log
timeout -k 10 120 python tools/copy_width.py > $O/copy_width.log 2>&1; grep -v amdgpu.ids $O/copy_width.log | head -5
