#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05n
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_dispatch_gpu.py tests/test_sputils_gpu.py -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "tests exit=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "^E " $O/gpu_tests.log | head -30; exit $rc; }
{
for round in 1 2; do
for lib in sp_coupler_amd/libspc_hip.so build/variants/libspc_f32vwaves1.so; do
SPC_LIB=$R/$lib timeout -k 10 300 python tools/kbench.py --dtype f32 --sizes 4096,35718,174264 --levels 91,160 --tag "$(basename $lib .so)" | grep "n="
SPC_LIB=$R/$lib timeout -k 10 300 python tools/kbench.py --dtype f32 --sizes 88838 --levels 137,512 --tag "$(basename $lib .so)" | grep "n="
done
done
} > $O/kbench_f32v_waves.log 2>&1; grep -v amdgpu.ids $O/kbench_f32v_waves.log
