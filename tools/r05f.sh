#!/bin/bash
# Round 5, sixth GPU pass (gpurun_out/r05f/): parity of the restructured K1 / K3 / K5, A/B of the fp32 changes (quotients through
# fp64, two work items in flight) and of the double kernels against the previous commit, K5 / K4 / K7 timings, the K6 round stamps.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests exit=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "^E " $O/gpu_tests.log | head -30; exit $rc; }
{
for round in 1 2; do
  for lib in sp_coupler_amd/libspc_hip.so build/variants/libspc_head.so build/variants/libspc_f32ieeediv.so build/variants/libspc_f32unroll1.so; do
    SPC_LIB=$R/$lib timeout -k 10 200 python tools/kbench.py --dtype f32 --sizes 35718 --levels 91,160 --tag "$(basename $lib .so)" | grep "n=" || exit 1
    SPC_LIB=$R/$lib timeout -k 10 200 python tools/kbench.py --dtype f32 --sizes 88838 --levels 137,512 --tag "$(basename $lib .so)" | grep "n=" || exit 1
  done
  for lib in sp_coupler_amd/libspc_hip.so build/variants/libspc_head.so; do
    SPC_LIB=$R/$lib timeout -k 10 200 python tools/kbench.py --dtype f64 --sizes 1024,35718 --levels 91,160 --tag "$(basename $lib .so)" | grep "n=" || exit 1
    SPC_LIB=$R/$lib timeout -k 10 200 python tools/kbench.py --dtype f64 --sizes 88838 --levels 137,512 --tag "$(basename $lib .so)" | grep "n=" || exit 1
  done
done
} > $O/kbench_ab.log 2>&1; echo "A/B exit=$?"; grep -v amdgpu.ids $O/kbench_ab.log
timeout -k 10 300 python tools/kbench_aux.py --sizes 1024,35718 --vn-cols 2,16 > $O/kbench_aux.log 2>&1; echo "kbench_aux exit=$?"; grep -v amdgpu.ids $O/kbench_aux.log
SPC_LIB=$R/build/variants/libspc_head.so timeout -k 10 300 python tools/kbench_aux.py --sizes 1024,35718 --vn-cols "" > $O/kbench_aux_head.log 2>&1; echo "kbench_aux head exit=$?"; grep -v amdgpu.ids $O/kbench_aux_head.log
for n in 2 16; do timeout -k 10 200 python tools/stamps_k6.py $n 64 > $O/stamps_k6_$n.log 2>&1; echo "stamps k6 $n exit=$?"; grep -v amdgpu.ids $O/stamps_k6_$n.log; done
echo "r05f done"
