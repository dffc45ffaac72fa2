#!/bin/bash
# Round 5, seventh GPU pass (gpurun_out/r05g/): parity, K6 with the noise plane in registers (timing + stamps), fp32 / fp64 A/B against
# the previous commit.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05g
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests exit=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "^E " $O/gpu_tests.log | head -30; exit $rc; }
for lib in sp_coupler_amd/libspc_hip.so build/variants/libspc_head.so sp_coupler_amd/libspc_hip.so build/variants/libspc_head.so; do
  echo "== $lib"; SPC_LIB=$R/$lib timeout -k 10 300 python tools/kbench_aux.py --sizes "" --vn-cols 2,16,256 --vn-shapes 64x64x160,92x92x160 2>&1 | grep "K6"
done > $O/k6_ab.log 2>&1; echo "k6 A/B exit=$?"; cat $O/k6_ab.log
for n in 2 16; do timeout -k 10 200 python tools/stamps_k6.py $n 64 > $O/stamps_k6_$n.log 2>&1; echo "stamps k6 $n exit=$?"; grep -v amdgpu.ids $O/stamps_k6_$n.log; done
{
for round in 1 2; do
  for lib in sp_coupler_amd/libspc_hip.so build/variants/libspc_head.so; do
    SPC_LIB=$R/$lib timeout -k 10 200 python tools/kbench.py --dtype f32 --sizes 35718 --levels 91,160 --tag "$(basename $lib .so)" | grep "n=" || exit 1
    SPC_LIB=$R/$lib timeout -k 10 200 python tools/kbench.py --dtype f32 --sizes 88838 --levels 137,512 --tag "$(basename $lib .so)" | grep "n=" || exit 1
    SPC_LIB=$R/$lib timeout -k 10 200 python tools/kbench.py --dtype f64 --sizes 1024,35718 --levels 91,160 --tag "$(basename $lib .so)" | grep "n=" || exit 1
  done
done
} > $O/kbench_ab.log 2>&1; echo "A/B exit=$?"; grep -v amdgpu.ids $O/kbench_ab.log
echo "r05g done"
