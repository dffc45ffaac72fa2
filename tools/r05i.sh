#!/bin/bash
# Round 5, ninth GPU pass (gpurun_out/r05i/): full parity suite on the final kernels, K6 timings / stamps, aux kernel timings.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05i
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests exit=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "^E " $O/gpu_tests.log | head -30; exit $rc; }
for rc_ in 1 0 1 0; do export SPC_VN_TREE_SHFL=$rc_;
  echo "== SPC_VN_RCACHE=$rc_ SPC_VN_TREE_SHFL=$rc_"; SPC_VN_RCACHE=$rc_ timeout -k 10 300 python tools/kbench_aux.py --sizes "" --vn-cols 2,16,256 --vn-shapes 64x64x160,92x92x160,128x128x160 2>&1 | grep "K6"
done > $O/k6_ab.log 2>&1; echo "k6 A/B exit=$?"; cat $O/k6_ab.log
unset SPC_VN_TREE_SHFL
for n in 2 16; do echo "== n=$n"; timeout -k 10 200 python tools/stamps_k6.py $n 64 2>&1 | grep -v amdgpu.ids; done > $O/stamps_k6.log 2>&1; echo "stamps exit=$?"; cat $O/stamps_k6.log
timeout -k 10 300 python tools/kbench_aux.py --sizes 1024,35718 --sputils --vn-cols "" > $O/kbench_aux.log 2>&1; echo "kbench_aux exit=$?"; grep -v amdgpu.ids $O/kbench_aux.log
echo "r05i done"
