#!/bin/bash
# Round 5, fifth GPU pass (gpurun_out/r05e/): parity of the new K5, K5 / K4 / K7 timings, the K6 round stamps.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05e
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests exit=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "^E " $O/gpu_tests.log | head -30; exit $rc; }
timeout -k 10 300 python tools/kbench_aux.py --sizes 1024,35718 --sputils --vn-cols 2,16 > $O/kbench_aux.log 2>&1; echo "kbench_aux exit=$?"; grep -v amdgpu.ids $O/kbench_aux.log
timeout -k 10 300 python tools/kbench_aux.py --sizes 88838 --levels 137,512 --vn-cols "" > $O/kbench_aux_config5.log 2>&1; echo "kbench_aux cfg5 exit=$?"; grep -v amdgpu.ids $O/kbench_aux_config5.log
for n in 2 16; do timeout -k 10 200 python tools/stamps_k6.py $n 64 > $O/stamps_k6_$n.log 2>&1; echo "stamps k6 $n exit=$?"; grep -v amdgpu.ids $O/stamps_k6_$n.log; done
echo "r05e done"
