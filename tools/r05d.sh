#!/bin/bash
# Round 5, fourth GPU pass (gpurun_out/r05d/): full parity suite (plan refresh, pitch bounds, per-device slow paths), mutation control.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05d
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests exit=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "^E " $O/gpu_tests.log | head -30; exit $rc; }
timeout -k 10 600 python tools/mutation_control.py > $O/mutation_control.log 2>&1; echo "mutation control exit=$?"; grep -v amdgpu.ids $O/mutation_control.log | tail -8
echo "r05d done"
