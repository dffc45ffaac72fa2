#!/bin/bash
# Round-4 second K7 pass, evidence on one GPU box (outputs under gpurun_out/r04b/): parity tests, warm timings, rocprofv3 kernel stats
# of the micro-benchmark, the counter passes.   usage: tools/r04b_evidence.sh
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests exit=$?"; tail -1 $O/gpu_tests.log
timeout -k 10 300 python tools/kbench_aux.py --sizes 1024,35718 --sputils --vn-cols 2,16 > $O/kbench_aux.log 2>&1; echo "kbench_aux exit=$?"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_aux -- python3 $R/tools/kbench_aux.py --sizes 35718 --sputils --vn-cols 2 > $O/kbench_aux_under_rocprof.log 2>&1; echo "rocprof aux exit=$?"
cd $R
bash tools/k7_profile.sh r04b_final > $O/k7_profile.out 2>&1; echo "k7 profile exit=$?"
echo "r04b evidence done"
