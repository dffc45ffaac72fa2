#!/bin/bash
# round 5: the two new semantic properties (surface fluxes, variability nudge) on the kernels + the mutation control with 21 mutants
set -o pipefail
O=gpurun_out/r05z; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_semantic_gpu.py tests/test_vnudge.py -m gpu -x -q > $O/semantic_gpu.log 2>&1; rc=$?; echo "semantic gpu exit=$rc"; tail -15 $O/semantic_gpu.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python tools/mutation_control.py > $O/mutation_control.log 2>&1; echo "mutation control exit=$?"; tail -30 $O/mutation_control.log
