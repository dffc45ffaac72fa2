#!/bin/bash
# Round 5 soak (gpurun_out/r05q/): the fuzz tests with more trials and another seed, on the final library
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05q
mkdir -p $O
cd $R
SPC_FUZZ_TRIALS=600 SPC_FUZZ_SEED=515 timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_sputils_gpu.py -m gpu -q -k 'random or fuzz' > $O/soak.log 2>&1; echo "soak exit=$?"; tail -3 $O/soak.log
