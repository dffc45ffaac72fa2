#!/bin/bash
# one GPU-box visit: parity tests, bench, rocprofv3 kernel stats (outputs under gpurun_out/)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $R/gpurun_out/tests.log 2>&1
echo "tests exit=$?" >> $R/gpurun_out/tests.log
tail -5 $R/gpurun_out/tests.log
timeout -k 10 200 python bench.py --steps 2000 --warmup 200 > $R/gpurun_out/bench.log 2>&1 && tail -2 $R/gpurun_out/bench.log
export TMPDIR=/tmp
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 500 --warmup 50 --cpu-seconds 0 > $R/gpurun_out/prof.log 2>&1
echo "rocprof exit=$?"
find $R/gpurun_out/prof -name '*stats*' | head
