#!/bin/bash
# one GPU-box visit: parity tests, bench, rocprofv3 kernel stats and HBM PMC passes (outputs under gpurun_out/)
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
mkdir -p $R/gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $R/gpurun_out/tests.log 2>&1
echo "tests exit=$?" >> $R/gpurun_out/tests.log
tail -3 $R/gpurun_out/tests.log
timeout -k 10 200 python bench.py --steps 2000 --warmup 200 > $R/gpurun_out/bench_$TAG.log 2>&1 && tail -1 $R/gpurun_out/bench_$TAG.log
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 500 --warmup 50 --cpu-seconds 0 > $R/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof stats exit=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python3 $R/tools/pmc_run.py 1024 8 > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
echo "pmc fetch exit=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python3 $R/tools/pmc_run.py 1024 8 > $R/gpurun_out/pmc_write_$TAG.log 2>&1
echo "pmc write exit=$?"
cd $R && python tools/pmc_summary.py gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG 1024 268435456 gpurun_out/traffic_$TAG.json > $R/gpurun_out/pmc_summary_$TAG.log 2>&1
tail -30 $R/gpurun_out/pmc_summary_$TAG.log
timeout -k 10 300 python tools/kbench.py --sizes 1024,2048,4096,8192,16384,35718,348528 --cbs 0 > $R/gpurun_out/kbench_sizes_$TAG.log 2>&1
grep n= $R/gpurun_out/kbench_sizes_$TAG.log
