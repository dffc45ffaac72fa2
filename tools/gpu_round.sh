#!/bin/bash
# one GPU-box visit: parity tests, bench, rocprofv3 kernel stats and HBM PMC passes (outputs under gpurun_out/)
# usage: tools/gpu_round.sh TAG [steps...]   steps: tests bench prof pmc kbench  (default: all)
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
shift
STEPS=${@:-tests bench prof pmc kbench}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
has() { [[ " $STEPS " == *" $1 "* ]]; }
if has tests; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/tests_$TAG.log 2>&1
  echo "tests exit=$?" >> $R/gpurun_out/tests_$TAG.log
  tail -3 $R/gpurun_out/tests_$TAG.log
fi
if has bench; then
  timeout -k 10 400 python bench.py > $R/gpurun_out/bench_$TAG.log 2>&1
  echo "bench exit=$?"; tail -1 $R/gpurun_out/bench_$TAG.log | cut -c1-1500
fi
if has prof; then
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-anchor > $R/gpurun_out/prof_$TAG.log 2>&1
  # (--no-anchor: `--stats` averages per kernel NAME and the anchor launches the same K1 / K3 instantiation on 10x the columns;
  #  tools/trace_by_grid.py splits a full run's trace by grid size instead)
  echo "rocprof stats exit=$?"
  cd $R
fi
if has pmc; then
  cd /tmp
  N=${PMC_COLS:-35718}; ROT=${PMC_ROT:-2}
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python3 $R/tools/pmc_run.py $N $ROT > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
  echo "pmc fetch exit=$?"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python3 $R/tools/pmc_run.py $N $ROT > $R/gpurun_out/pmc_write_$TAG.log 2>&1
  echo "pmc write exit=$?"
  cd $R && python tools/pmc_summary.py gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG $N 268435456 gpurun_out/traffic_$TAG.json > $R/gpurun_out/pmc_summary_$TAG.log 2>&1
  tail -30 $R/gpurun_out/pmc_summary_$TAG.log
fi
if has kbench; then
  timeout -k 10 400 python tools/kbench.py --sizes ${KB_SIZES:-1024,2048,4096,8192,16384,35718,43566,348528} --cbs 0 > $R/gpurun_out/kbench_sizes_$TAG.log 2>&1
  grep n= $R/gpurun_out/kbench_sizes_$TAG.log
fi
