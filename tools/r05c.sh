#!/bin/bash
# Round 5, third GPU pass (gpurun_out/r05c/): full parity suite incl. the semantic properties, the mutation control, rocprofv3
# kernel stats of the bench command (config 5 legs, per-column grid), PMC traffic of the per-column-grid kernels.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests exit=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/mutation_control.py > $O/mutation_control.log 2>&1; echo "mutation control exit=$?"; cat $O/mutation_control.log | grep -v amdgpu.ids
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-anchor --no-dropin --no-small-batch --multi-devices none > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "rocprof bench exit=$?"
for mode in percol ""; do
  tag=${mode:-shared}
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$tag -- python3 $R/tools/pmc_run.py 35718 2 $mode > $O/pmc_fetch_$tag.log 2>&1; echo "pmc fetch $tag exit=$?"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$tag -- python3 $R/tools/pmc_run.py 35718 2 $mode > $O/pmc_write_$tag.log 2>&1; echo "pmc write $tag exit=$?"
  (cd $R && PMC_TAG="round 5, $tag LES grid" python tools/pmc_summary.py $O/pmc_fetch_$tag $O/pmc_write_$tag 35718 268435456 $O/traffic_$tag.json > $O/pmc_summary_$tag.log 2>&1; grep "bytes_per_launch\|bytes_per_column" $O/pmc_summary_$tag.log)
done
cd $R
find $O/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
head -12 $O/bench_kernel_stats.csv | cut -c1-200
echo "r05c done"
