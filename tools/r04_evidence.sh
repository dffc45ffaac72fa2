#!/bin/bash
# Round-4 evidence on one GPU box (outputs under gpurun_out/r04/): bench lines, rocprofv3 kernel stats of bench.py and of the
# K4 / K7 micro-benchmark, PMC traffic of K1 / K3 / K4, SQ counters of K4, the K7 counter passes, size sweeps.
# usage: tools/r04_evidence.sh [steps...]   steps: bench prof aux pmc sq k7 kbench rehearsal (default: all)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
STEPS=${@:-bench prof aux pmc sq k7 kbench rehearsal}
has() { [[ " $STEPS " == *" $1 "* ]]; }
cd $R
if has bench; then
  timeout -k 10 500 python bench.py --multi-devices 0,0 > $O/bench.json 2> $O/bench.err; echo "bench exit=$?"
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench steps20 exit=$?"
fi
if has prof; then
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-anchor --no-dropin --multi-devices none > $O/bench_under_rocprof.json 2> $O/prof_bench.err; echo "rocprof bench exit=$?"
  cd $R
fi
if has aux; then
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_aux -- python3 $R/tools/kbench_aux.py --sizes 35718 --sputils --vn-cols 2 > $O/kbench_aux_under_rocprof.log 2>&1; echo "rocprof aux exit=$?"
  cd $R
  timeout -k 10 300 python tools/kbench_aux.py --sizes 1024,35718 --sputils --vn-cols 2,16 > $O/kbench_aux.log 2>&1; echo "kbench_aux exit=$?"
fi
if has pmc; then
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/pmc_run.py 35718 2 k4 > $O/pmc_fetch.log 2>&1; echo "pmc fetch exit=$?"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/pmc_run.py 35718 2 k4 > $O/pmc_write.log 2>&1; echo "pmc write exit=$?"
  cd $R && PMC_TAG="round 4" python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 35718 268435456 $O/traffic.json > $O/pmc_summary.log 2>&1
  grep -E "hbm_bytes_per_launch|hbm_bytes_per_column" $O/pmc_summary.log
fi
if has sq; then
  sed -e "s#gpurun_out/sq_k4b#gpurun_out/r04/sq_k4b#g" -e "s#gpurun_out/sq_k4#gpurun_out/r04/sq_k4#g" -e 's#"gpurun_out/%s/#"gpurun_out/r04/%s/#' tools/sq_counters_k4.sh > /tmp/sq_k4_r04.sh
  bash /tmp/sq_k4_r04.sh > $O/sq_counters_k4.log 2>&1; echo "sq k4 exit=$?"
fi
if has k7; then
  bash tools/k7_profile.sh r04_final > $O/k7_profile.out 2>&1; echo "k7 profile exit=$?"
fi
if has kbench; then
  timeout -k 10 500 python tools/kbench.py --sizes 1024,2048,4096,8192,16384,35718,43566,87132,174264,348528 --cbs 0 > $O/kbench_sizes.log 2>&1; grep n= $O/kbench_sizes.log
fi
if has rehearsal; then
  timeout -k 10 300 python bench.py --gpus 2 --device 0 --cols 87132 --steps 20 --warmup 5 > $O/bench_n2_rehearsal.json 2> $O/bench_n2_rehearsal.err; echo "n2 rehearsal exit=$?"
fi
echo "r04 evidence done"
