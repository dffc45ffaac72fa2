#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05m
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "tests exit=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "^E " $O/gpu_tests.log | head -30; exit $rc; }
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit=$?"; tail -1 $O/smoke.log
/usr/bin/time -v timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench exit=$?"; grep "Elapsed (wall clock)\|Maximum resident" $O/bench_steps20.err
timeout -k 10 300 python bench.py --gpus 2 --device 0 --steps 5 --warmup 2 --cols 20000 > $O/bench_n2_rehearsal.json 2> $O/bench_n2_rehearsal.err; echo "bench N=2 rehearsal exit=$?"; tail -3 $O/bench_n2_rehearsal.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05m/bench_n2_rehearsal.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","n_gpus","backend","ranks","distinct_devices","verified","per_rank_ms")})
PY
