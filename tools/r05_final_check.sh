#!/bin/bash
# what the driver runs at round end, on the tree as it stands: the GPU suite, smoke(), the default bench line
set -o pipefail
O=gpurun_out/r05i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests exit=$rc"; tail -3 $O/gpu_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; rc=$?; echo "smoke exit=$rc"; tail -2 $O/smoke.log; [ $rc -eq 0 ] || exit 1
( time timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench exit=$?"; cat $O/bench.time
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05i/bench.json").read().strip().splitlines()[-1])
print("%.4g" % d["value"], "ms %.4f" % d["ms_per_step"], "K1 %.3f K3 %.3f" % (d["roofline"]["frac"], d["roofline"]["backward"]["frac"]), "traffic", d["roofline"].get("traffic"), "verified", d["verified"], "cpu", d["cpu_baseline"]["value"])
PY
