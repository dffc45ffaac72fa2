#!/bin/bash
# Round 5, first GPU pass (outputs under gpurun_out/r05a/): parity suite, fp32 power A/B (spc_powf vs ocml powf), config 5 and
# per-column-grid kernel timings on the shipped library, the default bench line with its new `config5` / `per_column_grid` keys.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05a
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests exit=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
{
for round in 1 2; do
  for lib in sp_coupler_amd/libspc_hip.so build/variants/libspc_ocmlpowf.so; do
    SPC_LIB=$R/$lib timeout -k 10 200 python tools/kbench.py --dtype f32 --sizes 35718 --levels 91,160 --tag "$(basename $lib .so)" | grep "n=" || exit 1
    SPC_LIB=$R/$lib timeout -k 10 200 python tools/kbench.py --dtype f32 --sizes 88838 --levels 137,512 --tag "$(basename $lib .so)" | grep "n=" || exit 1
  done
done
} > $O/kbench_f32_pow_ab.log 2>&1; echo "f32 A/B exit=$?"; cat $O/kbench_f32_pow_ab.log
{
timeout -k 10 200 python tools/kbench.py --dtype f64 --sizes 88838 --levels 137,512 --tag config5 | grep "n="
timeout -k 10 200 python tools/kbench.py --dtype f64 --sizes 1024,35718 --levels 91,160 --tag shared | grep "n="
timeout -k 10 200 python tools/kbench.py --dtype f64 --sizes 1024,35718 --levels 91,160 --per-column-grid --tag percol | grep "n="
} > $O/kbench_config5_percol.log 2>&1; echo "kbench exit=$?"; cat $O/kbench_config5_percol.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit=$?"; tail -5 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05a/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "verified", d.get("verified"))
for k in ("f64","f32"):
    c=d.get("config5",{}).get(k,{})
    print("config5", k, {x:c.get(x) for x in ("value","ms_per_step","k1_avg_launch_us","k3_avg_launch_us","k1_frac","k3_frac","verified","error")})
    if c.get("check",{}).get("failures"): print(c["check"]["failures"])
c=d.get("per_column_grid",{})
print("per_column_grid", {x:c.get(x) for x in ("value","k1_avg_launch_us","k3_avg_launch_us","k1_frac","k3_frac","k1_vs_shared_grid","k3_vs_shared_grid","verified","error")})
PY
echo "r05a done"
