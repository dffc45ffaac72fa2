#!/bin/bash
# Round 5, second GPU pass (gpurun_out/r05b/): vector-instruction issue costs, where the fp32 kernels' time goes (diagnostic builds
# without divisions / without pow), the bench line with its legs checked.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05b
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 120 python tools/issue_rate.py > $O/issue_rate.log 2>&1; echo "issue exit=$?"; cat $O/issue_rate.log
{
for round in 1 2; do
  for lib in sp_coupler_amd/libspc_hip.so build/variants/libspc_exp1.so build/variants/libspc_exp3.so; do
    for dt in f32 f64; do
    SPC_LIB=$R/$lib timeout -k 10 200 python tools/kbench.py --dtype $dt --sizes 35718 --levels 91,160 --tag "$(basename $lib .so)" | grep "n=" || exit 1
    done
  done
done
} > $O/kbench_f32_exp.log 2>&1; echo "exp A/B exit=$?"; grep -v amdgpu.ids $O/kbench_f32_exp.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit=$?"; tail -5 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05b/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "verified", d.get("verified"))
for k in ("f64","f32"):
    c=d.get("config5",{}).get(k,{})
    print("config5", k, {x:c.get(x) for x in ("value","ms_per_step","k1_avg_launch_us","k3_avg_launch_us","k1_frac","k3_frac","verified","error")})
    print(json.dumps(c.get("check")))
c=d.get("per_column_grid",{})
print("per_column_grid", {x:c.get(x) for x in ("value","k1_avg_launch_us","k3_avg_launch_us","k1_frac","k3_frac","k1_vs_shared_grid","k3_vs_shared_grid","verified","error")})
print(json.dumps(c.get("check")))
PY
echo "r05b done"
