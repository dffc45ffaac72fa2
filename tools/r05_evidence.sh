#!/bin/bash
# Round 5 evidence on one GPU box (gpurun_out/r05f2/): the default bench line, the driver-style line (--steps 20 --warmup 5), the
# --dtype f32 and --config 5 lines, rocprofv3 kernel stats of the bench command, PMC traffic (shared grid: profiles/traffic.json),
# kernel sizes sweep.   usage: tools/r05_evidence.sh
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f2
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 600 python bench.py --multi-devices 0,0 > $O/bench.json 2> $O/bench.err; echo "bench exit=$?"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench steps20 exit=$?"
timeout -k 10 300 python bench.py --dtype f32 --no-dropin --no-anchor --no-small-batch --multi-devices none > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 exit=$?"
timeout -k 10 300 python bench.py --config 5 --no-dropin --no-anchor --no-small-batch --multi-devices none --steps 200 --warmup 10 > $O/bench_config5.json 2> $O/bench_config5.err; echo "bench config5 exit=$?"
timeout -k 10 300 python bench.py --per-column-grid --no-dropin --no-anchor --no-small-batch --multi-devices none > $O/bench_percol.json 2> $O/bench_percol.err; echo "bench percol exit=$?"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-anchor --no-dropin --no-small-batch --no-live-traffic --multi-devices none > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "rocprof bench exit=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/pmc_run.py 35718 2 > $O/pmc_fetch.log 2>&1; echo "pmc fetch exit=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/pmc_run.py 35718 2 > $O/pmc_write.log 2>&1; echo "pmc write exit=$?"
cd $R
PMC_TAG="round 5" python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 35718 268435456 $O/traffic.json > $O/pmc_summary.log 2>&1; grep "bytes_per_launch\"" $O/pmc_summary.log
find $O/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
head -9 $O/bench_kernel_stats.csv | cut -c1-160
timeout -k 10 400 python tools/kbench.py --sizes 1024,2048,4096,8192,16384,35718,43566,87132,174264,348528 --cbs 0 > $O/kbench_sizes.log 2>&1; grep n= $O/kbench_sizes.log
python - <<'PY'
import json
for f in ("bench.json","bench_steps20.json","bench_f32.json","bench_config5.json","bench_percol.json"):
    d=json.loads(open("gpurun_out/r05f2/"+f).read().strip().splitlines()[-1])
    print(f, "%.4g" % d["value"], "ms %.4f" % d["ms_per_step"], d["dtype"], "K1 frac %.3f" % d["roofline"]["frac"], "K3 frac %.3f" % d["roofline"]["backward"]["frac"], "verified", d.get("verified"),
          "dropin", d.get("dropin",{}).get("verified"), {k:("%.3g" % v["value"] if isinstance(v,dict) and "value" in v else None) for k,v in d.get("dropin",{}).items() if isinstance(v,dict)})
    for k in ("f64","f32"):
        c=d.get("config5",{}).get(k)
        if c: print("   config5", k, {x:(round(c[x],3) if isinstance(c.get(x),float) else c.get(x)) for x in ("value","k1_avg_launch_us","k3_avg_launch_us","k1_frac","k3_frac","verified","error")})
    c=d.get("per_column_grid")
    if c: print("   per_column_grid", {x:(round(c[x],3) if isinstance(c.get(x),float) else c.get(x)) for x in ("value","k1_avg_launch_us","k3_avg_launch_us","k1_frac","k3_frac","k1_vs_shared_grid","k3_vs_shared_grid","verified","error")})
    if "small_batch" in d: print("   small_batch", {x:round(d["small_batch"][x],3) for x in ("value","k1_avg_launch_us","k3_avg_launch_us","k1_frac","k3_frac")})
PY
echo "r05 evidence done"
