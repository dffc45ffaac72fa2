set -o pipefail
O=gpurun_out/r04c; mkdir -p $O
SPC_FUZZ_TRIALS=800 SPC_FUZZ_SEED=505 timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_sputils_gpu.py -m gpu -q -k 'random or fuzz' > $O/soak.log 2>&1; echo "soak exit=$?"; tail -1 $O/soak.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit=$?"; tail -1 $O/smoke.log
timeout -k 10 500 python bench.py --multi-devices 0,0 > $O/bench.json 2> $O/bench.err; echo "bench exit=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench steps20 exit=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r04c/bench.json","gpurun_out/r04c/bench_steps20.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("verified"), d.get("dropin",{}).get("verified"), {k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.get("dropin",{}).items() if k!="verified"} )
PY
