#!/bin/bash
# round 5: the N > 1 machinery of bench.py (RCCL process group, barrier, max-reduction, per-rank proof) on real hardware with
# ONE rank under torch.distributed.run (--force-dist); a rehearsal of the launch path, not a scaling measurement
set -o pipefail
O=gpurun_out/r05v; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-config5 --no-per-column-grid-leg --no-live-traffic > $O/bench_force_dist.json 2> $O/bench_force_dist.err
echo "force-dist exit=$?"; tail -3 $O/bench_force_dist.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05v/bench_force_dist.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','n_gpus','backend','ranks','force_dist','verified')})
print(d.get('per_rank_ms')); print(d['devices'])
PY
