#!/bin/bash
# SQ counter pass (own run, --pmc only) for K1/K3 at 1024 and 35718 columns -> gpurun_out/sq_<n>/
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for n in 1024 35718; do
  rot=8; [ $n -gt 2000 ] && rot=2
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/sq_$n -- python3 $R/tools/pmc_run.py $n $rot > $R/gpurun_out/sq_$n.log 2>&1
  echo "sq $n exit=$?"
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for n in (1024, 35718):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/sq_%d/**/*counter_collection.csv" % n, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_forward" in k or "k_backward" in k:
                agg["K1" if "k_forward" in k else "K3"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kern, cs in agg.items():
        m = {c: sum(v[len(v)//3:]) / max(1, len(v[len(v)//3:])) for c, v in cs.items()}
        wc = m.get("SQ_WAVE_CYCLES", 1)
        print("n=%d %s: " % (n, kern) + "  ".join("%s=%.3g" % (c, v) for c, v in sorted(m.items())))
        print("     wait_any %.0f%%  wait_inst %.0f%%  active_inst %.0f%%  lds_conflict/lds_active %.1f%%  valu/salu %.2f" % (
            100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, m.get("SQ_LDS_IDX_ACTIVE", 1)), m.get("SQ_INSTS_VALU", 0) / max(1, m.get("SQ_INSTS_SALU", 1))))
PY
