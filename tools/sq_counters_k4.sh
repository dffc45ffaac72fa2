#!/bin/bash
# SQ counter pass (own run, --pmc only) for K3 and K4 at 35718 columns -> gpurun_out/sq_k4/
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/sq_k4 -- python3 $R/tools/pmc_run.py 35718 2 k4 > $R/gpurun_out/sq_k4.log 2>&1
echo "sq k4 exit=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --output-format csv -d $R/gpurun_out/sq_k4b -- python3 $R/tools/pmc_run.py 35718 2 k4 > $R/gpurun_out/sq_k4b.log 2>&1
echo "sq k4b exit=$?"
cd $R && python3 - <<'PY'
import csv, glob, collections
for d in ("sq_k4", "sq_k4b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            name = "K4" if "k_backward_cons" in k else ("K3" if "k_backward" in k else ("K1" if "k_forward" in k else None))
            if name:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kern in sorted(agg):
        m = {c: sum(v[len(v)//3:]) / max(1, len(v[len(v)//3:])) for c, v in agg[kern].items()}
        print("%s %s: " % (d, kern) + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(m.items())))
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            print("     wait_any %.0f%%  wait_inst %.0f%%  active_inst %.0f%%  lds_conflict/lds_active %.1f%%" % (
                100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, m.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
