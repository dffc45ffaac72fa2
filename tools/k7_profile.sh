#!/bin/bash
# rocprofv3 passes over the K7 operators at 35 718 rows: kernel-trace stats, FETCH_SIZE, WRITE_SIZE and two SQ passes
# (each --pmc pass its own run, no tracing next to it) -> gpurun_out/k7_<tag>/ and one summary gpurun_out/k7_<tag>.log
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/k7_$TAG
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/k7_run.py 35718 40 > $O/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/k7_run.py 35718 12 > $O/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/k7_run.py 35718 12 > $O/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/sq1 -- python3 $R/tools/k7_run.py 35718 12 > $O/sq1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/sq2 -- python3 $R/tools/k7_run.py 35718 12 > $O/sq2.log 2>&1 || exit 1
cd $R && python3 tools/k7_summary.py $O > gpurun_out/k7_$TAG.log 2>&1
echo "k7 profile done: gpurun_out/k7_$TAG.log"
