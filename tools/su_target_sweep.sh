set -o pipefail
O=gpurun_out/r04b; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_sputils_gpu.py -m gpu -x -q > $O/t2.log 2>&1; echo "tests exit=$?"; tail -2 $O/t2.log
timeout -k 10 100 python tools/kbench_aux.py --sizes 35718 --sputils --vn-cols 2 > $O/aux_default.log 2>&1; echo default; grep -h "K7" $O/aux_default.log | cut -c1-150
for t in 480 640 800 960 1120 1280 1440; do
  SPC_SU_FIT=0 SPC_SU_TARGET=$t timeout -k 10 100 python tools/kbench_aux.py --sizes 35718 --sputils --vn-cols 2 > $O/aux_t$t.log 2>&1 || exit 1
  echo "target $t"; grep -h "K7" $O/aux_t$t.log | cut -c1-150
done
