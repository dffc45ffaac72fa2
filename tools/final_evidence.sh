R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
timeout -k 10 500 python tools/kbench.py --sizes 1024,2048,4096,8192,16384,35718,43566,87132,174264,348528 --cbs 0 > gpurun_out/kbench_sizes_final.log 2>&1; grep n= gpurun_out/kbench_sizes_final.log
timeout -k 10 200 python tools/kbench.py --levels 137,512 --sizes 1024,11105,88838 --cbs 0 > gpurun_out/kbench_sizes_final5.log 2>&1; grep n= gpurun_out/kbench_sizes_final5.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_aux -- python3 $R/tools/kbench_aux.py --sizes 35718 --vn-cols 64 > $R/gpurun_out/prof_aux.log 2>&1; echo "aux prof exit=$?"
PMC_COLS=1024 PMC_ROT=8 bash -c 'N=1024; ROT=8; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d '$R'/gpurun_out/pmc_fetch_c2 -- python3 '$R'/tools/pmc_run.py $N $ROT > '$R'/gpurun_out/pmc_fetch_c2.log 2>&1; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d '$R'/gpurun_out/pmc_write_c2 -- python3 '$R'/tools/pmc_run.py $N $ROT > '$R'/gpurun_out/pmc_write_c2.log 2>&1'
cd $R && PMC_TAG="round 2 final, config 2" python tools/pmc_summary.py gpurun_out/pmc_fetch_c2 gpurun_out/pmc_write_c2 1024 268435456 gpurun_out/traffic_c2.json > gpurun_out/pmc_summary_c2.log 2>&1; grep -E "hbm_bytes_per_launch|hbm_bytes_per_column" gpurun_out/pmc_summary_c2.log
SPC_FUZZ_TRIALS=1500 SPC_FUZZ_SEED=99 timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "random_geometries or random_batch_sizes" > gpurun_out/t_soak_final.log 2>&1; tail -2 gpurun_out/t_soak_final.log
