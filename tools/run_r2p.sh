export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
bash tools/prof_bench_pmc.sh r02_pmc 2>&1 | tail -5
tail -3 gpurun_out/r02_pmc/pmc_FETCH_SIZE.log | cut -c1-300
M=./tools/k2_microbench
( echo "# store c2 (full-line zero rows)"; $M 2000 5 1; echo "# nodisc store"; $M 2000 5 1 1280 1024 0 1 1 0 0; echo "# cycle8 store"; $M 2000 5 1 1280 1024 0 1 1 8; echo "# 1680 store"; $M 2000 5 1 1680 1050 ) 2>&1 | cut -c1-200
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
