set -o pipefail
export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
O=gpurun_out/r2b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q > $O/pytest_kernels.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_kernels.log; tail -5 $O/pytest_kernels.log
M=./tools/k2_microbench
( echo "# trigger chain2"; $M 2000 5 0; echo "# trigger chain3"; ABUB_K2_CHAIN=3 $M 2000 5 0;
  echo "# store chain2"; $M 2000 5 1; echo "# store chain3"; ABUB_K2_CHAIN=3 $M 2000 5 1;
  echo "# store rowmachine"; ABUB_K2_BOUND=0 $M 2000 5 1; echo "# trigger rowmachine"; ABUB_K2_BOUND=0 $M 2000 5 0;
  echo "# store nochain"; $M 2000 5 1 1280 1024 0 1 0;
  echo "# sparse sigma2 trigger / store"; $M 2000 5 0 1280 1024 0 2; $M 2000 5 1 1280 1024 0 2;
  echo "# 1680 trigger c2/c3, store c2/c3, store rowmachine"; $M 2000 5 0 1680 1050; ABUB_K2_CHAIN=3 $M 2000 5 0 1680 1050; $M 2000 5 1 1680 1050; ABUB_K2_CHAIN=3 $M 2000 5 1 1680 1050; ABUB_K2_BOUND=0 $M 2000 5 1 1680 1050 ) > $O/micro.jsonl 2>&1
cat $O/micro.jsonl
timeout -k 10 200 ./tools/rowload_bench 2000 0 > $O/rowload.jsonl 2>&1; tail -3 $O/rowload.jsonl
