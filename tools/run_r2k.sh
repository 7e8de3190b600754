export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
O=gpurun_out/r2k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "k3 or fused" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
ABUB_K3_KF=1 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "k3 or fused" > $O/pytest1.log 2>&1; echo "pytest kf1 rc=$?"; tail -3 $O/pytest1.log
M=./tools/k2_microbench
( echo "# k3 kf2 / kf1 (discs)"; $M 2000 5 0 1280 1024 0 1 1 0 1 1; ABUB_K3_KF=1 $M 2000 5 0 1280 1024 0 1 1 0 1 1;
  echo "# k3 kf2 / kf1 (nodisc)"; $M 2000 5 0 1280 1024 0 1 1 0 0 1; ABUB_K3_KF=1 $M 2000 5 0 1280 1024 0 1 1 0 0 1;
  echo "# k3 kf2 / kf1 1680"; $M 2000 5 0 1680 1050 0 1 1 0 1 1; ABUB_K3_KF=1 $M 2000 5 0 1680 1050 0 1 1 0 1 1 ) 2>&1 | tee $O/micro.jsonl
