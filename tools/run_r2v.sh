export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
O=gpurun_out/r2v; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_config3_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
M=./tools/k2_microbench
for rep in 1 2; do
for sp in 1 0; do
  echo "# split=$sp: 1280 trigger/store (discs), nodisc trigger; 1680 trigger/store, nodisc trigger"
  ( ABUB_K2_SPLIT=$sp $M 2000 5 0; ABUB_K2_SPLIT=$sp $M 2000 5 1; ABUB_K2_SPLIT=$sp $M 2000 5 0 1280 1024 0 1 1 0 0; ABUB_K2_SPLIT=$sp $M 2000 5 0 1680 1050; ABUB_K2_SPLIT=$sp $M 2000 5 1 1680 1050; ABUB_K2_SPLIT=$sp $M 2000 5 0 1680 1050 0 1 1 0 0 ) 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('   W',r['W'],'store',r['store'],'ms',r['ms_avg'],'min',r['ms_min'])"
done; done
