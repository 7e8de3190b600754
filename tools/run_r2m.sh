export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
O=gpurun_out/r2m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_events.py tests/test_bellows.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
ABUB_K3_SCAN=0 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "k3 or fused" > $O/pytest1.log 2>&1; echo "pytest noscan rc=$?"; tail -3 $O/pytest1.log
M=./tools/k2_microbench
( echo "# k3 sigma2 scan / noscan (discs)"; $M 2000 5 0 1280 1024 0 2 1 0 1 1; ABUB_K3_SCAN=0 $M 2000 5 0 1280 1024 0 2 1 0 1 1;
  echo "# k3 sigma2 scan / noscan (nodisc)"; $M 2000 5 0 1280 1024 0 2 1 0 0 1; ABUB_K3_SCAN=0 $M 2000 5 0 1280 1024 0 2 1 0 0 1;
  echo "# k3 sigma1 scan / noscan (discs; noisy: every row non-zero)"; $M 2000 5 0 1280 1024 0 1 1 0 1 1; ABUB_K3_SCAN=0 $M 2000 5 0 1280 1024 0 1 1 0 1 1;
  echo "# k3 sigma2 cycle8 scan / noscan (L2 resident)"; $M 2000 5 0 1280 1024 0 2 1 8 0 1; ABUB_K3_SCAN=0 $M 2000 5 0 1280 1024 0 2 1 8 0 1;
  echo "# k3 sigma2 scan / noscan 1680"; $M 2000 5 0 1680 1050 0 2 1 0 1 1; ABUB_K3_SCAN=0 $M 2000 5 0 1680 1050 0 2 1 0 1 1 ) 2>&1 | tee $O/micro.jsonl
timeout -k 10 600 python bench.py --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python3 -c "
import json
r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('value',r['value'],'ms',r['ms_per_step'],'roof',r['roofline']['frac'],r['roofline']['ms_per_launch'], r['config']['stage_ms'], r['config']['latency_one_step_at_a_time_ms'])
"
