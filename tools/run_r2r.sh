export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2r; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_events.py tests/test_config3_gpu.py tests/test_bellows.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
M=./tools/k2_microbench
( echo "# trigger / store (discs)"; $M 2000 5 0; $M 2000 5 1; echo "# 1680 trigger / store"; $M 2000 5 0 1680 1050; $M 2000 5 1 1680 1050;
  echo "# k3 sigma1 / sigma2 (discs)"; $M 2000 5 0 1280 1024 0 1 1 0 1 1; $M 2000 5 0 1280 1024 0 2 1 0 1 1 ) 2>&1 | cut -c1-220
cd /tmp; rm -rf /tmp/kb
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb -- python3 $R/bench.py --steps 6 --warmup 2 --inflight 1 --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --min-seconds 0 --latency-steps 0 > $O/bench_trace.log 2>&1
for f in $(find /tmp/kb -name '*kernel_stats.csv'); do cp $f $O/bench_kernel_stats.csv; done
python3 - <<'PY'
import csv
for r in csv.DictReader(open('/root/repo/gpurun_out/r2r/bench_kernel_stats.csv')):
    if r['Name'].startswith(('void k','k_')):
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
cd $R; timeout -k 10 600 python bench.py --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value',r['value'],'ms',r['ms_per_step'],'roof',r['roofline']['frac'],r['roofline']['ms_per_launch'],r['config']['latency_one_step_at_a_time_ms'])"
