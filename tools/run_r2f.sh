set -o pipefail
export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_config3_gpu.py tests/test_gpu_events.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -12 $O/pytest.log
M=./tools/k2_microbench
( echo "# trigger c2 / c3"; $M 2000 5 0; ABUB_K2_CHAIN=3 $M 2000 5 0;
  echo "# store c2 / c3"; $M 2000 5 1; ABUB_K2_CHAIN=3 $M 2000 5 1;
  echo "# cycle8 trigger/store"; $M 2000 5 0 1280 1024 0 1 1 8; $M 2000 5 1 1280 1024 0 1 1 8;
  echo "# sparse sigma2 trigger / store"; $M 2000 5 0 1280 1024 0 2; $M 2000 5 1 1280 1024 0 2;
  echo "# 1680 trigger c2, store c2"; $M 2000 5 0 1680 1050; $M 2000 5 1 1680 1050 ) > $O/micro.jsonl 2>&1
cut -c1-230 $O/micro.jsonl
cd /tmp
for mode in 0 1; do
  rm -rf /tmp/kt_$mode
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$mode -- $R/tools/k2_microbench 2000 5 $mode > $O/kt_$mode.log 2>&1
  for f in $(find /tmp/kt_$mode -name '*kernel_stats.csv'); do cp $f $O/micro_store${mode}_kernel_stats.csv; done
done
python3 - <<'PY'
import csv
for f in ('micro_store0_kernel_stats.csv','micro_store1_kernel_stats.csv'):
    print(f)
    for r in csv.DictReader(open('/root/repo/gpurun_out/r2f/'+f)):
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
PY
