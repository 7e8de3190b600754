set -o pipefail
export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 3000 $O/bench_default.json
cd /tmp
for mode in 0 1; do
  rm -rf /tmp/kt_$mode
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$mode -- $R/tools/k2_microbench 2000 5 $mode > $O/kt_$mode.log 2>&1
  for f in $(find /tmp/kt_$mode -name '*kernel_stats.csv'); do cp $f $O/micro_store${mode}_kernel_stats.csv; done
done
cat $O/micro_store0_kernel_stats.csv $O/micro_store1_kernel_stats.csv | cut -c1-200
