#!/bin/bash
# GPU PNG decode: throughput per batch size / compression level, kernel split
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r03
export TMPDIR=/tmp
for cfg in "656 1" "1312 1" "656 6" "328 1"; do
  timeout -k 10 200 python tools/png_bench.py $cfg 2>&1 | tail -1
done
cd /tmp; rm -rf /tmp/png_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/png_kt -- python3 $R/tools/png_bench.py 656 1 > $R/gpurun_out/r03/png_trace.log 2>&1
for f in $(find /tmp/png_kt -name '*kernel_stats.csv'); do head -1 $f > $R/gpurun_out/r03/png_kernel_stats.csv; grep -E 'k_png' $f >> $R/gpurun_out/r03/png_kernel_stats.csv; done
cat $R/gpurun_out/r03/png_kernel_stats.csv
