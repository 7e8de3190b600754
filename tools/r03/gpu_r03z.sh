#!/bin/bash
# GPU PNG decode: tests in both modes, then throughput
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r03
for m in 2 1; do
  ABUB_PNG_WAVES=$m timeout -k 10 300 python -m pytest tests/test_gpu_png.py -q > gpurun_out/png_test_$m.log 2>&1; rc=$?
  echo "waves=$m: $(grep -E 'passed|failed' gpurun_out/png_test_$m.log | tail -1)"
  [ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E  )" gpurun_out/png_test_$m.log | cut -c1-250 | head -20; exit $rc; }
done
for cfg in "656 1" "1024 1" "1024 6"; do
  for m in 2 1; do echo "waves=$m $(ABUB_PNG_WAVES=$m timeout -k 10 200 python tools/png_bench.py $cfg 2>&1 | tail -1)"; done
done
