#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03s; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_events.py -x -q > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
cp autobub3hs_amd/libabub_hip.so /tmp/new.so
for v in base new base new; do
  [ $v = base ] && cp variants/k3base.so autobub3hs_amd/libabub_hip.so || cp /tmp/new.so autobub3hs_amd/libabub_hip.so
  echo "$v K3 discs compact: $(timeout -k 5 120 ./tools/k2_microbench 2000 8 0 1280 1024 0 1 1 0 1 1 1 | tail -1 | cut -c60-170)"
  echo "$v K3 quiet: $(timeout -k 5 120 ./tools/k2_microbench 2000 8 0 1280 1024 0 1 1 0 0 1 0 | tail -1 | cut -c60-170)"
  echo "$v K3 1680 discs: $(timeout -k 5 120 ./tools/k2_microbench 2000 8 0 1680 1050 0 1 1 0 1 1 1 | tail -1 | cut -c60-170)"
done | tee $O/k3_ab.txt
cp /tmp/new.so autobub3hs_amd/libabub_hip.so
