#!/bin/bash
# A/B of builds of libabub_hip.so under variants/<name>/ on the GPU PNG decode (interleaved, one box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp autobub3hs_amd/libabub_hip.so /tmp/keep_libabub_hip.so
for rep in 1 2; do
for v in $(ls variants); do
  cp variants/$v/libabub_hip.so autobub3hs_amd/libabub_hip.so
  echo "$v L1: $(timeout -k 10 200 python tools/png_bench.py 1024 1 2>&1 | tail -1 | cut -c40-110)   L6: $(timeout -k 10 200 python tools/png_bench.py 1024 6 2>&1 | tail -1 | cut -c40-110)"
done; done
cp /tmp/keep_libabub_hip.so autobub3hs_amd/libabub_hip.so
