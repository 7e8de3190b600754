#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03i; mkdir -p $O
cd $R
cp autobub3hs_amd/libabub_hip.so /tmp/base.so
for v in base glob base glob; do
  [ $v = base ] && cp /tmp/base.so autobub3hs_amd/libabub_hip.so || cp variants/$v.so autobub3hs_amd/libabub_hip.so
  echo "variant $v 1280: $(timeout -k 10 200 python3 tools/ab_k2.py --reps 5 chain=4 chain=3 chain=2 2>/dev/null | cut -c1-20,60-128 | tr '\n' ' ')"
  echo "variant $v 1680: $(timeout -k 10 200 python3 tools/ab_k2.py --width 1680 --height 1050 --reps 4 chain=4,split=2 chain=3,split=2 2>/dev/null | cut -c1-30,70-138 | tr '\n' ' ')"
done | tee $O/bufglob.txt
cp /tmp/base.so autobub3hs_amd/libabub_hip.so
