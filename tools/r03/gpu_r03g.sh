#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
cp autobub3hs_amd/libabub_hip.so /tmp/base.so
for v in base exp1 exp2 base exp1 exp2; do
  [ $v = base ] && cp /tmp/base.so autobub3hs_amd/libabub_hip.so || cp variants/$v.so autobub3hs_amd/libabub_hip.so
  echo "variant $v: $(timeout -k 10 200 python3 tools/ab_k2.py --reps 4 --nocheck chain=4 chain=2 chain=8 2>/dev/null | cut -c1-20,60-140 | tr '\n' ' ')"
done | tee $O/valu_exp.txt
cp /tmp/base.so autobub3hs_amd/libabub_hip.so
