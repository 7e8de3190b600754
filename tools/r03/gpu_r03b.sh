#!/bin/bash
# round-3 GPU batch B: where the chained scan's time goes -- traffic of the workgroup variants, VALU sensitivity, clock
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
bash tools/pmc_ab_k2.sh $O "" default wg=4,sync=0 wg=4,sync=2 wg=2,sync=0 > $O/pmc_wg.jsonl 2> $O/pmc_wg.err || { cat $O/pmc_wg.jsonl; tail -5 $O/pmc_wg.err; exit 1; }
cat $O/pmc_wg.jsonl
cp autobub3hs_amd/libabub_hip.so /tmp/base.so
for v in base drop2 drop4 base; do
  [ $v = base ] && cp /tmp/base.so autobub3hs_amd/libabub_hip.so || cp variants/$v.so autobub3hs_amd/libabub_hip.so
  echo "variant $v: $(timeout -k 10 200 python3 tools/ab_k2.py --reps 4 --nocheck default chain=2 2>/dev/null | tr '\n' ' ')"
done | tee $O/valu_drop.txt
cp /tmp/base.so autobub3hs_amd/libabub_hip.so
# clock under load: the native microbench from HBM and with L2-resident inputs (cycle = 8 frames)
export TMPDIR=/tmp; cd /tmp
for mode in "0" "8"; do
  rm -rf /tmp/clk_$mode
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --kernel-include-regex "k2_bound_chain" --output-format csv -d /tmp/clk_$mode -- $R/tools/k2_microbench 2000 5 0 1280 1024 0 1 1 $mode > $O/clk_$mode.log 2>&1 || { tail -3 $O/clk_$mode.log; exit 1; }
  for f in $(find /tmp/clk_$mode -name '*counter_collection.csv'); do cp $f $O/clk_$mode.csv; done
done
python3 - $O <<'PY'
import csv, sys, os
o = sys.argv[1]
for mode in ("0", "8"):
    acc = {}
    for r in csv.DictReader(open(os.path.join(o, f"clk_{mode}.csv"))):
        acc.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    g = acc["GRBM_GUI_ACTIVE"]
    print("cycle", mode, {k: sum(x for x, _ in v) / len(v) for k, v in acc.items()}, "ns", sum(t for _, t in g) / len(g),
          "clock GHz", sum(x for x, _ in g) / sum(t for _, t in g))
PY
