#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03p; mkdir -p $O
cd $R
B="python3 bench.py --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --latency-steps 3 --min-seconds 1.5 --regime-steps 0"
P='import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{\"metric")][-1]); s=r["config"]["stage_ms"]; print("value",round(r["value"]),"ms",round(r["ms_per_step"],3),"lat",round(r["config"]["latency_one_step_at_a_time_ms"]["median"],2),"jobs",s["trigger_jobs"],"s1",s["stage1_ms"],"s2",s["stage2_ms"],"s3",s["stage3_ms"],"s4",s["stage4_ms"],"pairs",s["pairs"])'
for reg in default post_trigger_dense; do
for blk in "24 8" "16 8" "14 4" "12 4" "20 4" "40 8"; do
  set -- $blk
  echo "$reg block0=$1 block=$2: $(ABUB_PIPE_BLOCK0=$1 ABUB_PIPE_BLOCK=$2 timeout -k 10 300 $B --regime $reg 2>/dev/null | python3 -c "$P")"
done; done | tee $O/blocks_ab.txt
echo "noisy: $(timeout -k 10 300 $B --regime noisy 2>/dev/null | python3 -c "$P")" | tee -a $O/blocks_ab.txt
