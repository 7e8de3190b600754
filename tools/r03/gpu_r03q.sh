#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03q; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -8 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
B="python3 bench.py --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --latency-steps 3 --min-seconds 1.5 --regime-steps 0"
P='import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{\"metric")][-1]); s=r["config"]["stage_ms"]; print("value",round(r["value"]),"ms",round(r["ms_per_step"],3),"lat",round(r["config"]["latency_one_step_at_a_time_ms"]["median"],2),"jobs",s["trigger_jobs"],"ondemand",s["jobs_completed_on_demand"],"s1",s["stage1_ms"],"s2",s["stage2_ms"],"s3",s["stage3_ms"],"s4",s["stage4_ms"])'
for reg in default post_trigger_dense noisy; do
for d in 1 0; do
  echo "$reg defer=$d: $(ABUB_PIPE_DEFER=$d timeout -k 10 300 $B --regime $reg 2>/dev/null | python3 -c "$P")"
done; done | tee $O/defer_ab.txt
echo "default 1680 defer=1: $(timeout -k 10 300 $B --width 1680 --height 1050 2>/dev/null | python3 -c "$P")" | tee -a $O/defer_ab.txt
