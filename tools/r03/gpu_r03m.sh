#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03m; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -8 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
B="python bench.py --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --latency-steps 3 --min-seconds 1.5"
for lazy in 1 0 1 0; do
  echo "lazy=$lazy: $(ABUB_PIPE_LAZY=$lazy timeout -k 10 300 $B 2>/dev/null | python3 -c 'import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{\"metric")][-1]); print("value",round(r["value"]),"ms",round(r["ms_per_step"],3),"lat",round(r["config"]["latency_one_step_at_a_time_ms"]["median"],2), r["config"]["stage_ms"])')"
done | tee $O/lazy_ab.txt
