#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03o; mkdir -p $O
cd $R
B="python3 bench.py --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --latency-steps 0 --min-seconds 1.5 --regime-steps 0"
P='import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{\"metric")][-1]); print("value",round(r["value"]),"ms",round(r["ms_per_step"],3),"roof_ms",round(r["roofline"]["ms_per_launch"],3), r["config"]["stage_ms"])'
for v in "" "--no-masks" "--slabs 1" "--no-masks --slabs 1" ""; do
  echo "[$v]: $(timeout -k 10 300 $B $v 2>/dev/null | python3 -c "$P")"
done | tee $O/hygiene_ab.txt
cd /tmp; rm -rf /tmp/pr_dense
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_dense -- python3 $R/bench.py --steps 2 --warmup 1 --inflight 1 --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --latency-steps 0 --min-seconds 0 --regime-steps 4 > $O/regimes.log 2>&1 || { tail -5 $O/regimes.log; exit 1; }
for f in $(find /tmp/pr_dense -name '*kernel_stats.csv'); do cp $f $O/regimes_kernel_stats.csv; done
grep '^{"metric' $O/regimes.log | tail -1 > $O/regimes_line.json
python3 - $O/regimes_line.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
for k,v in r["config"]["regimes"].items(): print(k, {a:b for a,b in v.items() if a!="k2_pass_over_every_frame"}, v["k2_pass_over_every_frame"])
PY
head -12 $O/regimes_kernel_stats.csv | cut -c1-60,300-420
