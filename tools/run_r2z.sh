export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
R=$PWD
A="--steps 3 --warmup 1 --inflight 1 --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --min-seconds 0 --latency-steps 0"
for s in 1 0; do
ABUB_K2_STRIPS=$s bash tools/prof_bench.sh r2z_s$s $A > /dev/null 2>&1
echo "== K2_STRIPS=$s"
python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/r2z_s$s/kernel_stats.csv')):
    n=r['Name']
    if any(k in n for k in ('sus_','k2_bound_scan','k2_rows','k2_bound_chain','k2_strips')):
        print(f"  {n[:48]:48s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
done
B="python bench.py --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --latency-steps 0 --min-seconds 1"
P='import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{\"metric")][-1]); print("value",round(r["value"]),"ms",round(r["ms_per_step"],3),"roof_ms",round(r["roofline"]["ms_per_launch"],3))'
for s in 1 0 1 0; do
echo "bench K2_STRIPS=$s: $(ABUB_K2_STRIPS=$s timeout -k 10 300 $B 2>/dev/null | python3 -c "$P")"
done
