export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2z; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_events.py tests/test_config3_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for d in 0 1; do for st in 0 1; do
echo "K2 1280 store=$st discs=$d: $(timeout -k 5 120 ./tools/k2_microbench 4000 6 $st 1280 1024 0 1 1 0 $d | tail -1 | cut -c84-200)"
done; done
for c in 2 3; do for st in 0 1; do
echo "K2 1680 chain=$c store=$st: $(ABUB_K2_CHAIN=$c timeout -k 5 120 ./tools/k2_microbench 3000 6 $st 1680 1050 | tail -1 | cut -c84-200)"
done; done
A="--steps 3 --warmup 1 --inflight 1 --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --min-seconds 0 --latency-steps 0"
bash tools/prof_bench.sh r2z_rp $A > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/r2z_rp/kernel_stats.csv')):
    n=r['Name']
    if any(k in n for k in ('sus_','k2_bound_scan','k2_rows','k2_bound_chain','k3_')):
        print(f"  {n[:48]:48s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
B="python bench.py --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --latency-steps 0 --min-seconds 1"
P='import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{\"metric")][-1]); print("value",round(r["value"]),"ms",round(r["ms_per_step"],3),"roof_ms",round(r["roofline"]["ms_per_launch"],3))'
for i in 1 2; do
echo "bench: $(timeout -k 10 300 $B 2>/dev/null | python3 -c "$P")"
done
echo "bench 1680: $(timeout -k 10 300 $B --width 1680 --height 1050 2>/dev/null | python3 -c "$P")"
