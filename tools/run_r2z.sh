export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
B="python bench.py --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --latency-steps 0 --min-seconds 1"
P='import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{\"metric")][-1]); print("value",round(r["value"]),"ms",round(r["ms_per_step"],3),"roof_ms",round(r["roofline"]["ms_per_launch"],3),r["config"].get("host_threads_per_pipeline"))'
for pt in 0 16 8 3 0 16; do
echo "bench inflight=6 pipe_threads=$pt: $(timeout -k 10 300 $B --pipe-threads $pt 2>/dev/null | python3 -c "$P")"
done
echo "bench inflight=8 pt=0: $(timeout -k 10 300 $B --inflight 8 2>/dev/null | python3 -c "$P")"
echo "bench 1680 inflight=6: $(timeout -k 10 300 $B --width 1680 --height 1050 2>/dev/null | python3 -c "$P")"
echo "bench 1680 inflight=3: $(timeout -k 10 300 $B --width 1680 --height 1050 --inflight 3 2>/dev/null | python3 -c "$P")"
