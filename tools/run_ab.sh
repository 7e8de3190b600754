# A/B of two builds of libabub_hip.so (variants/base, variants/new), interleaved on one box
B="python bench.py --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --latency-steps 0 --min-seconds 1.5"
P='import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{\"metric")][-1]); print("value",round(r["value"]),"ms",round(r["ms_per_step"],3),"roof_ms",round(r["roofline"]["ms_per_launch"],3))'
for rep in 1 2 3; do
for v in base new; do
cp variants/$v/libabub_hip.so autobub3hs_amd/libabub_hip.so
export LD_LIBRARY_PATH=$PWD/autobub3hs_amd
echo "$v bench: $(timeout -k 10 300 $B 2>/dev/null | python3 -c "$P")"
echo "$v K3 micro discs=1 compact: $(timeout -k 5 120 ./tools/k2_microbench 2000 8 0 1280 1024 0 1 1 0 1 1 1 | tail -1 | cut -c60-150)"
echo "$v K3 micro quiet: $(timeout -k 5 120 ./tools/k2_microbench 2000 8 0 1280 1024 0 1 1 0 0 1 0 | tail -1 | cut -c60-150)"
done; done
for v in base new; do
cp variants/$v/libabub_hip.so autobub3hs_amd/libabub_hip.so
echo "$v bench 1680: $(timeout -k 10 300 $B --width 1680 --height 1050 2>/dev/null | python3 -c "$P")"
echo "$v K3 micro 1680 discs: $(timeout -k 5 120 ./tools/k2_microbench 2000 8 0 1680 1050 0 1 1 0 1 1 1 | tail -1 | cut -c60-150)"
done
