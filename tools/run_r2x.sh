export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
for rep in 1 2; do for c in 2 3; do
ABUB_K2_CHAIN=$c timeout -k 10 600 python bench.py --no-cpu-baseline --ingest-events 0 --stream-steps 0 --micro-frames 0 --latency-steps 0 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain $c value',round(r['value']),'ms',round(r['ms_per_step'],3),'roof',round(r['roofline']['frac'],4),round(r['roofline']['ms_per_launch'],3))"
done; done
