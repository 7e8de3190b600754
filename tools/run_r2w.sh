export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
O=gpurun_out/r2w; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 600 python bench.py --no-cpu-baseline --ingest-events 0 --stream-steps 0 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value',r['value'],'ms',r['ms_per_step'],'roof',r['roofline']['frac'],r['roofline']['ms_per_launch'])
m=r['config']['microbench']; print('  micro',m['frames'],{k:(round(v['us_per_job'],4),round(v['frac_of_8TBps'],3)) for k,v in m.items() if isinstance(v,dict) and 'us_per_job' in v})"
