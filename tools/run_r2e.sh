set -o pipefail
export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
timeout -k 10 600 python bench.py --micro-frames 2000 --cpu-seconds 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; python3 -c "
import json
r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('value',r['value'],'ms',r['ms_per_step'],'roof',r['roofline']['frac'],r['roofline']['ms_per_launch'])
print('ingest',json.dumps(r['config'].get('ingest_inclusive'),indent=0))
m=r['config']['microbench']; print({k:(v['us_per_job'],v['frac_of_8TBps']) for k,v in m.items() if isinstance(v,dict) and 'us_per_job' in v})
"
