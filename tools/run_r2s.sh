export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2s; mkdir -p $O
cd /tmp
for c in 0 8 16 32; do
rm -rf /tmp/kb
ABUB_K3_CHUNKS=$c timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb -- python3 $R/bench.py --steps 6 --warmup 2 --inflight 1 --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --min-seconds 0 --latency-steps 0 > $O/bench_trace_c$c.log 2>&1
for f in $(find /tmp/kb -name '*kernel_stats.csv'); do cp $f $O/bench_kernel_stats_c$c.csv; done
done
python3 - <<'PY'
import csv
for b in (0,8,16,32):
    print('chunks',b)
    for r in csv.DictReader(open(f'/root/repo/gpurun_out/r2s/bench_kernel_stats_c{b}.csv')):
        if 'k3_' in r['Name']:
            print(f"  {r['Name'][:40]:40s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
PY
cd $R; timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "k3 or fused" 2>&1 | tail -2
