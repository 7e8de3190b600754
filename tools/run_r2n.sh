export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2n; mkdir -p $O
cd /tmp
for scan in 1 0; do
  rm -rf /tmp/kb_$scan
  ABUB_K3_SCAN=$scan timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb_$scan -- python3 $R/bench.py --steps 6 --warmup 2 --inflight 1 --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --min-seconds 0 --latency-steps 0 > $O/bench_$scan.log 2>&1
  for f in $(find /tmp/kb_$scan -name '*kernel_stats.csv'); do cp $f $O/bench_scan${scan}_kernel_stats.csv; done
done
python3 - <<'PY'
import csv
for f in ('bench_scan1_kernel_stats.csv','bench_scan0_kernel_stats.csv'):
    print(f)
    for r in csv.DictReader(open('/root/repo/gpurun_out/r2n/'+f)):
        if r['Name'].startswith(('void k','k_','void at')) and float(r['Percentage'])>0.3:
            print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
