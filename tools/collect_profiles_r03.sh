#!/bin/bash
# Collects what profiles/r03/ holds of the final tree (run on the GPU box from the repo root): bench lines, PMC passes and
# kernel-trace stats on bench.py itself at both geometries, the chip's copy ceilings.  Results land under gpurun_out/r03/.
export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03; mkdir -p $O
timeout -k 10 800 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 800 python bench.py --width 1680 --height 1050 --cpu-seconds 6 --micro-frames 4000 --ingest-events 48 > $O/bench_1680x1050.json 2> $O/bench_1680.err; echo "bench1680 rc=$?"
bash tools/prof_bench_pmc.sh r03/pmc_bench > $O/pmc_bench.log 2>&1; echo "pmc rc=$?"; tail -1 $O/pmc_bench.log | cut -c1-300
bash tools/prof_bench_pmc.sh r03/pmc_bench_1680 --width 1680 --height 1050 > $O/pmc_bench_1680.log 2>&1; echo "pmc1680 rc=$?"; tail -1 $O/pmc_bench_1680.log | cut -c1-300
cd /tmp; rm -rf /tmp/kt_def
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_def -- python3 $R/bench.py --steps 6 --warmup 3 --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --min-seconds 0 --latency-steps 0 --regime-steps 0 > $O/trace_default.log 2>&1
for f in $(find /tmp/kt_def -name '*kernel_stats.csv'); do head -1 $f > $O/bench_default_kernel_stats.csv; grep -E 'k2_|k3_|sus_|k1_|k4_|k_hist|k_pairs|k_slot|k1b|k_sigma|k_fill' $f >> $O/bench_default_kernel_stats.csv; done
cd $R
./tools/rowload_bench 2000 0 copy > $O/copy_ceiling.jsonl 2>&1
# GPU PNG decode alone (tools/png_bench.py): two waves per stream (default) and one, two compression levels, the real geometry
{
  for cfg in "1024 1" "1024 6" "1024 1 1680 1050"; do
    for m in 2 1; do echo "waves=$m: $(ABUB_PNG_WAVES=$m timeout -k 10 300 python tools/png_bench.py $cfg 2>&1 | tail -1)"; done
  done
} > $O/png_bench.txt 2>&1
cd /tmp; rm -rf /tmp/png_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/png_kt -- python3 $R/tools/png_bench.py 1024 1 > $O/png_trace.log 2>&1
for f in $(find /tmp/png_kt -name '*kernel_stats.csv'); do head -1 $f > $O/png_kernel_stats.csv; grep -E 'k_png' $f >> $O/png_kernel_stats.csv; done
cd $R
ls $O
