#!/bin/bash
# Collects everything profiles/r02/ holds (run on the GPU box from the repo root): bench lines, kernel-trace stats,
# PMC passes on bench.py itself and on the native K2 microbench.  Results land under gpurun_out/r02/.
export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 600 python bench.py --width 1680 --height 1050 --cpu-seconds 6 --micro-frames 4000 > $O/bench_1680x1050.json 2> $O/bench_1680.err; echo "bench1680 rc=$?"
bash tools/prof_bench_pmc.sh r02/pmc_bench > $O/pmc_bench.log 2>&1; tail -1 $O/pmc_bench.log | cut -c1-400
bash tools/prof_bench_pmc.sh r02/pmc_bench_1680 --width 1680 --height 1050 > $O/pmc_bench_1680.log 2>&1; tail -1 $O/pmc_bench_1680.log | cut -c1-400
# default bench (several steps in flight) kernel-trace stats
cd /tmp; rm -rf /tmp/kt_def
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_def -- python3 $R/bench.py --steps 6 --warmup 3 --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --min-seconds 0 --latency-steps 0 > $O/trace_default.log 2>&1
for f in $(find /tmp/kt_def -name '*kernel_stats.csv'); do head -1 $f > $O/bench_default_kernel_stats.csv; grep -E 'k2_|k3_|sus_|k1_|k4_|k_hist|k_pairs|k_slot|k1b|k_sigma|k_fill' $f >> $O/bench_default_kernel_stats.csv; done
cd $R
# native microbench (BASELINE configs[2] kernel) with counters: trigger-only, store, store through the row machine alone
bash tools/prof_k2.sh r02/k2_hist 2000 0 > $O/prof_k2_hist.log 2>&1; python3 tools/summarize_pmc.py gpurun_out/r02/k2_hist $O/k2_hist_pmc_summary.json
bash tools/prof_k2.sh r02/k2_store 2000 1 > $O/prof_k2_store.log 2>&1; python3 tools/summarize_pmc.py gpurun_out/r02/k2_store $O/k2_store_pmc_summary.json
ABUB_K2_BOUND=0 bash tools/prof_k2.sh r02/k2_store_rowmachine 2000 1 > $O/prof_k2_store_rm.log 2>&1; python3 tools/summarize_pmc.py gpurun_out/r02/k2_store_rowmachine $O/k2_store_rowmachine_pmc_summary.json
bash tools/prof_k2.sh r02/k2_hist_1680 2000 0 1680 1050 > $O/prof_k2_hist_1680.log 2>&1; python3 tools/summarize_pmc.py gpurun_out/r02/k2_hist_1680 $O/k2_hist_1680_pmc_summary.json
./tools/valu_rate > $O/valu_rate.jsonl 2>&1
./tools/rowload_bench 2000 0 > $O/rowload.jsonl 2>&1
ls $O
