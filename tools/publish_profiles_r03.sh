#!/bin/bash
# copies the summaries of tools/collect_profiles_r03.sh (gpurun_out/r03/, scratch) into profiles/r03/ (tracked)
set -e
S=gpurun_out/r03; D=profiles/r03; mkdir -p $D
cp $S/bench_default.json $S/bench_1680x1050.json $S/bench_default_kernel_stats.csv $S/copy_ceiling.jsonl $D/
cp $S/pmc_bench/bench_pmc_summary.json $D/bench_pmc_summary.json
cp $S/pmc_bench/kernel_stats.csv $D/bench_inflight1_kernel_stats.csv
cp $S/pmc_bench/kernel_trace_abub.csv $D/bench_inflight1_kernel_trace_abub.csv
cp $S/pmc_bench_1680/bench_pmc_summary.json $D/bench_1680x1050_pmc_summary.json
cp $S/pmc_bench_1680/kernel_stats.csv $D/bench_1680x1050_inflight1_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE SQ_WAVE_CYCLES TCC_HIT_sum SQ_INSTS_VMEM_RD; do cp $S/pmc_bench/pmc_$c.csv $D/bench_pmc_$c.csv; done
cp $S/png_bench.txt $S/png_kernel_stats.csv $D/
