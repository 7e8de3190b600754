#!/bin/bash
# usage: tools/prof_bench_pmc.sh <tag> [bench args...]   (run on the GPU box from the repo root; ONE rank only: no --gpus N > 1 --
#        the profiler's preloaded library initialises the GPU before bench.py could start its ranks)
# rocprofv3 --pmc passes wrapping `python3 bench.py` ITSELF (the program directly after `--`), one pass per counter
# group (FETCH_SIZE and WRITE_SIZE do not fit one pass), plus a kernel-trace + stats pass of the same command.
# Leaves per-dispatch rows of this repo's kernels under gpurun_out/<tag>/ and a summary JSON (tools/summarize_bench_pmc.py).
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
case " $* " in *" --gpus "[2-9]*|*" --gpus="[2-9]*) echo "prof_bench_pmc.sh: profile a single rank (no --gpus N > 1 under rocprofv3)"; exit 2;; esac
O=$R/gpurun_out/$TAG; mkdir -p $O
rm -f $O/pmc_*.csv $O/bench_pmc_summary.json   # never summarise the counter files of an earlier collection
ARGS="--steps 2 --warmup 1 --inflight 1 --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --min-seconds 0 --latency-steps 0 --regime-steps 0 $@"
cd /tmp
KPAT='k2_|k3_|sus_|k1_|k4_|k_hist|k_pairs|k_slot|k1b|k_sigma|k_fill'
rm -rf /tmp/bp_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp_trace -- python3 $R/bench.py $ARGS > $O/trace_bench.log 2>&1 || { echo "kernel-trace pass failed"; tail -3 $O/trace_bench.log; exit 1; }
for f in $(find /tmp/bp_trace -name '*kernel_stats.csv'); do head -1 $f > $O/kernel_stats.csv; grep -E "$KPAT" $f >> $O/kernel_stats.csv; done
for f in $(find /tmp/bp_trace -name '*kernel_trace.csv'); do head -1 $f > $O/kernel_trace_abub.csv; grep -E "$KPAT" $f >> $O/kernel_trace_abub.csv; done
grep '^{"metric' $O/trace_bench.log | tail -1 > $O/bench_line.json
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/bp_$name
  # --kernel-include-regex: counters are collected on this repo's kernels only (profiles/r03/README.md: the one collection
  # of round 2 in which every pass died did so inside the launch of torch's first element-wise kernel)
  timeout -k 10 400 rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex "k2_|k3_|sus_|k_hist" --output-format csv -d /tmp/bp_$name -- python3 $R/bench.py $ARGS > $O/pmc_$name.log 2>&1 || { echo "prof_bench_pmc.sh: pmc pass $name FAILED (rc $?): no summary is written"; tail -5 $O/pmc_$name.log; exit 1; }
  n=0
  for f in $(find /tmp/bp_$name -name '*counter_collection.csv'); do
    head -1 $f > $O/pmc_$name.csv; grep -E "$KPAT" $f >> $O/pmc_$name.csv; n=$((n+1))
  done
  [ $n -eq 0 ] && { echo "prof_bench_pmc.sh: pmc pass $name left no counter file: no summary is written"; exit 1; }
done
python3 $R/tools/summarize_bench_pmc.py $O $O/bench_pmc_summary.json
