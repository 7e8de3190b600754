#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args...]   (run on the GPU box from the repo root; ONE rank only: no --gpus N > 1 --
#        the profiler's preloaded library initialises the GPU before bench.py could start its ranks)
# rocprofv3 kernel-trace + stats of bench.py; keeps only the small summaries under gpurun_out/<tag>/
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
case " $* " in *" --gpus "[2-9]*|*" --gpus="[2-9]*) echo "prof_bench.sh: profile a single rank (no --gpus N > 1 under rocprofv3)"; exit 2;; esac
OUT=/tmp/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/$TAG
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py "$@" > $R/gpurun_out/$TAG/bench.log 2>&1
rc=$?
for f in $(find $OUT -name '*kernel_stats.csv'); do cp $f $R/gpurun_out/$TAG/kernel_stats.csv; done
for f in $(find $OUT -name '*kernel_trace.csv'); do
  head -1 $f > $R/gpurun_out/$TAG/kernel_trace_abub.csv
  grep -E 'k2_|k1_|k3_|sus_|k4_|k_hist|k_fill|k_sigma|k1b' $f | tail -400 >> $R/gpurun_out/$TAG/kernel_trace_abub.csv
done
tail -1 $R/gpurun_out/$TAG/bench.log
exit $rc
