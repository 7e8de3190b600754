#!/bin/bash
# GPU PNG decode: kernel split + counters of the inflate kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r03
export TMPDIR=/tmp
N=${1:-656}
cd /tmp; rm -rf /tmp/png_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/png_kt -- python3 $R/tools/png_bench.py $N 1 > $R/gpurun_out/r03/png_trace.log 2>&1 || { tail -5 $R/gpurun_out/r03/png_trace.log; exit 1; }
for f in $(find /tmp/png_kt -name '*kernel_stats.csv'); do head -1 $f > $R/gpurun_out/r03/png_kernel_stats.csv; grep -E 'k_png' $f >> $R/gpurun_out/r03/png_kernel_stats.csv; done
cat $R/gpurun_out/r03/png_kernel_stats.csv
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pp_$name
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex "k_png_inflate" --output-format csv -d /tmp/pp_$name -- python3 $R/tools/png_bench.py $N 1 > $R/gpurun_out/r03/png_pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; tail -3 $R/gpurun_out/r03/png_pmc_$name.log; exit 1; }
  for f in $(find /tmp/pp_$name -name '*counter_collection.csv'); do
    python3 - $f <<'P'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, "mean per dispatch", sum(v)/len(v), "n", len(v))
P
  done
done
