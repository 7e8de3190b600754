#!/bin/bash
# usage: tools/prof_k2.sh <tag> <frames> <store> [W H R sigma]  -- microbench + separate PMC passes (run on the GPU box)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; F=${2:-2000}; ST=${3:-0}; EXTRA="${@:4}"
O=$R/gpurun_out/$TAG; mkdir -p $O
export LD_LIBRARY_PATH=$R/autobub3hs_amd:$LD_LIBRARY_PATH
cd /tmp
$R/tools/k2_microbench $F 5 $ST $EXTRA > $O/micro.json 2>&1 || { cat $O/micro.json; exit 1; }
cat $O/micro.json
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$name
  timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$name -- $R/tools/k2_microbench $F 2 $ST $EXTRA > $O/pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; tail -3 $O/pmc_$name.log; continue; }
  for f in $(find /tmp/pmc_$name -name '*counter_collection.csv'); do
    head -1 $f > $O/pmc_$name.csv; grep -E "k2_rows|k2_bound_scan|k2_sad_chain|sus_tail" $f >> $O/pmc_$name.csv
  done
done
ls -la $O
