#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03l; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -5 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 tools/ab_k2.py --store 1 --reps 5 default chain=2 scanpf=1 > $O/ab_1280_store.jsonl 2> $O/ab_1280_store.err || { tail -5 $O/ab_1280_store.err; exit 1; }
cat $O/ab_1280_store.jsonl
timeout -k 10 400 python3 tools/ab_k2.py --width 1680 --height 1050 --store 1 --reps 4 default > $O/ab_1680_store.jsonl 2> $O/ab_1680_store.err || { tail -5 $O/ab_1680_store.err; exit 1; }
cat $O/ab_1680_store.jsonl
for s in 1 0; do timeout -k 5 120 ./tools/k2_microbench 4000 5 $s 1280 1024 | tail -1; done | tee $O/micro.txt
