#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03j; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q > $O/pytest_kernels.log 2>&1; rc=$?
tail -3 $O/pytest_kernels.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 tools/ab_k2.py --reps 6 chain=4 chain=4,pf=2 chain=3 chain=3,pf=2 > $O/ab_1280.jsonl 2> $O/ab_1280.err || { tail -5 $O/ab_1280.err; exit 1; }
cat $O/ab_1280.jsonl
timeout -k 10 400 python3 tools/ab_k2.py --store 1 --reps 4 chain=4 chain=4,pf=2 chain=3,pf=2 > $O/ab_1280_store.jsonl 2> $O/ab_1280_store.err || { tail -5 $O/ab_1280_store.err; exit 1; }
cat $O/ab_1280_store.jsonl
