#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -5 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 tools/ab_k2.py --reps 5 default chain=2 scanpf=1 > $O/ab_1280.jsonl 2> $O/ab_1280.err || { tail -5 $O/ab_1280.err; exit 1; }
cat $O/ab_1280.jsonl
timeout -k 10 400 python3 tools/ab_k2.py --width 1680 --height 1050 --reps 5 default chain=2 split=0 > $O/ab_1680.jsonl 2> $O/ab_1680.err || { tail -5 $O/ab_1680.err; exit 1; }
cat $O/ab_1680.jsonl
