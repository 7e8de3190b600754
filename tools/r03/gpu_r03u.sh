#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03u; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q > $O/pytest_kernels.log 2>&1; rc=$?
tail -5 $O/pytest_kernels.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 tools/ab_k2.py --reps 5 default scanpf=1 handoff=1,wg=4,chain=4 handoff=1,wg=4,chain=5 handoff=1,wg=5,chain=4 handoff=1,wg=2,chain=4 wg=4,chain=4,scanpf=1 > $O/ab_1280.jsonl 2> $O/ab_1280.err || { tail -5 $O/ab_1280.err; exit 1; }
cat $O/ab_1280.jsonl
timeout -k 10 400 python3 tools/ab_k2.py --width 1680 --height 1050 --reps 4 default handoff=1,wg=4,chain=4 handoff=1,wg=5,chain=4 handoff=1,wg=2,chain=4 > $O/ab_1680.jsonl 2> $O/ab_1680.err || { tail -5 $O/ab_1680.err; exit 1; }
cat $O/ab_1680.jsonl
bash tools/pmc_ab_k2.sh $O "" handoff=1,wg=4,chain=5 handoff=1,wg=4,chain=4 > $O/pmc.jsonl 2> $O/pmc.err || { tail -5 $O/pmc.err; }
python3 tools/pmc_parse.py $O 8000 1280 1024
