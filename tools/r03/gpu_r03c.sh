#!/bin/bash
# round-3 GPU batch C: v_sad_u8 chained scan -- parity, A/B against the u16 scan, PMC of the best settings
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q > $O/pytest_kernels.log 2>&1; rc=$?
tail -5 $O/pytest_kernels.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 tools/ab_k2.py --reps 5 sad=0 sad=1 sad=1,chain=3 sad=1,chain=2 sad=1,wg=4,sync=0 sad=1,wg=4,sync=2 sad=1,wg=4,sync=4 sad=1,wg=2,sync=2 sad=1,split=0 sad=1,chain=2,wg=4,sync=2 > $O/ab_1280.jsonl 2> $O/ab_1280.err || { tail -5 $O/ab_1280.err; exit 1; }
cat $O/ab_1280.jsonl
timeout -k 10 400 python3 tools/ab_k2.py --width 1680 --height 1050 --reps 5 sad=0 sad=1 sad=1,chain=3 sad=1,chain=4 sad=1,split=2 sad=1,chain=3,split=2 sad=1,wg=4,sync=2 sad=1,chain=3,split=2,wg=4,sync=2 sad=1,chain=3,split=2,wg=2,sync=2 > $O/ab_1680.jsonl 2> $O/ab_1680.err || { tail -5 $O/ab_1680.err; exit 1; }
cat $O/ab_1680.jsonl
timeout -k 10 400 python3 tools/ab_k2.py --store 1 --reps 4 sad=0 sad=1 sad=1,chain=3 sad=1,wg=4,sync=2 sad=1,chain=2 > $O/ab_1280_store.jsonl 2> $O/ab_1280_store.err || { tail -5 $O/ab_1280_store.err; exit 1; }
cat $O/ab_1280_store.jsonl
bash tools/pmc_ab_k2.sh $O "" sad=1 sad=1,wg=4,sync=2 > $O/pmc.jsonl 2> $O/pmc.err || { cat $O/pmc.jsonl; tail -5 $O/pmc.err; exit 1; }
cat $O/pmc.jsonl
