#!/bin/bash
# round-3 GPU batch A: kernel parity after the TU split + workgroup variants, copy ceilings, wg/sync A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q > $O/pytest_kernels.log 2>&1; rc=$?
tail -3 $O/pytest_kernels.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 ./tools/rowload_bench 2000 0 copy > $O/copy.jsonl 2>&1 || exit 1
echo "copy done"
timeout -k 10 400 python3 tools/ab_k2.py --reps 5 default wg=2,sync=0 wg=4,sync=0 wg=4,sync=1 wg=4,sync=2 wg=4,sync=4 wg=4,sync=8 wg=2,sync=2 wg=3,sync=2 chain=2,wg=4,sync=2 chain=2,wg=4,sync=0 chain=2 > $O/ab_1280.jsonl 2> $O/ab_1280.err || { tail -5 $O/ab_1280.err; exit 1; }
cat $O/ab_1280.jsonl
timeout -k 10 400 python3 tools/ab_k2.py --width 1680 --height 1050 --reps 5 default wg=2,sync=0 wg=4,sync=0 wg=4,sync=2 wg=4,sync=4 wg=2,sync=2 chain=3,wg=4,sync=2 chain=3 split=2,wg=4,sync=2 split=2 > $O/ab_1680.jsonl 2> $O/ab_1680.err || { tail -5 $O/ab_1680.err; exit 1; }
cat $O/ab_1680.jsonl
timeout -k 10 400 python3 tools/ab_k2.py --store 1 --reps 4 default wg=4,sync=0 wg=4,sync=2 wg=4,sync=4 wg=2,sync=2 > $O/ab_1280_store.jsonl 2> $O/ab_1280_store.err || { tail -5 $O/ab_1280_store.err; exit 1; }
cat $O/ab_1280_store.jsonl
