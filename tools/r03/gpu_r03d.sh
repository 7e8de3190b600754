#!/bin/bash
# round-3 GPU batch D: whole chains in one workgroup (K x NW = jobs of a residue chain)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
timeout -k 10 400 python3 tools/ab_k2.py --reps 5 sad=1 chain=5 chain=5,wg=4,sync=0 chain=5,wg=4,sync=1 chain=5,wg=4,sync=2 chain=5,wg=4,sync=4 chain=4,wg=5,sync=0 chain=4,wg=5,sync=2 chain=5,wg=2,sync=1 chain=3,wg=7,sync=2 > $O/ab_1280.jsonl 2> $O/ab_1280.err || { tail -5 $O/ab_1280.err; exit 1; }
cat $O/ab_1280.jsonl
timeout -k 10 400 python3 tools/ab_k2.py --width 1680 --height 1050 --reps 5 chain=4 chain=3 chain=3,wg=7,sync=2 chain=3,wg=7,sync=0 chain=3,wg=4,sync=1 chain=4,wg=5,sync=2 chain=2,wg=5,sync=2 chain=2,wg=5,sync=1 > $O/ab_1680.jsonl 2> $O/ab_1680.err || { tail -5 $O/ab_1680.err; exit 1; }
cat $O/ab_1680.jsonl
bash tools/pmc_ab_k2.sh $O "" chain=5,wg=4,sync=1 chain=5,wg=4,sync=2 chain=5,wg=4,sync=0 > $O/pmc.jsonl 2> $O/pmc.err || { cat $O/pmc.jsonl; tail -5 $O/pmc.err; exit 1; }
cat $O/pmc.jsonl
