#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
timeout -k 10 400 python3 tools/ab_k2.py --reps 6 chain=4 chain=6 chain=8 chain=5 chain=8,split=0 > $O/ab_1280.jsonl 2> $O/ab_1280.err || { tail -5 $O/ab_1280.err; exit 1; }
cat $O/ab_1280.jsonl
bash tools/pmc_ab_k2.sh $O "" chain=8 chain=6 > $O/pmc.jsonl 2> $O/pmc.err || { cat $O/pmc.jsonl; tail -5 $O/pmc.err; exit 1; }
python3 tools/pmc_parse.py $O 8000 1280 1024
