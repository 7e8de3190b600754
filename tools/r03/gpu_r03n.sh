#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03n; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -8 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
( time timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time || { tail -20 $O/bench_default.err; exit 1; }
tail -3 $O/bench_default.time
python3 - $O/bench_default.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
c=r["config"]
print("value",round(r["value"]),"ms",round(r["ms_per_step"],3),"roof",round(r["roofline"]["frac"],4),round(r["roofline"]["ms_per_launch"],3))
print("trigger_search",c["trigger_search"]["jobs_per_step"],"masks",c["masks"]["stacks_through_the_bellows_fallback_per_step"],"slabs",c["slabs"]["copies_of_the_run_in_hbm"])
print("regimes",{k:(round(v["value_frames_per_s"]),round(v["k2_pass_over_every_frame"]["ms"],3),v["k2_pass_over_every_frame"]["handed_over_pieces_of_32_rows"]) for k,v in c["regimes"].items()})
print("ingest",round(c["ingest_inclusive"]["frames_per_s"]),c["ingest_inclusive"]["decode_threads"],c["ingest_inclusive"]["seconds"])
print("pcie",round(c["pcie_inclusive"]["frames_per_s"]))
m=c["microbench"]; print("micro store",round(m["store_mode"]["frac_of_8TBps"],4),"trig",round(m["trigger_only"]["frac_of_8TBps"],4))
print("cpu",r["cpu_baseline"]["value"])
PY
