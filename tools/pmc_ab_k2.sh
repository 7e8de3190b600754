#!/bin/bash
# usage: tools/pmc_ab_k2.sh <outdir> "<ab_k2 args>" cfg [cfg ...]
# Per configuration: two rocprofv3 --pmc passes (FETCH_SIZE WRITE_SIZE is too wide for one pass on gfx950: FETCH_SIZE, then
# TCC_HIT_sum TCC_MISS_sum) around `python3 tools/ab_k2.py --reps 1 <cfg>`; prints mean counters of k2_bound_chain dispatches.
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$1; shift; ARGS=$1; shift
mkdir -p $O
cd /tmp
for cfg in "$@"; do
  tag=$(echo $cfg | tr ',=' '__')
  for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU"; do
    name=$(echo $pass | cut -d' ' -f1)
    rm -rf /tmp/pa_$name
    timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex "k2_bound_chain|k2_sad_chain" --output-format csv -d /tmp/pa_$name -- python3 $R/tools/ab_k2.py $ARGS --reps 1 --nocheck $cfg > $O/pmc_${tag}_$name.log 2>&1 || { echo "pmc pass $name of $cfg failed"; tail -3 $O/pmc_${tag}_$name.log; exit 1; }
    for f in $(find /tmp/pa_$name -name '*counter_collection.csv'); do cp $f $O/pmc_${tag}_$name.csv; done
  done
  python3 - $O $tag "$cfg" <<'PY'
import csv, sys, glob, os
o, tag, cfg = sys.argv[1:4]
acc = {}
for f in glob.glob(os.path.join(o, f"pmc_{tag}_*.csv")):
    for r in csv.DictReader(open(f)):
        if "k2_bound_chain" not in r["Kernel_Name"] and "k2_sad_chain" not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
out = {"cfg": cfg}
for k, v in acc.items():
    out[k] = sum(x for x, _ in v) / len(v)
    out.setdefault("ns_" + k, sum(t for _, t in v) / len(v))
if "FETCH_SIZE" in out:
    out["fetch_GB_x2"] = out["FETCH_SIZE"] * 1024 * 2 / 1e9
if "TCC_HIT_sum" in out:
    out["l2_hit"] = out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"])
if "GRBM_GUI_ACTIVE" in out:
    out["clock_GHz"] = out["GRBM_GUI_ACTIVE"] / out["ns_GRBM_GUI_ACTIVE"]
import json
print(json.dumps(out))
PY
done
