#!/usr/bin/env python3
"""Parses the per-configuration counter CSVs written by tools/pmc_ab_k2.sh: python3 tools/pmc_parse.py <dir> <jobs> <W> <H>"""
import csv, glob, json, os, re, sys
o, jobs, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
tags = sorted({re.sub(r"_(FETCH_SIZE|TCC_HIT_sum|GRBM_GUI_ACTIVE)\.csv$", "", os.path.basename(f))[4:] for f in glob.glob(os.path.join(o, "pmc_*.csv"))})
for tag in tags:
    acc = {}
    for name in ("FETCH_SIZE", "TCC_HIT_sum", "GRBM_GUI_ACTIVE"):
        f = os.path.join(o, f"pmc_{tag}_{name}.csv")
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            if "k2_sad_chain" not in r["Kernel_Name"] and "k2_bound_chain" not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {"cfg": tag}
    for k, v in acc.items():
        out[k] = sum(x for x, _ in v) / len(v)
        out["ns_" + k] = sum(t for _, t in v) / len(v)
    res = {"cfg": tag}
    if "FETCH_SIZE" in out:
        res["fetch_GB"] = round(out["FETCH_SIZE"] * 2048 / 1e9, 3)
        res["bytes_per_job_over_P"] = round(out["FETCH_SIZE"] * 2048 / jobs / (W * H), 4)
        res["ms"] = round(out["ns_FETCH_SIZE"] / 1e6, 4)
        res["fabric_TBps"] = round(out["FETCH_SIZE"] * 2048 / out["ns_FETCH_SIZE"] / 1e3, 3)
    if "TCC_HIT_sum" in out:
        res["l2_hit"] = round(out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"]), 4)
    if "GRBM_GUI_ACTIVE" in out:
        res["clock_GHz"] = round(out["GRBM_GUI_ACTIVE"] / out["ns_GRBM_GUI_ACTIVE"] / 8, 3)
        res["valu_per_px"] = round(out["SQ_INSTS_VALU"] * 64 / (jobs * W * H), 3)
    print(json.dumps(res))
