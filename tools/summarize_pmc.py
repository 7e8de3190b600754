#!/usr/bin/env python3
"""Summarise tools/prof_k2.sh output (gpurun_out/<tag>/pmc_*.csv + micro.json) into one JSON:
per-launch and per-job HBM traffic with the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE counts
half of a wide streaming read -> x2; WRITE_SIZE exact; both in KiB), SQ utilisation ratios, L2 hit rate."""
import csv
import glob
import json
import os
import sys


def main(d, out):
    micro = json.loads(open(os.path.join(d, "micro.json")).read().strip().splitlines()[-1])
    # one microbench launch = one dispatch of each K2 kernel involved (trigger-only: bound scan + exact groups +
    # row machine on the handed-over rows; store mode: the row machine alone): mean per dispatch and kernel, then
    # summed over the kernels
    per = {}
    meta = {}
    for f in glob.glob(os.path.join(d, "pmc_*.csv")):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"].split("(")[0]
            per.setdefault(r["Counter_Name"], {}).setdefault(kn, []).append(float(r["Counter_Value"]))
            if kn not in meta:
                meta[kn] = {k: r.get(k) for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size")}
    m = {cn: sum(sum(v) / len(v) for v in kd.values()) for cn, kd in per.items()}
    by_kernel = {cn: {kn: sum(v) / len(v) for kn, v in kd.items()} for cn, kd in per.items()
                 if cn in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")}
    jobs = micro["frames"]
    fetch = m.get("FETCH_SIZE", 0) * 1024 * 2
    write = m.get("WRITE_SIZE", 0) * 1024
    P = micro["W"] * micro["H"]
    res = {
        "micro": micro, "kernels": meta,
        "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write,
        "hbm_bytes_per_job": (fetch + write) / jobs, "hbm_bytes_per_job_over_WH": (fetch + write) / jobs / P,
        "algorithmic_bytes_per_job": (4 if micro["store"] else 3) * P,
        "l2_hit_rate": m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)),
        "valu_insts_per_pixel_lane": m.get("SQ_INSTS_VALU", 0) * 64 / (jobs * P),
        "wave_cycles_share": {k: m.get(k, 0) / max(1.0, m.get("SQ_WAVE_CYCLES", 1)) for k in
                              ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU")},
        "counters_mean": m, "counters_by_kernel": by_kernel,
        "corrections": "FETCH_SIZE KiB x2 (gfx950 wide streaming reads), WRITE_SIZE KiB x1; separate --pmc passes",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("hbm_bytes_per_job_over_WH", "l2_hit_rate", "valu_insts_per_pixel_lane", "wave_cycles_share")}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
