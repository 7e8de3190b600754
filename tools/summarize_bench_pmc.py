#!/usr/bin/env python3
"""Summarise tools/prof_bench_pmc.sh output (rocprofv3 --pmc passes wrapping `python3 bench.py`) into one JSON.

Per kernel of this repo: mean counters per dispatch, restricted to the dispatches of the K2 trigger-only pass at full
size (grid of the whole run).  HBM traffic uses the gfx950 corrections of MI355X_MICROARCH.md: FETCH_SIZE (KiB) counts
half of a wide streaming read -> x2; WRITE_SIZE (KiB) x1.  `hbm_bytes_per_launch` = the sum over the kernels of one
launch of the pass (bound scan + handed-over rows + bin-0 fix): what bench.py reports as roofline.traffic."""
import csv
import glob
import json
import os
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")


def main(d, out):
    line = json.loads(open(os.path.join(d, "bench_line.json")).read().strip())
    W, H = line["config"]["width"], line["config"]["height"]
    njobs = line["roofline"]["jobs_per_launch"]
    per = {}   # counter -> kernel -> [values per dispatch]
    meta = {}
    for f in glob.glob(os.path.join(d, "pmc_*.csv")):
        for r in csv.DictReader(open(f)):
            kn = short(r["Kernel_Name"])
            per.setdefault(r["Counter_Name"], {}).setdefault(kn, []).append((int(r["Grid_Size"]), float(r["Counter_Value"])))
            meta.setdefault(kn, {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")})
    missing = [c for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum") if c not in per]
    if missing:
        raise SystemExit(f"summarize_bench_pmc.py: counter(s) {missing} are missing in {d}: a pass failed; no summary")
    # the trigger-only pass over every frame of the run (bench.py's roofline leg): the chained scan's dispatches with the
    # largest grid, and the kernels launched with it
    scan = [k for k in per.get("FETCH_SIZE", {}) if k.startswith("k2_sad_chain") or k.startswith("k2_bound_chain")]
    if not scan:
        raise SystemExit("no chained-scan dispatches in the counter files")
    scan = scan[0]
    full_grid = max(g for g, _ in per["FETCH_SIZE"][scan])
    pass_kernels = {scan: full_grid}
    for k in per["FETCH_SIZE"]:
        if k.startswith("k2_rows") and k.endswith("false>"):  # list mode of the trigger-only pass (no candidate list)
            pass_kernels[k] = None
    def mean(counter, kn, grid):
        v = [x for g, x in per.get(counter, {}).get(kn, []) if grid is None or g == grid]
        return sum(v) / len(v) if v else 0.0
    by_kernel = {}
    for kn, grid in pass_kernels.items():
        by_kernel[kn] = {c: mean(c, kn, grid) for c in per}
        by_kernel[kn].update(meta.get(kn, {}))
    fetch = sum(v.get("FETCH_SIZE", 0) for v in by_kernel.values()) * 1024 * 2
    write = sum(v.get("WRITE_SIZE", 0) for v in by_kernel.values()) * 1024
    hit = sum(v.get("TCC_HIT_sum", 0) for v in by_kernel.values())
    miss = sum(v.get("TCC_MISS_sum", 0) for v in by_kernel.values())
    P = W * H
    sc = by_kernel[scan]
    res = {
        "command": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --inflight 1 ... (tools/prof_bench_pmc.sh)",
        "W": W, "H": H, "jobs_per_launch": njobs, "pass_kernels": sorted(pass_kernels),
        "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
        "hbm_bytes_per_job_over_WH": (fetch + write) / njobs / P,
        "compulsory_bytes_per_launch": float(P) * njobs,
        "l2_hit_rate": hit / max(1.0, hit + miss),
        "scan_valu_insts_per_pixel": sc.get("SQ_INSTS_VALU", 0) * 64 / (njobs * P),
        "scan_wave_cycles_share": {k: sc.get(k, 0) / max(1.0, sc.get("SQ_WAVE_CYCLES", 1)) for k in
                                   ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")},
        "by_kernel": by_kernel,
        "corrections": "FETCH_SIZE KiB x2 (gfx950 wide streaming reads), WRITE_SIZE KiB x1; separate --pmc passes",
        "bench_line_of_the_trace_pass": {k: line[k] for k in ("value", "ms_per_step")} | {"roofline_ms_per_launch": line["roofline"]["ms_per_launch"]},
    }
    # K3 (the tracking frames of the triggered stacks): scan + suspect-list tail + handed-over pieces, per dispatch
    k3k = [k for k in per.get("FETCH_SIZE", {}) if k.startswith("k3_") or k.startswith("sus_tail_list<3")]
    if k3k:
        k3 = {kn: {c: mean(c, kn, None) for c in per} for kn in k3k}
        for kn in k3k:
            k3[kn].update(meta.get(kn, {}))
        nfr = 10 * int(line["config"].get("triggered_stacks", 0))  # <= 10 tracking frames per triggered stack
        f3 = sum(v.get("FETCH_SIZE", 0) for v in k3.values()) * 1024 * 2
        w3 = sum(v.get("WRITE_SIZE", 0) for v in k3.values()) * 1024
        scan3 = [k for k in k3k if k.startswith("k3_bound_scan")]
        res["k3"] = {
            "kernels": sorted(k3k), "tracking_frames_per_launch_at_most": nfr,
            "hbm_bytes_per_launch": f3 + w3,
            "hbm_bytes_per_frame_over_WH": (f3 + w3) / max(1, nfr) / P,
            "l2_hit_rate": sum(v.get("TCC_HIT_sum", 0) for v in k3.values()) /
                           max(1.0, sum(v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0) for v in k3.values())),
            "scan_valu_insts_per_pixel": (k3[scan3[0]].get("SQ_INSTS_VALU", 0) * 64 / max(1, nfr * P)) if scan3 else None,
            "by_kernel": k3,
        }
    # kernel durations of the trace pass (no counters): the pass's kernels at full grid
    st = os.path.join(d, "kernel_trace_abub.csv")
    if os.path.exists(st):
        dur = {}
        for r in csv.DictReader(open(st)):
            kn = short(r["Kernel_Name"])
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            if kn in pass_kernels and (pass_kernels[kn] is None or grid == pass_kernels[kn]):
                dur.setdefault(kn, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        res["kernel_us_mean_trace_pass"] = {k: sum(v) / len(v) for k, v in dur.items()}
        res["kernel_dispatches_trace_pass"] = {k: len(v) for k, v in dur.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("hbm_bytes_per_job_over_WH", "l2_hit_rate", "scan_valu_insts_per_pixel",
                                          "scan_wave_cycles_share", "kernel_us_mean_trace_pass") if k in res}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
