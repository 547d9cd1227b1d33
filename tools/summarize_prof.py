"""Summarise the rocprofv3 output of tools/prof.sh: per-launch counter means of one kernel and the
HBM traffic corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts 128-B requests at 64 B: double it)."""
import csv, glob, json, re, sys

def main():
    out_dir, kregex, args = sys.argv[1], sys.argv[2], sys.argv[3]
    rx = re.compile(kregex)
    res = {"command": f"rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --no-cpu {args} (one run per counter group, tools/prof.sh)",
           "kernel_regex": kregex, "counters": {}}
    for f in glob.glob(f"{out_dir}/pmc_*/*/*_counter_collection.csv"):
        acc = {}
        for row in csv.DictReader(open(f)):
            if rx.search(row["Kernel_Name"]):
                acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
                res["kernel"] = row["Kernel_Name"]
        for name, per in acc.items():
            v = list(per.values())
            res["counters"][name] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    for f in glob.glob(f"{out_dir}/trace/*/*_kernel_stats.csv"):
        for row in csv.DictReader(open(f)):
            if rx.search(row["Name"]):
                res["kernel_trace"] = {"calls": int(row["Calls"]), "average_ns": float(row["AverageNs"]),
                                       "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
    # per-launch durations: the first launch of a process is cold (code object load, page faults) and skews the
    # --stats average of a short run; the bench's own HIP-event average excludes its warm-up steps
    for f in glob.glob(f"{out_dir}/trace/*/*_kernel_trace.csv"):
        d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
             for r in csv.DictReader(open(f)) if rx.search(r["Kernel_Name"])]
        d = [x for _, x in sorted(d)]
        if len(d) > 1 and "kernel_trace" in res:
            res["kernel_trace"]["median_ns"] = float(sorted(d)[len(d) // 2])
            res["kernel_trace"]["average_ns_without_first_launch"] = sum(d[1:]) / (len(d) - 1)
    c = res["counters"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fr, wr = c["FETCH_SIZE"]["mean_per_launch"] * 1024, c["WRITE_SIZE"]["mean_per_launch"] * 1024
        res["hbm_traffic_per_launch_bytes"] = {"fetch_raw": fr, "fetch_corrected_x2": 2 * fr, "write": wr,
                                               "total_corrected": 2 * fr + wr}
        if "kernel_trace" in res:
            res["hbm_gbps_corrected"] = (2 * fr + wr) / res["kernel_trace"]["average_ns"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        # MFMA busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
        cycles = c["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8
        res["mfma_busy_frac_of_simd_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024 / cycles
        if "kernel_trace" in res:
            res["shader_clock_ghz"] = cycles / res["kernel_trace"]["average_ns"]
    print(json.dumps(res, indent=1))

main()
