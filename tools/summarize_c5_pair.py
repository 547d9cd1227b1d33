"""Summary of tools/prof.sh's output for the C5 workload, whose ranking call is TWO kernels (item_sparse_wave_kernel +
the workgroup kernel in list mode for the users it hands over): per-kernel durations and counter sums per call.
python tools/summarize_c5_pair.py gpurun_out/prof_r03_c5 > profiles/r03_c5_pmc.json"""
import csv, glob, json, sys
root = sys.argv[1]
def kind(name):
    return "wave" if "item_sparse_wave" in name else ("workgroup" if "item_sparse_topk" in name else None)
dur = {}
for f in glob.glob(f"{root}/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = kind(r["Kernel_Name"])
        if k:
            dur.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
kernels = {k: {"calls": len(v), "average_ns": sum(v) / len(v)} for k, v in dur.items()}
counters = {}
for f in glob.glob(f"{root}/pmc_*/*/*_counter_collection.csv"):
    acc = {}
    for r in csv.DictReader(open(f)):
        k = kind(r["Kernel_Name"])
        if k:
            acc.setdefault((r["Counter_Name"], k, r["Dispatch_Id"]), 0.0)
            acc[(r["Counter_Name"], k, r["Dispatch_Id"])] += float(r["Counter_Value"])
    per = {}
    for (c, k, _), v in acc.items():
        per.setdefault((c, k), []).append(v)
    for (c, k), v in per.items():
        counters.setdefault(c, {})[k] = sum(v) / len(v)
fetch = sum(counters["FETCH_SIZE"].values()) * 1024.0            # KiB per call, both kernels (guide: x2 on gfx950 for 128-byte requests)
write = sum(counters["WRITE_SIZE"].values()) * 1024.0
per_call_ms = sum(k["average_ns"] for k in kernels.values()) / 1e6
out = {
    "command": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --no-cpu --no-extra --config c5_rp3_ebnerd --steps 3 --warmup 1 (one run per counter group, tools/prof.sh; this summary: tools/summarize_c5_pair.py)",
    "note": "C5 ranking = two kernels per call: item_sparse_wave_kernel (histories <= 64 rows) + item_sparse_topk_kernel<true,true> (the users handed over); counters are means per call and kernel",
    "kernels": kernels, "per_call_ms": per_call_ms, "counters_per_call": counters,
    "hbm_traffic_per_launch_bytes": {"fetch_raw": fetch, "fetch_corrected_x2": 2 * fetch, "write": write, "total_corrected": 2 * fetch + write},
    "hbm_gbps_corrected": (2 * fetch + write) / (per_call_ms * 1e-3) / 1e9,
}
print(json.dumps(out, indent=1))
