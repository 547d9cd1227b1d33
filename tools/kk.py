import json,sys
d=json.load(open("gpurun_out/kk.json")); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2), round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_ms"],2), d["wide_path"])
