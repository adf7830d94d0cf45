#!/usr/bin/env python3
"""Average duration of the k_runs dispatches that bench.py's `roofline.launch_ms` is over, from the rocprofv3
--kernel-trace CSV of the same command.  Usage: trace_window.py <dir with *kernel_trace.csv> <bench json line file>"""
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
w = line["roofline"]["launch_window"]
rows = [r for r in csv.DictReader(open(f)) if "k_runs<" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
sel = rows[w["first_dispatch"]: w["first_dispatch"] + w["count"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in sel]
allk = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print(json.dumps({"kernel": "k_runs<256, false>", "dispatches_in_trace": len(rows), "window": [w["first_dispatch"], w["count"]],
                  "trace_mean_ms_window": sum(d) / max(len(d), 1), "hip_event_mean_ms_same_run": line["roofline"]["launch_ms"],
                  "trace_mean_ms_all_dispatches": sum(allk) / max(len(allk), 1)}, indent=1))
