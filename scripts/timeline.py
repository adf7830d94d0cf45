#!/usr/bin/env python3
"""Print the kernel timeline of a rocprofv3 --kernel-trace CSV (start/end relative to the first dispatch)."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "rawdtw" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
last = rows[-int(sys.argv[2]):] if len(sys.argv) > 2 else rows
for r in last:
    n = r["Kernel_Name"].split("(")[0].replace("void rawdtw::", "").replace("rawdtw::", "")[:28]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{n:28s} q={r.get('Queue_Id','?'):>3s} start {s:9.1f} us  end {e:9.1f} us  dur {e-s:7.1f}")
