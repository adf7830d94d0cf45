#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per pass) per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections, json
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        k = k.split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, d in sorted(agg.items()):
    out[k] = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    out[k]["dispatches"] = max(len(v) for v in d.values())
json.dump(out, sys.stdout, indent=1)
