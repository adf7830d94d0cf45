#!/bin/bash
# Workload sensitivity of the bench line (GPU box): anchor density of the true chains and the decoy chains' gaps.
# Usage: bash scripts/sensitivity.sh <outdir>
OUT=${1:-gpurun_out/sens}; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --modes-reads 0 --rounds 0 "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; }
run hit0.10 --hit-prob 0.10
run hit0.20 --hit-prob 0.20
run hit0.40 --hit-prob 0.40
run gap48 --decoy-gap-median 48
run gap200 --decoy-gap-median 200
run decoys4 --decoys-per-read 4
python3 - $OUT <<'PY'
import json, sys, glob, os
rows = []
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    b = d["batch0"]
    rows.append({"case": os.path.basename(f)[:-5], "synth": d["config"]["synth"], "gcups": round(d["value"], 1), "ms_per_step": round(d["ms_per_step"], 4),
                 "kernel_gcups": round(d["kernel_replay"]["gcups"], 1), "dtw_jobs": b["dtw_jobs"], "cells": b["cells"],
                 "cells_per_job": round(b["cells"] / b["dtw_jobs"], 1), "lane_class_share": round(b["tile_class_jobs"] / b["dtw_jobs"], 4),
                 "k_runs_alone_ms": d["launches"]["alone_ms"]["k_runs"], "roofline_frac_alone": round(d["roofline"]["alone"]["frac"], 4)})
json.dump(rows, open(sys.argv[1] + "/summary.json", "w"), indent=1)
for r in rows: print(r)
PY
