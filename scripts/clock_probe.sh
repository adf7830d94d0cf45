#!/bin/bash
# Engine clock under the DTW launch: GRBM_GUI_ACTIVE (GPU-busy clocks) over the kernel trace's durations, every launch of one
# bench batch alone on the chip (scripts/stream_probe.py).  Usage (GPU box): bash scripts/clock_probe.sh <outdir>
OUT=${1:-gpurun_out/clock}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/p -- python3 scripts/stream_probe.py 16384 "" stream_debug=4 > $OUT/p.log 2>&1 || echo "pass failed"
python3 - $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
tr = {r["Dispatch_Id"]: r for f in glob.glob(d + "/p/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        t = tr.get(r["Dispatch_Id"])
        if not t: continue
        dur = int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append((float(r["Counter_Value"]), dur))
for k, v in agg.items():
    if "rawdtw" not in k: continue
    out = {}
    for c, xs in v.items():
        # (the counter is summed over the 8 XCDs; its window is a few microseconds wider than the dispatch: short kernels read high)
        out[c] = "%.0f clocks per XCD / %.1f us = %.2f GHz" % (sum(x for x, _ in xs) / len(xs) / 8, sum(t for _, t in xs) / len(xs) / 1e3, sum(x for x, _ in xs) / 8 / max(sum(t for _, t in xs), 1))
    print(k, out)
PY
