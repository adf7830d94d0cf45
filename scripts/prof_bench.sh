#!/bin/bash
# rocprofv3 --kernel-trace --stats over bench.py (run on the GPU box).  Usage: bash scripts/prof_bench.sh <outdir> [bench args]
OUT=${1:-gpurun_out/prof}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv"); sys.exit(1)
for r in list(csv.DictReader(open(f[0])))[:18]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(11), r["Percentage"].rjust(6))
PY
