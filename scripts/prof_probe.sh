#!/bin/bash
# rocprofv3 --kernel-trace --stats over scripts/stream_probe.py: every launch of one bench batch alone on the chip (run on the GPU box).
# Usage: bash scripts/prof_probe.sh <outdir> [probe args]
OUT=${1:-gpurun_out/prof_probe}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 scripts/stream_probe.py "$@" > "$OUT/probe.txt" 2> "$OUT/probe.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv"); sys.exit(1)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(11), r["Percentage"].rjust(6))
PY
cat "$OUT/probe.txt"
