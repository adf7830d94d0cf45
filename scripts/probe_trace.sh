cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_probe; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 scripts/stream_probe.py 16384 "" > $OUT/log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_probe/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(11), ("min %.1f" % (float(r["MinNs"]) / 1e3)).rjust(11))
PY
