"""Overlap analysis of a rocprofv3 kernel trace of bench.py: per kernel total time, union coverage, pairwise concurrency."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rawdtw::", "")))
rows.sort()
# take the densest stretch: last 60% of the k_runs dispatches
key = sys.argv[2] if len(sys.argv) > 2 else "k_scan"
runs = [r for r in rows if r[2].startswith(key)]
t0, t1 = runs[len(runs) // 5][0], runs[-len(runs) // 5][1]
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
wall = t1 - t0
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in sel: tot[n] += e - s; cnt[n] += 1
ev = sorted([(s, 1) for s, e, n in sel] + [(e, -1) for s, e, n in sel])
busy = 0; depth = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[depth] += t - last
    if depth > 0: busy += t - last
    depth += d; last = t
n_batches = cnt[[k for k in cnt if k.startswith("k_runs")][0]]
print("window %.2f ms, %d batches, %.4f ms per batch; some kernel running %.1f%% of the time" % (wall / 1e6, n_batches, wall / 1e6 / n_batches, 100.0 * busy / wall))
print("kernels running at once: " + "  ".join("%d: %.1f%%" % (k, 100.0 * v / wall) for k, v in sorted(hist.items())))
for n, v in tot.most_common(12): print("  %-50s %6d calls  %.4f ms per batch  mean %.1f us" % (n[:50], cnt[n], v / 1e6 / n_batches, v / 1e3 / cnt[n]))
