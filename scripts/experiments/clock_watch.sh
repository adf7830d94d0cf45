#!/bin/bash
# engine clock and power while the fresh-batch pipeline runs (sysfs, every ~5 ms): bash scripts/experiments/clock_watch.sh
OUT=gpurun_out/clock_watch; mkdir -p $OUT
H=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
echo "hwmon: $H"; ls $H 2>/dev/null | tr '\n' ' '; echo
( for i in $(seq 1 4000); do echo "$(date +%s.%N) $(cat $H/freq1_input 2>/dev/null) $(cat $H/power1_average 2>/dev/null || cat $H/power1_input 2>/dev/null) $(cat $H/temp1_input 2>/dev/null)"; sleep 0.004; done > $OUT/samples.txt ) &
W=$!
timeout -k 10 300 python bench.py --no-cpu-baseline --modes-reads 0 --rounds 0 > $OUT/bench.json 2> $OUT/bench.err
kill $W 2>/dev/null
python3 - $OUT <<'PY'
import sys, json, collections
rows=[l.split() for l in open(sys.argv[1]+'/samples.txt') if len(l.split())>=3]
f=[int(r[1])/1e6 for r in rows if r[1].isdigit()]
p=[int(r[2])/1e6 for r in rows if r[2].isdigit()]
print('samples', len(rows), 'span s', float(rows[-1][0])-float(rows[0][0]) if rows else 0)
if f:
    c=collections.Counter(int(x/50)*50 for x in f); print('sclk MHz histogram', sorted(c.items()))
if p:
    c=collections.Counter(int(x/50)*50 for x in p); print('power W histogram', sorted(c.items()))
d=json.loads(open(sys.argv[1]+'/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['region_ms'])
PY
